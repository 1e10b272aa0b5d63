import json, os, sys, time, cProfile, pstats
from argparse import Namespace
sys.path.insert(0, "gradient-quantization_amd")
import torch
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open("tests/golden/resnet50_cifar_shapes.json"))["parameter_shapes"]
base = dict(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256)
params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
q = Quantizer(NearestNeighborCompressor, params, Namespace(**base))
grads = [torch.randn(p.shape, device="cuda") * 1e-3 for p in params]
def step():
    for p, g in zip(params, grads): p.grad = g
    q.record(0, epoch=1); q.apply()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
