import os, sys, time, json
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
from argparse import Namespace
import torch
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
args=Namespace(c_dim=16,k_bit=8,n_bit=6,no_cuda=False,random=1,ef=False,two_phase=False,scale="exp",num_users=1,mode="ps",cr=256,gq_rng=os.environ.get("GQ_PROF_RNG", "device"))
params=[torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
q=Quantizer(NearestNeighborCompressor, params, args)
grads=[torch.randn(p.shape, device="cuda")*1e-3 for p in params]
def step():
    for p,g in zip(params,grads): p.grad=g
    q.record(0,epoch=1); q.apply()
for _ in range(3): step()
torch.cuda.synchronize()
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
t0=time.perf_counter(); r=torch.rand(q._draw_total); print("rand ms", (time.perf_counter()-t0)*1e3, q._draw_total, torch.get_num_threads())
