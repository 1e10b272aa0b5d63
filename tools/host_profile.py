"""Host-side cost of one record() + apply() over the ResNet-50 parameter list (cProfile, top entries)."""
import cProfile, json, os, pstats, sys
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer

shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
base = dict(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp",
            num_users=1, mode="ps", cr=256)
params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
q = Quantizer(NearestNeighborCompressor, params, Namespace(**base))
grads = [torch.randn(p.shape, device="cuda") * 1e-3 for p in params]


def step():
    for p, g in zip(params, grads):
        p.grad = g
    q.record(0, epoch=1)
    q.apply()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 35)
