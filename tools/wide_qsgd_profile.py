import json, os, sys, time
from argparse import Namespace
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd.compressors import QSGDCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
base = dict(c_dim=0, k_bit=8, n_bit=1, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256)
params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    q = Quantizer(QSGDCompressor, params, Namespace(**base))
grads = [torch.randn(p.shape, device="cuda") * 1e-3 for p in params]
for _ in range(30):
    for p, g in zip(params, grads):
        p.grad = g
    q.record(0, epoch=1)
    q.apply()
torch.cuda.synchronize()
