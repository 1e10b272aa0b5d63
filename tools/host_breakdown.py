"""Where the host time of one PSQuantizer step goes (ResNet-50 list, HSQ d16 k8 n6, device RNG): wall-clock
sections of record() / apply(), GPU idle in between.  python tools/host_breakdown.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
from argparse import Namespace
import torch
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd import quantizers as Q
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
args = Namespace(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp", num_users=1,
                 mode="ps", cr=256)
params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
q = Q.Quantizer(NearestNeighborCompressor, params, args)
grads = [torch.randn(p.shape, device="cuda") * 1e-3 for p in params]
T = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
grp = None
def step():
    for p, g in zip(params, grads):
        p.grad = g
    q.record(0, epoch=1)
    q.apply()
for _ in range(5):
    step()
obj = q._groups[0][2]
obj._upload = timed("upload (pointers -> pinned header -> H2D)", obj._upload)
obj.encode = timed("group.encode (incl. upload)", obj.encode)
obj.decode_mean = timed("group.decode_mean", obj.decode_mean)
obj._batch.decode = timed("  native decode call", obj._batch.decode)
obj._batch.encode = timed("  native encode call", obj._batch.encode)
obj._batch.levels = timed("  native levels call", obj._batch.levels)
obj._out_buffer = timed("  _out_buffer", obj._out_buffer)
q.record = timed("record", q.record)
q.apply = timed("apply", q.apply)
q._decode_all = timed("_decode_all", q._decode_all)
torch._foreach_copy_ = timed("_foreach_copy_ (dense tensors -> wire)", torch._foreach_copy_)
N = 200
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    t1 = time.perf_counter()
    for p, g in zip(params, grads):
        p.grad = g
    T["set p.grad (bench harness)"] = T.get("set p.grad (bench harness)", 0.0) + time.perf_counter() - t1
    q.record(0, epoch=1)
    q.apply()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N * 1e6
print("step %.1f us" % tot)
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-48s %7.1f us" % (k, v / N * 1e6))
