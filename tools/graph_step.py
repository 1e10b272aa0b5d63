"""Does a HIP graph of the step (encode + levels + decode-mean) beat three stream launches?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
import gc
gc.disable()      # (a CUDAGraph finalized by the collector INSIDE a stream capture takes the process down: gq_amd.quantizers._capturing)
from gq_amd import native
from gq_amd.codebook import load_codebook
from gq_amd.wire import HSQWire
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
torch.manual_seed(1234)
SIZE = 25_000_000
g = torch.randn(SIZE, device=dev)
M = SIZE // 16
wire = HSQWire(M)
payload = wire.alloc(dev)
codes, levels, lb_ub = wire.views(payload)
u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M)
out = torch.empty(SIZE, dtype=torch.float32, device=dev)
def step():
    native.hsq_encode(g, cb, codes, u, ws)
    native.hsq_levels(u, 6, 0, None, 0, ws, lb_ub, levels)
    native.hsq_decode_sum_packed(payload.view(1, -1), M, cb, 6, out, 1, wire.codes_off, wire.levels_off, wire.lbub_off)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("stream launches: %.1f us/step" % timeit(step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s):
    step()
ref = out.clone()
print("graph replay:    %.1f us/step" % timeit(gr.replay))
print("same result:", torch.equal(ref, out))
