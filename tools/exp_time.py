"""Time gq_hsq_encode (25M elements) for one library build: GQ_LIB_PATH=... [GQ_AB_D=8|16|32] python tools/exp_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
D = int(os.environ.get("GQ_AB_D", "16"))
cb = torch.from_numpy(load_codebook(D, 256)).to(dev)
torch.manual_seed(1234)
g = torch.randn(25_000_000, device=dev)
M = g.numel() // D
codes = torch.empty(M, dtype=torch.uint8, device=dev)
u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M)
native.mark_worklist(ws, M)
impl = int(os.environ.get("IMPL", "4"))
for _ in range(5):
    native.hsq_encode(g, cb, codes, u, ws, impl=impl)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30):
    native.hsq_encode(g, cb, codes, u, ws, impl=impl)
e.record()
torch.cuda.synchronize()
print("%s impl=%d: %.1f us per encode (one launch), fixups=%d" % (
    os.path.basename(os.environ.get("GQ_LIB_PATH", "product")), impl, s.elapsed_time(e) / 30 * 1e3, native.fixup_count(ws, M)))
