"""Encode time vs size: separates fixed cost from per-tile cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
torch.manual_seed(1234)
gfull = torch.randn(50_000_000, device=dev)
for n in [1024, 1_000_000, 6_250_000, 12_500_000, 25_000_000, 50_000_000]:
    g = gfull[:n]
    M = n // 16
    codes = torch.empty(M, dtype=torch.uint8, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = native.new_workspace(dev, M)
    for impl in (4, 1):
        for _ in range(3):
            native.hsq_encode(g, cb, codes, u, ws, impl=impl)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            native.hsq_encode(g, cb, codes, u, ws, impl=impl)
        e.record()
        torch.cuda.synchronize()
        print("n=%9d impl=%d: %.1f us" % (n, impl, s.elapsed_time(e) / 20 * 1e3))
