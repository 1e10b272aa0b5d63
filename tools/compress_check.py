"""gq_hsq_compress (encode + levels in one C call) against gq_hsq_encode + gq_hsq_levels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
def ev(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
ok = True
for M in (1, 63, 64, 65, 1000, 4099, 200001, 1562500):
    for mode in (0, 2, 1):
        torch.manual_seed(M + mode)
        g = torch.randn(M * 16, device=dev) * 0.01
        r = torch.rand(M, device=dev) if mode == 1 else None
        outs = []
        for fused in (False, True):
            codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
            ws = native.new_workspace(dev, M); lbub = torch.empty(2, dtype=torch.float32, device=dev)
            lv = torch.empty(M, dtype=torch.uint8, device=dev)
            if fused:
                native.hsq_compress(g, cb, codes, u, ws, 6, mode, r, 99, lbub, lv)
            else:
                native.hsq_encode(g, cb, codes, u, ws)
                native.hsq_levels(u, 6, mode, r, 99, ws, lbub, lv)
            torch.cuda.synchronize()
            outs.append((codes, u, lbub, lv))
        same = all(torch.equal(a.view(torch.uint8) if a.dtype == torch.uint8 else a.view(torch.int32), b.view(torch.uint8) if b.dtype == torch.uint8 else b.view(torch.int32)) for a, b in zip(*outs))
        ok &= same
        if not same:
            print("MISMATCH M=%d mode=%d" % (M, mode), [bool(torch.equal(a, b)) for a, b in zip(*outs)])
print("fused == separate:", ok)
M = 1562500
g = torch.randn(M * 16, device=dev)
codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M); lbub = torch.empty(2, dtype=torch.float32, device=dev); lv = torch.empty(M, dtype=torch.uint8, device=dev)
def sep():
    native.hsq_encode(g, cb, codes, u, ws); native.hsq_levels(u, 6, 0, None, 0, ws, lbub, lv)
one = lambda: native.hsq_compress(g, cb, codes, u, ws, 6, 0, None, 0, lbub, lv)
for _ in range(3):
    print("25 M elements: two calls %.1f us, gq_hsq_compress %.1f us" % (ev(sep), ev(one)))
