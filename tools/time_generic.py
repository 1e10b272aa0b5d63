"""Time gq_hsq_encode for several (d, K) on a 25M-element gradient: the generic exact-f32 MFMA
kernel (impl 2) next to what the dispatcher picks (impl 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
dev = torch.device("cuda:0")
torch.manual_seed(1234)
N = 25_000_000
cases = [(8, 256), (16, 256), (32, 256), (16, 64), (16, 32), (16, 16), (8, 32), (32, 64), (32, 32), (24, 64), (12, 512), (16, 1024), (16, 4096), (32, 4096), (128, 256)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for d, K in cases:
    n = N // d * d
    g = torch.randn(n, device=dev)
    cb = torch.randn(K, d, device=dev)
    cb = cb / cb.norm(dim=1, keepdim=True)
    M = n // d
    codes = torch.empty(M, dtype=torch.uint8 if K <= 256 else torch.int32, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = native.new_workspace(dev, M)
    ref = None
    for impl in (5 if d <= 96 else 2, 0):
        for _ in range(2):
            native.hsq_encode(g, cb, codes, u, ws, impl=impl)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        s.record()
        for _ in range(reps):
            native.hsq_encode(g, cb, codes, u, ws, impl=impl)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / reps * 1e3
        if ref is None:
            ref = (codes.clone(), u.clone())
            same = ""
        else:
            same = " identical to the exact kernel" if torch.equal(ref[0], codes) and torch.equal(ref[1].view(torch.int32), u.view(torch.int32)) else " DIFFERS from the exact kernel"
        flops = 2.0 * n * K
        print("d=%3d K=%4d impl=%d: %8.1f us  %7.2f G elements/s  %6.1f TFLOP/s f32-equivalent  %5.2f TB/s read%s"
              % (d, K, impl, us, n / us / 1e3, flops / us / 1e6, 4.0 * n / us / 1e6, same))
