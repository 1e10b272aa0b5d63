"""HBM write/copy ceilings next to the decode kernel (100 MB f32 output)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
dev = torch.device("cuda:0")
n = 25_000_000
a = torch.empty(n, device=dev); b = torch.randn(n, device=dev)
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps * 1e3
us = t(lambda: a.fill_(1.0)); print("fill_  100 MB: %.1f us  %.2f TB/s written" % (us, 0.1 / us * 1e3))
us = t(lambda: a.zero_()); print("zero_  100 MB: %.1f us  %.2f TB/s written" % (us, 0.1 / us * 1e3))
us = t(lambda: a.copy_(b)); print("copy_  100 MB: %.1f us  %.2f TB/s read+written" % (us, 0.2 / us * 1e3))
us = t(lambda: torch.mul(b, 2.0, out=a)); print("mul    100 MB: %.1f us  %.2f TB/s read+written" % (us, 0.2 / us * 1e3))
