"""A/B of gq_hsq_levels (25 M-element gradient's 1,562,500 projections) for several library builds on ONE box:
    python tools/ab_levels.py product tools/exp/libgq_X.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
torch.manual_seed(1234)
g = torch.randn(25_000_000, device=dev)
M = g.numel() // 16
codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
lv = torch.empty(M, dtype=torch.uint8, device=dev); lbub = torch.empty(2, device=dev)
ws = native.new_workspace(dev, M)
native.hsq_encode(g, cb, codes, u, ws)
for _ in range(2000):
    native.hsq_levels(u, 6, 0, None, 0, ws, lbub, lv)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(2000):
    native.hsq_levels(u, 6, 0, None, 0, ws, lbub, lv)
e.record(); torch.cuda.synchronize()
a = s.elapsed_time(e) / 2000 * 1e3
s.record()
for _ in range(500):
    native.hsq_encode(g, cb, codes, u, ws)
    native.hsq_levels(u, 6, 0, None, 0, ws, lbub, lv)
e.record(); torch.cuda.synchronize()
print("%%.2f %%.2f" %% (a, s.elapsed_time(e) / 500 * 1e3))
''' % ROOT
libs = sys.argv[1:] or ["product"]
for rep in range(2):
    for l in libs:
        env = dict(os.environ)
        if l != "product":
            env["GQ_LIB_PATH"] = os.path.join(ROOT, l)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print("%-36s levels back to back / encode+levels: %s us" % (l, out[-1] if out else "?"), flush=True)
