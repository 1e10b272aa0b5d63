"""A/B timing of gq_hsq_encode (25 M elements; GQ_AB_D = sub-dimension, default 16) for library builds, alternated in child processes on one box:
    python tools/ab_time.py product tools/exp/libgq_B.so ...      ('product' = the in-tree library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
D = int(os.environ.get("GQ_AB_D", "16"))
cb = torch.from_numpy(load_codebook(D, 256)).to(dev)
torch.manual_seed(1234)
g = torch.randn(25_000_000, device=dev)
if os.environ.get("GQ_AB_INPUT") == "zeros":      # DVFS check: the same binary on all-zero data (MI355X_MICROARCH.md, DVFS give-back item 1)
    g.zero_()
M = g.numel() // D
codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M)
for _ in range(3000):
    native.hsq_encode(g, cb, codes, u, ws)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(2000):
    native.hsq_encode(g, cb, codes, u, ws)
e.record(); torch.cuda.synchronize()
print("%%.2f" %% (s.elapsed_time(e) / 2000 * 1e3))
''' % ROOT
libs = sys.argv[1:] or ["product"]
res = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        env = dict(os.environ)
        if l != "product":
            env["GQ_LIB_PATH"] = os.path.join(ROOT, l)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        res[l].append(float(out[-1]) if out else float("nan"))
for l in libs:
    print("%-32s %s us" % (l, res[l]))
