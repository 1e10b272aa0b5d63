"""A/B timing of gq_pvq_encode (25 M elements, K 256; GQ_AB_D = sub-dimension, default 16) for library builds, alternated in
child processes on one box:   python tools/pvq_ab.py product tools/exp/libgq_pvq_X.so ...   ('product' = the in-tree library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "gradient-quantization_amd"))
import numpy as np, torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
D = int(os.environ.get("GQ_AB_D", "16"))
cb = load_codebook(D, 256)
cdag = torch.from_numpy(np.linalg.pinv(cb.T).astype(np.float32)).contiguous().to(dev)
torch.manual_seed(1234)
g = torch.randn(25_000_000 // D * D, device=dev) * 1e-2
M = g.numel() // D
codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M)
for _ in range(200):
    native.pvq_encode(g, cdag, codes, u, ws, native.RANDOM_DEVICE, None, 7)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(300):
    native.pvq_encode(g, cdag, codes, u, ws, native.RANDOM_DEVICE, None, 7)
e.record(); torch.cuda.synchronize()
clk = ""
if "diag16" in os.environ.get("GQ_LIB_PATH", ""):
    clk = " (in-kernel clock %%.2f GHz)" %% (u[0].item() / u[1].item() * 0.1)
print("%%.1f%%s" %% (s.elapsed_time(e) / 300 * 1e3, clk))
''' % ROOT
if __name__ == "__main__":
    libs = sys.argv[1:] or ["product"]
    rounds = int(os.environ.get("GQ_AB_ROUNDS", "2"))
    res = {l: [] for l in libs}
    for _ in range(rounds):
        for l in libs:
            env = dict(os.environ)
            if l != "product":
                env["GQ_LIB_PATH"] = os.path.join(ROOT, l) if not os.path.isabs(l) else l
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
            res[l].append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-200:])
    for l in libs:
        print("%-40s %s us" % (l, "  ".join(res[l])))
