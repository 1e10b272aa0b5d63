"""gq_pvq_encode on 25 M elements (d 16, K 256): the MFMA kernel and, with GQ_PVQ_VALU=1, the VALU cross-check."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import numpy as np, torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
cb = load_codebook(16, 256)
cdag = torch.from_numpy(np.linalg.pinv(cb.T).astype(np.float32)).contiguous().to(dev)
g = torch.randn(25_000_000, device=dev) * 1e-2
M = g.numel() // 16
codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M)
for _ in range(200):     # (the clock settles over the first ~100 launches: 10 timed launches after 3 read 218 us for a 150 us kernel)
    native.pvq_encode(g, cdag, codes, u, ws, native.RANDOM_DEVICE, None, 7)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(300):
    native.pvq_encode(g, cdag, codes, u, ws, native.RANDOM_DEVICE, None, 7)
e.record(); torch.cuda.synchronize()
print("%s: %.1f us per 25 M elements" % ("VALU kernel" if os.environ.get("GQ_PVQ_VALU") else "two-sweep MFMA kernel" if os.environ.get("GQ_PVQ_TWO_SWEEPS") else "one-sweep MFMA kernel", s.elapsed_time(e) / 300 * 1e3))
