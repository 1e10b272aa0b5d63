"""Does the d16 encode care where its input comes from?  The same 100 MB gradient launch after launch (it stays in the 256 MB
memory-side cache) against four gradients in turn (every launch reads HBM): time per launch and -- with the stamped twin
(GQ_LIB_PATH=.../libgq_hsq_clock.so) -- cycles per tile and phase, in-kernel clock.    python tools/cold_vs_warm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch, numpy as np
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
torch.manual_seed(1234)
N = 25_000_000
gs = [torch.randn(N, device=dev) for _ in range(4)]
M = N // 16
codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M)
stamped = "clock" in os.environ.get("GQ_LIB_PATH", "")
names = ["loop top", "16 chains", "tracker merge", "rescoring", "f16 conversion", "queueing + stores"]

def stamps():
    wl = ws[native.WS_LOG_FIRST:native.WS_LOG_FIRST + M]
    raw = wl[M - 65536:M - 65536 + 256 * 8 * 12 * 2].contiguous().view(torch.int64).view(-1, 12).cpu().numpy().astype(np.float64)
    seg, entry, rt0, rt1, rt2 = raw[:, :6], raw[:, 6], raw[:, 7], raw[:, 8], raw[:, 10]
    tiles = raw[:, 9].mean()
    cyc = seg.sum(1).mean()
    loop_us = (rt1 - rt0).mean() / 100
    return ("%.0f cycles per tile at %.2f GHz (loop %.1f us, prologue %.2f us, launch %.1f us): " % (cyc / tiles, cyc / loop_us / 1e3, loop_us, (rt0 - entry).mean() / 100, (rt2.max() - entry.min()) / 100)
            + "  ".join("%s %.0f" % (n, v / tiles) for n, v in zip(names, seg.mean(0))))

def timed(rotate, reps=60):
    for i in range(8):
        native.hsq_encode(gs[i % 4 if rotate else 0], cb, codes, u, ws, impl=4)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps):
        native.hsq_encode(gs[i % 4 if rotate else 0], cb, codes, u, ws, impl=4)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

for rnd in range(3):
    for rotate in (False, True):
        us = timed(rotate)
        print("%-28s %6.2f us per launch%s" % ("four gradients in turn:" if rotate else "one gradient:", us, ("   " + stamps()) if stamped else ""))
