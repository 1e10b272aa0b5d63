"""Read the stamps of tools/exp/libgq_stamp.so (tools/stamp_build.py): where a tile's cycles go,
the in-kernel clock, prologue length and how far apart the waves finish."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch, numpy as np
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
torch.manual_seed(1234)
g = torch.randn(25_000_000, device=dev)
M = g.numel() // 16
codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M)
for _ in range(3):
    native.hsq_encode(g, cb, codes, u, ws, impl=4)
torch.cuda.synchronize()
wl = ws[native.WS_LOG_FIRST:native.WS_LOG_FIRST + M]
raw = wl[M - 65536:M - 65536 + 256 * 8 * 12 * 2].contiguous().view(torch.int64).view(-1, 12).cpu().numpy().astype(np.float64)
seg, entry, rt0, rt1, rt2, drained = raw[:, :6], raw[:, 6], raw[:, 7], raw[:, 8], raw[:, 10], raw[:, 11]
names = ["prefetch issue / loop top", "16 chains (MFMA + keys)", "tracker merge + swaps", "exact rescoring (LDS gather)",
         "next-tile f16 conversion", "queueing of unsettled subvectors + stores"]
tiles_w = raw[:, 9]                                   # tiles each wave actually processed (dynamic scheduling)
first = (np.arange(len(raw)) % 8) < 4          # waves 0-3 of a workgroup (one per SIMD) vs waves 4-7
tiles = tiles_w.mean()
cyc = seg.sum(1).mean()
loop_us = (rt1 - rt0).mean() / 100
print("loop: %.0f shader cycles per wave in %.1f us -> in-kernel clock %.2f GHz; %.0f cycles per tile per wave" % (cyc, loop_us, cyc / loop_us / 1e3, cyc / tiles))
for n, v in zip(names, seg.mean(0)):
    print("  %-30s %6.0f cycles/tile  %5.1f %%" % (n, v / tiles, 100 * v / cyc))
print("prologue per wave %.1f us (min %.1f, max %.1f); first entry -> last loop end %.1f us; loop-end skew %.1f us"
      % ((rt0 - entry).mean() / 100, (rt0 - entry).min() / 100, (rt0 - entry).max() / 100, (rt1.max() - entry.min()) / 100, (rt1.max() - rt1.min()) / 100))
scans, drained = drained // 1000, drained % 1000
print("second pass: %.1f queued subvectors per wave on average (max %d; %d per launch of the first 256 workgroups), %d of them scanned exactly afterwards; loop end -> end of the wave %.2f us on average (max %.2f); the last wave ends %.1f us after the first entry"
      % (drained.mean(), drained.max(), drained.sum(), scans.sum(), ((rt2 - rt1) / 100).mean(), ((rt2 - rt1) / 100).max(), (rt2.max() - entry.min()) / 100))
end = (rt1 - entry.min()) / 100
blk = np.arange(len(end)) // 8
for lo, hi in [(0, 64), (64, 128), (128, 192), (192, 256)]:
    m = (blk >= lo) & (blk < hi)
    print("blocks %3d-%3d: loop ends at %.1f us on average (min %.1f, max %.1f), loop length %.1f us, %.1f tiles per wave, %.0f cycles per tile"
          % (lo, hi - 1, end[m].mean(), end[m].min(), end[m].max(), ((rt1 - rt0)[m] / 100).mean(), tiles_w[m].mean(), (seg.sum(1)[m] / tiles_w[m]).mean()))
xcd = blk % 8
print("per XCD (workgroup index % 8), first-dispatched half of the grid: tiles per wave / cycles per tile / loop end us")
for x in range(8):
    m = (xcd == x) & first
    print("  xcd %d: %.2f tiles, %.0f cycles/tile, ends %.1f us (max %.1f) | second half: %.2f tiles, ends %.1f us (max %.1f)"
          % (x, tiles_w[m].mean(), (seg.sum(1)[m] / np.maximum(tiles_w[m], 1)).mean(), end[m].mean(), end[m].max(),
             tiles_w[(xcd == x) & ~first].mean(), end[(xcd == x) & ~first].mean(), end[(xcd == x) & ~first].max()))
cu = blk
slow = np.argsort([end[cu == c].max() for c in range(256)])[-8:]
print("slowest CUs-slots (workgroup index % 256):", slow, [round(float(end[cu == c].max()), 1) for c in slow])

# where the launch's tail comes from: the end of every workgroup (its last wave, scans included) and of every wave inside it
wend = ((rt2 - entry.min()) / 100).reshape(-1, 8)
wg_end, wg_mean = wend.max(1), wend.mean(1)
print("workgroup ends (last wave, scans included): mean %.1f us, percentiles 10/50/90/99/100: %s" % (wg_end.mean(), np.round(np.percentile(wg_end, [10, 50, 90, 99, 100]), 1)))
print("inside a workgroup: last wave - mean wave = %.2f us on average (max %.2f); first wave to stop - last = %.2f us on average"
      % ((wg_end - wg_mean).mean(), (wg_end - wg_mean).max(), (wg_end - wend.min(1)).mean()))
print("across workgroups: latest workgroup end - mean workgroup end = %.2f us; mean workgroup end - mean wave end = %.2f us" % (wg_end.max() - wg_end.mean(), wg_end.mean() - wend.mean()))
# is the spread across workgroups systematic?  ends by XCD (index % 8) and by dispatch order (index // 32)
bx = np.arange(256) % 8
print("workgroup end by XCD (index %% 8):      " + "  ".join("%d: %.1f" % (x, wg_end[bx == x].mean()) for x in range(8)))
print("workgroup end by dispatch octile:       " + "  ".join("%d: %.1f" % (o, wg_end[(np.arange(256) // 32) == o].mean()) for o in range(8)))
pro = ((rt0 - entry) / 100).reshape(-1, 8).mean(1)
tl = tiles_w.reshape(-1, 8).sum(1)
print("prologue by dispatch octile (us):       " + "  ".join("%d: %.2f" % (o, pro[(np.arange(256) // 32) == o].mean()) for o in range(8)))
print("correlation of a workgroup's end with: its prologue %.2f, its entry time %.2f, its tiles %.2f, its second-pass entries %.2f"
      % (np.corrcoef(wg_end, pro)[0, 1], np.corrcoef(wg_end, ((entry - entry.min()) / 100).reshape(-1, 8).mean(1))[0, 1],
         np.corrcoef(wg_end, tl)[0, 1], np.corrcoef(wg_end, drained.reshape(-1, 8).sum(1))[0, 1]))
