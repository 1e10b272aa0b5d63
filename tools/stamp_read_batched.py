"""Stamps of the BATCHED prefilter kernel (tools/exp/libgq_stamp.so) on two 12.5 M-element tensors."""
import os, sys
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch, numpy as np
from gq_amd import native
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = [(12500000,), (12500000,)]
args = Namespace(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=0, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256)
params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
q = Quantizer(NearestNeighborCompressor, params, args)
for p in params:
    p.grad = torch.randn(p.shape, device="cuda")
for _ in range(3):
    q.record(0, epoch=1); q.recorded = 0
torch.cuda.synchronize()
grp = q._groups[0][2]
ws = grp.ws
M = grp.ntiles * 64
wl = ws[2 * native.GQ_MAX_PARTIALS + 4:2 * native.GQ_MAX_PARTIALS + 4 + M]
raw = wl[M - 65536:M - 65536 + 512 * 4 * 10 * 2].contiguous().view(torch.int64).view(-1, 10).cpu().numpy().astype(np.float64)
seg, entry, rt0, rt1, ntl = raw[:, :6], raw[:, 6], raw[:, 7], raw[:, 8], raw[:, 9]
names = ["loop top (tile lookup, prefetch issue)", "16 chains", "tracker merge + swaps", "exact rescoring", "next-tile bf16 split", "fix-up + stores"]
cyc = seg.sum(1).mean(); loop_us = (rt1 - rt0).mean() / 100
print("loop: %.0f cycles per wave in %.1f us (%.2f GHz), %.1f tiles per wave, %.0f cycles per tile" % (cyc, loop_us, cyc / loop_us / 1e3, ntl.mean(), cyc / ntl.mean()))
for n, v in zip(names, seg.mean(0)):
    print("  %-40s %6.0f cycles/tile  %5.1f %%" % (n, v / ntl.mean(), 100 * v / cyc))
end = (rt1 - entry.min()) / 100
print("prologue %.1f us; last loop end %.1f us; mean loop end %.1f us" % ((rt0 - entry).mean() / 100, end.max(), end.mean()))
blk = np.arange(len(end)) // 4
for lo, hi in [(0, 128), (128, 256), (256, 384), (384, 512)]:
    m = (blk >= lo) & (blk < hi)
    print("blocks %3d-%3d: loop ends %.1f us avg (min %.1f, max %.1f), %.1f tiles per wave (min %d, max %d), %.0f cycles per tile"
          % (lo, hi - 1, end[m].mean(), end[m].min(), end[m].max(), ntl[m].mean(), ntl[m].min(), ntl[m].max(), (seg.sum(1)[m] / np.maximum(ntl[m], 1)).mean()))
late = np.argsort(end)[-8:]
print("latest waves:", [(int(w // 4), int(w % 4), round(float(end[w]), 1), int(ntl[w])) for w in late])
for w in late[-3:]:
    print("wave (%d,%d): per-tile cycles by phase:" % (w // 4, w % 4), [int(x / max(ntl[w], 1)) for x in seg[w]], "loop start %.1f us" % ((rt0[w] - entry.min()) / 100))
