"""The stamps (GQ_LIB_PATH=gradient-quantization_amd/libgq_hsq_clock.so) of the MULTI-TENSOR d16 encode next to the flat one on the
same elements: cycles per tile and phase, prologue, end of run -- where does the segment-table form lose its ~10 %?
    GQ_LIB_PATH=$PWD/gradient-quantization_amd/libgq_hsq_clock.so python tools/stamp_batched.py"""
import contextlib, json, os, sys
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import numpy as np, torch
from gq_amd import native
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
dev = torch.device("cuda:0")
names = ["prefetch issue / loop top", "16 chains (MFMA + keys)", "tracker merge + swaps", "exact rescoring (LDS gather)",
         "next-tile f16 conversion", "queueing of unsettled subvectors + stores"]


def report(label, ws, M):
    wl = ws[native.WS_LOG_FIRST:native.WS_LOG_FIRST + M]
    raw = wl[M - 65536:M - 65536 + 256 * 8 * 12 * 2].contiguous().view(torch.int64).view(-1, 12).cpu().numpy().astype(np.float64)
    seg, entry, rt0, rt1, rt2, drained = raw[:, :6], raw[:, 6], raw[:, 7], raw[:, 8], raw[:, 10], raw[:, 11]
    tiles_w = raw[:, 9]
    cyc, tiles = seg.sum(1).mean(), tiles_w.mean()
    loop_us = (rt1 - rt0).mean() / 100
    print("%s: %.1f tiles per wave, %.0f cycles per tile and wave, loop %.1f us at %.2f GHz" % (label, tiles, cyc / tiles, loop_us, cyc / loop_us / 1e3))
    for n, v in zip(names, seg.mean(0)):
        print("    %-44s %6.0f cycles/tile  %5.1f %%" % (n, v / tiles, 100 * v / cyc))
    wend = ((rt2 - entry.min()) / 100).reshape(-1, 8)
    print("    prologue %.2f us; loop end -> wave end %.2f us (max %.2f); workgroup ends: mean %.1f, max %.1f us; second-pass entries per wave %.1f"
          % ((rt0 - entry).mean() / 100, ((rt2 - rt1) / 100).mean(), ((rt2 - rt1) / 100).max(), wend.max(1).mean(), wend.max(), (drained % 1000).mean()))
    per_wg = tiles_w.reshape(-1, 8).sum(1)
    cyc_wg = (seg.sum(1) / np.maximum(tiles_w, 1)).reshape(-1, 8).mean(1)
    print("    tiles per workgroup min/mean/max %d/%.1f/%d; cycles per tile by workgroup octile: %s"
          % (per_wg.min(), per_wg.mean(), per_wg.max(), " ".join("%.0f" % cyc_wg[o * 32:(o + 1) * 32].mean() for o in range(8))))


def group_for(shapes):
    args = Namespace(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=0, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256, gq_graph=False)
    params = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
    with contextlib.redirect_stdout(sys.stderr):
        q = Quantizer(NearestNeighborCompressor, params, args)
    for p in params:
        p.grad = torch.randn(p.shape, device=dev) * 1e-3
    q.record(0, epoch=1); q.apply()
    for p in params:
        p.grad = torch.randn(p.shape, device=dev) * 1e-3
    grp = q._groups[0][2]
    gl = [params[i].grad.data for i in grp.idxs]
    grp.encode(gl, q._wire[0], 0, 0)
    for _ in range(3):
        grp._batch.encode(q._wire[0], None, -1)
    torch.cuda.synchronize()
    return q, grp, gl


q, grp, gl = group_for(shapes)
report("multi-tensor, the ResNet-50 list (76 tensors)", grp.ws, grp.ntiles * 64)
flat = torch.cat([g.reshape(-1) for g in gl])
M = flat.numel() // 16
codes, u, ws = torch.empty(M, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.float32, device=dev), native.new_workspace(dev, M)
for _ in range(3):
    native.hsq_encode(flat, grp.codebook, codes, u, ws, impl=4)
torch.cuda.synchronize()
report("flat, the same elements as one tensor", ws, M)
q2, grp2, gl2 = group_for([(11796480,), (11796480,)])
report("multi-tensor, two equal tensors", grp2.ws, grp2.ntiles * 64)
