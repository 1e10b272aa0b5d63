"""Batched (segment table) compress on two 12.5 M-element tensors vs the single-tensor kernels on 25 M."""
import os, sys
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = [(int(x),) for x in (sys.argv[1:] or ["12500000", "12500000"])]
args = Namespace(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=0, ef=False, two_phase=False, scale="exp",
                 num_users=1, mode="ps", cr=256)
params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
q = Quantizer(NearestNeighborCompressor, params, args)
for p in params:
    p.grad = torch.randn(p.shape, device="cuda")
q.record(0, epoch=1)
grp = q._groups[0][2]
gl = [p.grad.data for p in params]
wire = q._wire[0]
def ev(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print("batched compress (encode + levels + header upload) of %s: %.1f us" % (shapes, ev(lambda: grp.encode(gl, wire, 0, 0))))
from gq_amd import native
seg_table = grp._dev[:grp._table_words]
minmax = grp._dev[grp._table_words:].view(torch.int32)
enc = lambda: native.hsq_encode_batched(seg_table, grp.tile_seg, grp.nseg, grp.ntiles, grp.codebook, wire, grp.u_flat, minmax, grp.ws)
lev = lambda: native.hsq_levels_batched(seg_table, grp.tile_seg, grp.nseg, grp.ntiles, grp.u_flat, minmax, grp.n_bit, 0, 0, wire)
print("kernels only: batched encode %.1f us, batched levels %.1f us" % (ev(enc), ev(lev)))
