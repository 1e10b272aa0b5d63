"""gq_hsq_encode_batched on ONE 25 M-element tensor (hand-made segment table) next to gq_hsq_encode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
torch.manual_seed(1234)
N = 25_000_000
g = torch.randn(N, device=dev)
M = N // 16
ntiles = (M + 63) // 64
def up(x): return (x + 15) & ~15
wire = torch.zeros(up(M) * 2 + 16, dtype=torch.uint8, device=dev)
table = torch.zeros((1, 8), dtype=torch.int64)
table[0, 0], table[0, 1], table[0, 2], table[0, 3], table[0, 4], table[0, 5] = g.data_ptr(), M, 0, 0, up(M), 2 * up(M)
seg_table = table.to(dev)
tile_seg = torch.zeros(ntiles, dtype=torch.int32, device=dev)
minmax = torch.tensor([[-1, 0]], dtype=torch.int32, device=dev)
u_flat = torch.empty(ntiles * 64, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, ntiles * 64)
codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev); ws1 = native.new_workspace(dev, M)
def ev(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
tb = ev(lambda: native.hsq_encode_batched(seg_table, tile_seg, 1, ntiles, cb, wire, u_flat, minmax, ws))
ts = ev(lambda: native.hsq_encode(g, cb, codes, u, ws1))
print("one tensor: batched entry point %.1f us, single-tensor entry point %.1f us; same codes: %s" % (tb, ts, torch.equal(wire[:M], codes)))
