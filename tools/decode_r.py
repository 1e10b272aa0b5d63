"""decode-mean time as a function of the number of payloads R (what an N-rank step pays after the all-gather)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.codebook import load_codebook
from gq_amd.wire import HSQWire
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
SIZE = 25_000_000
M = SIZE // 16
wire = HSQWire(M)
u = torch.empty(M, dtype=torch.float32, device=dev)
ws = native.new_workspace(dev, M)
out = torch.empty(SIZE, dtype=torch.float32, device=dev)
for R in ([int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]):
    gathered = wire.alloc(dev, ranks=R)
    for r in range(R):
        torch.manual_seed(1234 + r)
        g = torch.randn(SIZE, device=dev)
        codes, levels, lb_ub = wire.views(gathered[r])
        native.hsq_encode(g, cb, codes, u, ws)
        native.hsq_levels(u, 6, 0, None, 0, ws, lb_ub, levels)
    def timed(n_bit):
        def dec():
            native.hsq_decode_sum_packed(gathered, M, cb, n_bit, out, R, wire.codes_off, wire.levels_off, wire.lbub_off)
        for _ in range(3): dec()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): dec()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 20 * 1e3
    us = timed(6)
    exact = out.clone()
    fma = timed(6 | native.AGGREGATE_FMA) if R >= 2 else us      # the opt-in fused accumulation (GQ_AGGREGATE_FMA; R in {2, 4, 8, 16} have a kernel of their own)
    rel = float((out.double() - exact.double()).norm() / exact.double().norm()) if R >= 2 else 0.0
    print("R=%d: %.1f us  (%.2f TB/s of output, %.1f B algorithmic per output element); GQ_AGGREGATE_FMA: %.1f us, relative L2 distance to the exact aggregate %.1e"
          % (R, us, 0.1 / us * 1e3, 2 * R / 16 + 4, fma, rel))
