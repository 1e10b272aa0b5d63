"""gq_hsq_encode on 25 M elements for d = 8 / 16 / 32 (K = 256, the packaged codebooks): the f16 prefilter kernel (impl 4,
hsq_encode_pf.hip) next to the exact f32 MFMA kernel (impl 1 for d = 16, impl 5 otherwise), 1000 untimed + 1000 timed launches
of the prefilter (50 of the exact kernel), outputs compared bit for bit.    python tools/time_pf_d.py [d ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
torch.manual_seed(1234)
N = 25_000_000
dims = [int(a) for a in sys.argv[1:]] or [32, 8, 16]
for d in dims:
    n = N // d * d
    g = torch.randn(n, device=dev)
    cb = torch.from_numpy(load_codebook(d, 256)).to(dev)
    M = n // d
    out = {}
    for impl in ((4, 5) if d != 16 else (4, 1)):
        codes = torch.empty(M, dtype=torch.uint8, device=dev)
        u = torch.empty(M, dtype=torch.float32, device=dev)
        ws = native.new_workspace(dev, M)
        reps = 1000 if impl == 4 else 50
        for _ in range(reps):
            native.hsq_encode(g, cb, codes, u, ws, impl=impl)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            native.hsq_encode(g, cb, codes, u, ws, impl=impl)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / reps * 1e3
        out[impl] = (codes, u)
        print("d=%2d impl=%d: %7.2f us   %6.3f of the 8 TB/s roofline (%.4f B per element)" % (d, impl, us, (4.0 + 2.0 / d) * n / us / 1e6 / 8.0, 4.0 + 2.0 / d), flush=True)
    a, b = list(out.values())
    same = torch.equal(a[0], b[0]) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
    print("d=%2d: outputs %s" % (d, "bit-identical" if same else "DIFFER (%d codes)" % int((a[0] != b[0]).sum())), flush=True)
