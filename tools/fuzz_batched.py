"""Randomised check of the multi-tensor kernels against the per-tensor path: random (d, K) with synthetic
codebooks, random tensor lists (ragged tiles, tiny and large tensors), users, error feedback, two-phase, ring;
HSQ (byte and 6-bit packed levels) and QSGD (incl. wide buckets).  Every aggregate, wire and residual must match bit for bit.
    python tools/fuzz_batched.py [seconds] [seed]"""
import contextlib, gc, io, os, sys, tempfile, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import numpy as np
import torch
from gq_amd.compressors import NearestNeighborCompressor, QSGDCompressor
from gq_amd.quantizers import Quantizer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp()
os.chdir(tmp)


def codebook(d, K):
    path = os.path.join(tmp, "codebooks", "learned_codebook")
    os.makedirs(path, exist_ok=True)
    f = os.path.join(path, "angular_dim_%d_Ks_%d.fvecs" % (d, K))
    if not os.path.exists(f):
        cb = rng.standard_normal((K, d)).astype(np.float32)
        rows = np.empty((K, d + 1), dtype="<i4")
        rows[:, 0] = d
        rows[:, 1:] = cb.view("<i4")
        rows.tofile(f)


def run(comp, shapes, users, seed, steps, **kw):
    base = dict(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=0, ef=False, two_phase=False, scale="0.5",
                num_users=users, mode="ps", cr=256)
    base.update(kw)
    params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
    with contextlib.redirect_stdout(io.StringIO()):
        q = Quantizer(comp, params, Namespace(**base))
    g = torch.Generator(device="cuda").manual_seed(seed)
    for _ in range(steps):
        for u in range(users):
            for p in params:
                p.grad = torch.randn(p.shape, device="cuda", generator=g) * 1e-2
            q.record(u, epoch=1)
        q.apply()
    return q, [p.grad.data.clone() for p in params]


t0, rounds = time.time(), 0
while time.time() - t0 < budget:
    rounds += 1
    hsq = rng.random() < 0.6
    users, steps = int(rng.integers(1, 4)), int(rng.integers(1, 3)) + (3 if rng.random() < 0.2 else 0)
    if rng.random() < 0.15:      # every decode-mean kernel: compile-time R up to 16, the chunked form above
        users = int(rng.choice([4, 5, 7, 8, 9, 12, 16, 17, 21]))
    extra = [{}, {"ef": True}, {"ef": True, "two_phase": True}, {"two_phase": True}, {"mode": "ring"},
             {"mode": "ring", "ef": True}][int(rng.integers(0, 6))]
    if hsq:
        d = int(rng.choice([8, 16, 32, 12, 10, 24, 48, 64, 100]))
        k_bit = int(rng.choice([5, 6, 8, 8, 8, 9, 10])) if d in (8, 16, 32) else int(rng.choice([5, 7, 8, 9]))
        n_bit = int(rng.choice([4, 6, 8, 9, 17, 32]))
        if 2 ** k_bit == d:      # K == d: a random orthogonal codebook per compressor instance, nothing to compare
            continue
        codebook(d, 2 ** k_bit)
        kw = dict(c_dim=d, k_bit=k_bit, n_bit=n_bit)
        if d == 16 and rng.random() < 0.5:      # GQ_LEVELS_PACKED6 wherever the configuration allows it (else it falls back to bytes)
            kw.update(gq_wire_levels="packed6")
            if rng.random() < 0.7:
                kw.update(k_bit=8, n_bit=int(rng.choice([2, 5, 6])))
                codebook(16, 256)
        comp = NearestNeighborCompressor
    else:
        d = int(rng.choice([0, 128, 64, 512, 4098, 8192, 2, 8, 16, 32, 256, 1024, 2048]))
        kw = dict(c_dim=d, n_bit=int(rng.choice([1, 2, 4, 6, 8, 9])))
        comp = QSGDCompressor
        d = d or 2
    shapes = []
    for _ in range(int(rng.integers(2, 9))):
        m = int(rng.choice([1, 3, 63, 64, 65, 130, 1000, 5000, 40000])) * int(rng.integers(1, 4))
        shapes.append((m * d,) if rng.random() < 0.5 else (m, d))
    shapes.append((10,))
    kw.update(extra)
    if rng.random() < 0.3 and kw.get("mode", "ps") == "ps":     # the batched quantizer replays recurring steps from HIP graphs (the reference path stays eager)
        kw.update(gq_graph=True)
    total = sum(int(np.prod(sh)) for sh in shapes)
    if total * 4 * (users + 8) * 2 > 120e9:      # both quantizers alive at once: parameters, gradients, residuals per user, two
        continue                                  # output buffers, clones -- a round of 4 GB tensors does not fit the 288 GB
    try:
        qb, gb = run(comp, shapes, users, rounds, steps, **kw)
        qp, gp = run(comp, shapes, users, rounds, steps, gq_no_batch=True, **kw)
    except AssertionError as e:
        if "not divisible" in str(e):
            continue
        raise
    ok = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(gb, gp)) and torch.equal(qb._wire, qp._wire)
    if kw.get("ef"):
        for pb, pp in zip(qb.parameters, qp.parameters):
            ok = ok and all(torch.equal(x, y) for x, y in zip(pb.error, pp.error))
            if kw.get("two_phase"):
                ok = ok and torch.equal(pb.server_error, pp.server_error)
    groups = [(g[0].__name__, len(g[1])) for g in qb._groups]
    del qb, qp, gb, gp
    gc.collect()                 # codec <-> single-tensor group cycles hold device buffers until collected
    torch.cuda.empty_cache()
    if not ok:
        print("MISMATCH", comp.__name__, kw, shapes, users, steps)
        sys.exit(1)
print("fuzz_batched: %d rounds, 0 mismatches (batched groups used in the last round: %s)"
      % (rounds, groups))
