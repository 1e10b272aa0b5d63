"""Randomised cross-check of the prefilter encode (impl 4) against the exact f32 MFMA kernels on adversarial
inputs: random sizes and scales, sparse / one-hot / integer-valued / duplicated subvectors, exact codeword
multiples and sums (ties), subnormal and huge magnitudes.  Bit-equality of codes and u is required.
    python tools/fuzz_prefilter.py [rounds] [seed] [smallk]
smallk (round 6): K drawn from 4 ... 256 in multiples of 4 (the first K rows of the codebook: --k-bit 5 / 6 and K == dim run the
kernels that score one / two row blocks, every other K the eight-block kernel over zero rows), a few non-finite values sprinkled
in, and the codes compared on EVERY subvector (rows behind the codebook must never win, whatever 0 x inf gives).    """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import numpy as np, torch
from gq_amd import native
from gq_amd.codebook import load_codebook
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
bad = 0
SMALLK = len(sys.argv) > 3 and sys.argv[3] == "smallk"
for it in range(rounds):
    d = int(rng.choice([16, 16, 8, 32, 12, 24]))      # (12 / 24: the repaired dimensions on the padded kernels; seeds of rounds <= 5 drew from [16, 16, 8, 32])
    cbn = load_codebook(d, 256)
    K = int(rng.choice([4, 8, 16, 28, 32, 36, 60, 64, 68, 100, 128, 200, 252, 256])) if SMALLK else 256
    cbn = np.ascontiguousarray(cbn[:K])
    if rng.rand() < 0.3:
        cbn = (cbn * rng.uniform(0.1, 20.0, (K, 1))).astype(np.float32)      # unnormalised rows
    M = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 50000, 200000]))
    kind = rng.choice(["randn", "sparse", "onehot", "ints", "dup", "codeword", "sum2", "tiny", "huge", "mixed"])
    scale = 10.0 ** rng.uniform(-6, 4)
    if kind == "randn":
        x = rng.standard_normal((M, d))
    elif kind == "sparse":
        x = rng.standard_normal((M, d)) * (rng.rand(M, d) < 0.15)
    elif kind == "onehot":
        x = np.zeros((M, d)); x[np.arange(M), rng.randint(0, d, M)] = rng.choice([-1.0, 1.0, 0.5, 3.0], M)
    elif kind == "ints":
        x = rng.randint(-3, 4, (M, d)).astype(np.float64)
    elif kind == "dup":
        x = np.tile(rng.standard_normal((max(1, M // 50 + 1), d)), (50, 1))[:M]
    elif kind == "codeword":
        x = cbn[rng.randint(0, K, M)] * rng.choice([-2.0, -1.0, 1.0, 0.25], (M, 1))
    elif kind == "sum2":
        x = cbn[rng.randint(0, K, M)] + rng.choice([-1.0, 1.0], (M, 1)) * cbn[rng.randint(0, K, M)]
    elif kind == "tiny":
        x = rng.standard_normal((M, d)); scale = 10.0 ** rng.uniform(-44, -30)
    elif kind == "huge":
        x = rng.standard_normal((M, d)); scale = 10.0 ** rng.uniform(25, 36)
    else:
        x = rng.standard_normal((M, d)) * np.exp(rng.standard_normal((M, d)) * 6)
    x = np.ascontiguousarray((x * scale).astype(np.float32).reshape(-1))
    if SMALLK and rng.rand() < 0.5:
        for v in (np.inf, -np.inf, np.nan):
            x[rng.randint(0, x.size, max(1, M // 200))] = v
    g = torch.from_numpy(x).to(dev)
    cb = torch.from_numpy(np.ascontiguousarray(cbn)).to(dev)
    out = {}
    exact = 1 if (d == 16 and K == 256) else 5
    for impl in (4, exact):
        codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
        ws = native.new_workspace(dev, M)
        native.hsq_encode(g, cb, codes, u, ws, impl=impl)
        torch.cuda.synchronize()
        out[impl] = (codes, u, ws[:2].clone())
    (ca, ua, wa), (cb_, ub, wb) = out[4], out[exact]
    finite = torch.isfinite(ub)
    same = torch.equal(ca[finite], cb_[finite]) and torch.equal(ua[finite].view(torch.int32), ub[finite].view(torch.int32))
    if SMALLK:
        same = same and torch.equal(ca, cb_) and torch.equal(torch.isnan(ua), torch.isnan(ub)) and int(ca.max()) < K
    if not same:
        bad += 1
        n = int(((ca != cb_) | (ua.view(torch.int32) != ub.view(torch.int32)))[finite].sum())
        print("MISMATCH round %d: d=%d K=%d M=%d kind=%s scale=%.3g: %d subvectors differ" % (it, d, K, M, kind, scale, n))
print("%d rounds, %d mismatching" % (rounds, bad))
sys.exit(1 if bad else 0)
