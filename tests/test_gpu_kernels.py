"""GPU parity tests: the HIP kernels (through the C ABI, include/gq_hsq.h) against
(1) the golden vectors captured from the reference and (2) the CPU oracle on seeded
inputs.  Bit-exact for codes / levels / u / lb / ub and for the single-payload decode;
the R-payload mean is bit-exact too (same summation order), asserted at 0 ulp."""
import contextlib
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def _capture(graph):
    """torch.cuda.graph(graph) with the garbage collector held off: a CUDAGraph of an earlier test that the collector finalizes
    INSIDE a capture raises in its destructor and takes the process down (see gq_amd.quantizers._capturing)."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was:
            gc.enable()

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
HSQ_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "hsq_*.npz")))
QSGD_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "qsgd_*.npz")))


def _cb(d, K):
    return np.load(os.path.join(GOLDEN, "codebook_d%d_k%d_normalized.npy" % (d, K)))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same(a, b):
    """Bitwise equal, except that any NaN equals any NaN (sign / payload of a NaN are not part of the contract:
    x86 generates -qNaN, gfx950 +qNaN)."""
    a, b = np.ascontiguousarray(a, np.float32).reshape(-1), np.ascontiguousarray(b, np.float32).reshape(-1)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


@pytest.fixture(scope="module")
def nat():
    from gq_amd import native
    native.lib()
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return native


def gpu_compress(nat, x, cb, n_bit, random, r=None, impl=0, code_dtype=None, level_dtype=torch.int32, seed=0):
    dev = torch.device("cuda:0")
    K, d = cb.shape
    g = torch.from_numpy(np.ascontiguousarray(x, np.float32).reshape(-1)).to(dev)
    c = torch.from_numpy(cb).to(dev)
    M = g.numel() // d
    if code_dtype is None:
        code_dtype = torch.uint8 if K <= 256 else torch.int32
    codes = torch.empty(M, dtype=code_dtype, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    partials = nat.new_workspace(dev, M)
    nat.mark_worklist(partials, M)
    nat.hsq_encode(g, c, codes, u, partials, impl=impl)
    out = dict(codes=codes, u=u, cb=c, M=M, ws=partials)
    if n_bit != 32:
        lb_ub = torch.empty(2, dtype=torch.float32, device=dev)
        levels = torch.empty(M, dtype=level_dtype, device=dev)
        rt = torch.from_numpy(np.ascontiguousarray(r, np.float32)).to(dev) if random == 1 else None
        nat.hsq_levels(u, n_bit, random, rt, seed, partials, lb_ub, levels)
        out.update(lb_ub=lb_ub, levels=levels)
    torch.cuda.synchronize()
    return out


def gpu_decode(nat, res, n_bit):
    dev = res["codes"].device
    K, d = res["cb"].shape
    out = torch.empty(res["M"] * d, dtype=torch.float32, device=dev)
    if n_bit == 32:
        nat.hsq_decode_sum(res["codes"], res["u"], None, res["cb"], 32, out, R=1)
    else:
        nat.hsq_decode_sum(res["codes"], res["levels"], res["lb_ub"], res["cb"], n_bit, out, R=1)
    torch.cuda.synchronize()
    return out.cpu().numpy()


IMPLS = {"auto": 0, "mfma_exact_d16k256": 1, "mfma_generic": 2, "valu": 3, "prefilter_d16k256": 4, "mfma_lds": 5}


@pytest.mark.parametrize("impl", ["auto", "mfma_exact_d16k256", "mfma_generic", "valu", "prefilter_d16k256", "mfma_lds"])
@pytest.mark.parametrize("name", HSQ_CASES)
def test_hsq_matches_reference_golden(nat, name, impl):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    d, K, n_bit, random = int(g["dim"]), int(g["K"]), int(g["n_bit"]), int(g["random"])
    if impl == "valu" and (d not in (8, 12, 16, 24, 32) or K * d * 4 > 65536):
        pytest.skip("valu cross-check kernel not built for this shape")
    if impl == "mfma_exact_d16k256" and (d, K) != (16, 256):
        pytest.skip("d16/K256 specialisation")
    if impl == "prefilter_d16k256" and not (K <= 256 and K % 4 == 0 and d in (8, 12, 16, 24, 32)):
        pytest.skip("the prefilter kernels are built for K <= 256 (a multiple of 4) and d in {8, 16, 32} (12 / 24, the repaired dimensions, as padded 16 / 32)")
    cb = _cb(d, K)
    r = g["r"] if random else None
    res = gpu_compress(nat, g["x"], cb, n_bit, random, r, impl=IMPLS[impl])
    codes = res["codes"].cpu().numpy().astype(np.int32)
    u = res["u"].cpu().numpy()
    assert np.array_equal(codes, g["codes"].astype(np.int32)), "codes differ from the reference"
    if codes.size == 1:  # M == 1: reference runs MKL sgemv (see test_oracle_golden.py)
        assert abs(int(_bits(u)[0]) - int(_bits(g["u"])[0])) <= 2
        return
    assert _same(u, g["u"]), "u differs bitwise from the reference"
    if n_bit != 32:
        lb_ub = res["lb_ub"].cpu().numpy()
        assert _same(lb_ub[0], g["lb"]) and _same(lb_ub[1], g["ub"])
        assert np.array_equal(res["levels"].cpu().numpy(), g["levels"])
    dec = gpu_decode(nat, res, n_bit)
    assert _same(dec, g["decoded"]), "decoded differs bitwise"


def test_lds_staged_encode_chunked_codebook_in_a_small_lds_budget():
    """The same shapes with the kernel's LDS budget forced down to 56 KiB (GQ_LDS_LIMIT is read once
    per process, hence the child process): every codebook is staged in several chunks and the results
    still equal the generic kernel's and the oracle's."""
    import subprocess
    import sys
    env = dict(os.environ, GQ_LDS_LIMIT="57344")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", __file__, "-k",
                        "any_shape and not 100-512 and not 96-256 and not 65-4096 and not 64-1024"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("d,K,M", [(16, 4096, 5000), (64, 1024, 1000), (100, 512, 777), (96, 256, 300), (5, 5, 1234),
                                   (3, 32, 64), (33, 2048, 129), (1, 64, 4097), (65, 4096, 200), (128, 256, 3000), (120, 300, 700),
                                   (112, 64, 129)])
def test_lds_staged_encode_any_shape_matches_generic_and_oracle(nat, oracle, d, K, M):
    """The LDS-staged exact MFMA kernel (the default for everything but d16/K256): whole and chunked
    codebooks, ragged tiles, K that is no multiple of 32, odd d, d up to 128 (two-wave workgroups above ~110: four
    tiles do not fit the LDS there) -- bit-identical to the generic kernel
    and to the oracle's fmaf chain (the reference's torch.mm arithmetic)."""
    rng = np.random.RandomState(d * 1000 + K)
    cb = rng.standard_normal((K, d)).astype(np.float32)
    cb /= np.maximum(np.linalg.norm(cb, axis=1, keepdims=True), 1e-20)
    x = (rng.standard_normal(M * d) * 0.1).astype(np.float32)
    x[:d] = 0.0                       # an all-zero subvector -> code 0, u = +0
    got = gpu_compress(nat, x, cb, 32, 0, impl=5)
    gen = gpu_compress(nat, x, cb, 32, 0, impl=2)
    auto = gpu_compress(nat, x, cb, 32, 0, impl=0)
    for other in (gen, auto):
        assert torch.equal(got["codes"], other["codes"])
        assert torch.equal(got["u"].view(torch.int32), other["u"].view(torch.int32))
    codes, u = oracle.hsq_encode(x, cb)
    assert np.array_equal(got["codes"].cpu().numpy().astype(np.int64), codes.astype(np.int64))
    assert np.array_equal(_bits(got["u"].cpu().numpy()), _bits(u))


@pytest.mark.parametrize("d,K,M", [(16, 512, 100003), (16, 768, 4097), (16, 1024, 65), (16, 4096, 20000),
                                   (8, 512, 50001), (8, 2048, 4097), (32, 512, 30001), (32, 1024, 65)])
def test_paged_prefilter_encode_larger_codebooks(nat, oracle, d, K, M):
    """d in {8, 16, 32}, K = 256 * pages (--k-bit 9 ... 12): the prefilter kernel once per page of 256 codewords, the pages'
    exact winners merged in place (an earlier page keeps a tie).  Codes, u, lb, ub and levels bit-identical to the
    exact LDS kernel and the oracle, including exact ties ACROSS pages."""
    rng = np.random.RandomState(K + M)
    cb = rng.standard_normal((K, d)).astype(np.float32)
    cb /= np.linalg.norm(cb, axis=1, keepdims=True)
    cb[300] = cb[10]            # the same codeword on two pages: the first one wins
    cb[K - 1] = -cb[20]         # |score| tie with opposite signs
    cb[256 + 7] = cb[256 + 200] # and a tie inside a later page
    x = (rng.standard_normal(M * d) * 0.05).astype(np.float32)
    x[0:d] = 0.0
    x[d:2 * d] = cb[10] * 0.5
    x[2 * d:3 * d] = cb[20] * -0.25
    x[3 * d:4 * d] = cb[256 + 7] * 3.0
    got = gpu_compress(nat, x, cb, 6, 0, impl=0)
    ref = gpu_compress(nat, x, cb, 6, 0, impl=5)
    assert torch.equal(got["codes"], ref["codes"]) and torch.equal(got["u"].view(torch.int32), ref["u"].view(torch.int32))
    assert torch.equal(got["lb_ub"].view(torch.int32), ref["lb_ub"].view(torch.int32))
    assert torch.equal(got["levels"], ref["levels"])
    codes, u = oracle.hsq_encode(x, cb)
    assert np.array_equal(got["codes"].cpu().numpy().astype(np.int64), codes.astype(np.int64))
    assert np.array_equal(_bits(got["u"].cpu().numpy()), _bits(u))
    c = got["codes"].cpu().numpy()
    assert c[0] == 0 and c[1] == 10 and c[2] == 20 and c[3] == 256 + 7


@pytest.mark.parametrize("scale", [1.0, 1e-3])
@pytest.mark.parametrize("M", [1, 2, 31, 32, 33, 63, 64, 65, 127, 4095, 4096, 4097, 100003])
def test_hsq_encode_vs_oracle_ragged(nat, oracle, M, scale):
    """Ragged subvector counts around the 64-subvector tile and the grid size."""
    rng = np.random.RandomState(1000 + M)
    cb = _cb(16, 256)
    x = (rng.standard_normal(M * 16) * scale).astype(np.float32)
    ref = oracle.hsq_compress(x, cb, 6, 0)
    res = gpu_compress(nat, x, cb, 6, 0)
    assert np.array_equal(res["codes"].cpu().numpy().astype(np.int32), ref["codes"])
    assert np.array_equal(_bits(res["u"].cpu().numpy()), _bits(ref["u"]))
    lb_ub = res["lb_ub"].cpu().numpy()
    assert _bits(lb_ub[0]) == _bits(ref["lb"]) and _bits(lb_ub[1]) == _bits(ref["ub"])
    assert np.array_equal(res["levels"].cpu().numpy(), ref["levels"])


@pytest.mark.parametrize("level_dtype", [torch.uint8, torch.int16, torch.int32])
def test_hsq_level_widths_and_given_r(nat, oracle, level_dtype):
    rng = np.random.RandomState(5)
    cb = _cb(16, 256)
    x = rng.standard_normal(16 * 5000).astype(np.float32)
    r = rng.random_sample(5000).astype(np.float32)
    ref = oracle.hsq_compress(x, cb, 6, 1, r)
    res = gpu_compress(nat, x, cb, 6, 1, r, level_dtype=level_dtype)
    assert np.array_equal(res["levels"].cpu().numpy().astype(np.int32), ref["levels"])
    assert ref["levels"].max() == 64  # the top element always rounds up
    dec = gpu_decode(nat, res, 6)
    assert np.array_equal(_bits(dec), _bits(oracle.hsq_decompress(ref["codes"], ref["levels"], ref["lb"], ref["ub"], cb, 6)))


def test_hsq_device_rng_is_stochastic_rounding(nat, oracle):
    """GQ_RANDOM_DEVICE: levels are floor or floor+1 of x, and unbiased on average."""
    rng = np.random.RandomState(6)
    cb = _cb(16, 256)
    M = 200000
    x = rng.standard_normal(16 * M).astype(np.float32)
    det = oracle.hsq_compress(x, cb, 6, 0)
    res = gpu_compress(nat, x, cb, 6, 2, seed=1234)
    lv = res["levels"].cpu().numpy().astype(np.int64)
    diff = lv - det["levels"]
    assert diff.min() >= 0 and diff.max() <= 1
    xs = np.abs((det["u"].astype(np.float64) - det["lb"]) / (np.float64(det["ub"]) - det["lb"])) * 64
    frac = xs - np.floor(np.minimum(xs, 63))
    assert abs(diff.mean() - frac.mean()) < 5e-3
    res2 = gpu_compress(nat, x, cb, 6, 2, seed=1234)
    assert torch.equal(res["levels"], res2["levels"])          # reproducible per seed
    res3 = gpu_compress(nat, x, cb, 6, 2, seed=99)
    assert not torch.equal(res["levels"], res3["levels"])


@pytest.mark.parametrize("R,M", [(1, 3000), (2, 3000), (3, 3000), (8, 3000), (5, 3001), (7, 2999), (9, 3000), (16, 3002),
                                 (19, 3003), (2, 1_100_003), (8, 600_001), (11, 600_002)])
def test_hsq_decode_sum_matches_oracle_mean(nat, oracle, R, M, K=256):
    """R <= 16: the compile-time-R pipelined kernels (a team's lanes fetch different payloads' words); above: the chunked
    one; M % 4 != 0: the partial last group; the large M give every lane several items (the words of the next item are
    requested while the current one is summed)."""
    rng = np.random.RandomState(40 + R)
    cb = np.ascontiguousarray(_cb(16, 256)[:K])     # (any unit-norm rows do: the oracle takes the same codebook)
    dev = torch.device("cuda:0")
    codes, levels, lbub, decs = [], [], [], []
    for r in range(R):
        x = (rng.standard_normal(16 * M) * (0.5 + r)).astype(np.float32)
        c = oracle.hsq_compress(x, cb, 6, 0)
        codes.append(c["codes"].astype(np.uint8))
        levels.append(c["levels"].astype(np.uint8))
        lbub.append([c["lb"], c["ub"]])
        decs.append(oracle.hsq_decompress(c["codes"], c["levels"], c["lb"], c["ub"], cb, 6))
    ref = oracle.mean_users(np.stack(decs, 0))
    out = torch.empty(M * 16, dtype=torch.float32, device=dev)
    nat.hsq_decode_sum(torch.from_numpy(np.stack(codes)).to(dev), torch.from_numpy(np.stack(levels)).to(dev),
                       torch.tensor(lbub, dtype=torch.float32, device=dev), torch.from_numpy(cb).to(dev), 6, out, R=R)
    got = out.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(ref))


@pytest.mark.parametrize("R", list(range(2, 20)))
def test_fma_aggregate_is_within_1e6_of_the_oracle_mean_and_opt_in(nat, oracle, R, M=40_001):
    """GQ_AGGREGATE_FMA (OR-ed into n_bit; the quantizer: $GQ_AGGREGATE=fma): the decode-mean over R payloads accumulates with
    fused multiply-adds.  Tolerance, written here: relative L2 <= 1e-6 against the oracle's bit-exact mean (the north star
    grants 1e-5 on the decoded aggregate).  Built for R = 2, 4, 8, 16; every other R ignores the flag and stays bit-exact; the
    default (no flag) is bit-exact for every R."""
    rng = np.random.RandomState(140 + R)
    cb = _cb(16, 256)
    dev = torch.device("cuda:0")
    codes, levels, lbub, decs = [], [], [], []
    for r in range(R):
        x = (rng.standard_normal(16 * M) * (0.5 + r)).astype(np.float32)
        c = oracle.hsq_compress(x, cb, 6, 0)
        codes.append(c["codes"].astype(np.uint8))
        levels.append(c["levels"].astype(np.uint8))
        lbub.append([c["lb"], c["ub"]])
        decs.append(oracle.hsq_decompress(c["codes"], c["levels"], c["lb"], c["ub"], cb, 6))
    ref = oracle.mean_users(np.stack(decs, 0))
    args = (torch.from_numpy(np.stack(codes)).to(dev), torch.from_numpy(np.stack(levels)).to(dev),
            torch.tensor(lbub, dtype=torch.float32, device=dev), torch.from_numpy(cb).to(dev))
    exact, fused = torch.empty(M * 16, dtype=torch.float32, device=dev), torch.empty(M * 16, dtype=torch.float32, device=dev)
    nat.hsq_decode_sum(*args, 6, exact, R=R)
    nat.hsq_decode_sum(*args, 6 | nat.AGGREGATE_FMA, fused, R=R)
    assert np.array_equal(_bits(exact.cpu().numpy()), _bits(ref))
    got = fused.cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert rel <= 1e-6, rel
    if R not in (2, 4, 8, 16):
        assert np.array_equal(_bits(fused.cpu().numpy()), _bits(ref))


@pytest.mark.parametrize("R", [3, 5, 6, 7, 9, 12, 13, 15, 17, 19, 33])
@pytest.mark.parametrize("span", ["subnormal", "huge", "mixed", "inf"])
def test_hsq_decode_mean_by_any_user_count_at_the_ends_of_the_float_range(nat, oracle, R, span):
    """ps_quantizer.py:48 divides the sum by R.  For an odd R the kernels use a four-operation quotient by the constant
    instead of the IEEE sequence (csrc/gq_common.hpp: exact for every input, tools/div_check.hip); an even R that is not a
    power of two divides.  Payloads whose norms make the sums subnormal, overflow to +-inf, or NaN, against the oracle's
    true division, bit for bit (any NaN equals any NaN)."""
    M = 2051
    rng = np.random.RandomState(700 + R)
    cb = _cb(16, 256)
    dev = torch.device("cuda:0")
    codes = rng.randint(0, 256, (R, M)).astype(np.uint8)
    levels = rng.randint(0, 65, (R, M)).astype(np.uint8)
    if span == "subnormal":
        lb = -rng.uniform(0, 4e-39, R)
        ub = rng.uniform(0, 4e-39, R)
    elif span == "huge":      # every payload finite (l * (ub - lb) < FLT_MAX for l <= 64); sums of many of them are not
        lb = rng.choice([-1.0, 1.0], R) * rng.uniform(1.0e38, 1.65e38, R)
        ub = lb + rng.uniform(0, 5e36, R)
    elif span == "mixed":
        lb = -10.0 ** rng.uniform(-44, 38, R)
        ub = 10.0 ** rng.uniform(-44, 38, R)
    else:                     # one payload's norms are +inf (level 0: 0 * inf = NaN): sums of +-inf and NaN
        lb = -rng.uniform(0, 1, R)
        ub = rng.uniform(0, 1, R)
        ub[R - 1] = np.inf
    lbub = np.stack([lb, ub], 1).astype(np.float32)
    with np.errstate(all="ignore"):
        decs = [oracle.hsq_decompress(codes[r], levels[r].astype(np.int32), lbub[r, 0], lbub[r, 1], cb, 6) for r in range(R)]
        ref = oracle.mean_users(np.stack(decs, 0))
    out = torch.empty(M * 16, dtype=torch.float32, device=dev)
    nat.hsq_decode_sum(torch.from_numpy(codes).to(dev), torch.from_numpy(levels).to(dev),
                       torch.from_numpy(lbub).to(dev), torch.from_numpy(cb).to(dev), 6, out, R=R)
    got = out.cpu().numpy()
    assert _same(got, ref)
    if span == "subnormal":
        assert (got != 0).any() and np.abs(got).max() < 1.2e-38
    if span == "inf":
        assert np.isinf(got).any() and np.isnan(got).any()


@pytest.mark.parametrize("R", [1, 2, 3, 5, 6, 7, 12, 19, 64, 257])
def test_mean_rows_divides_like_the_reference_mean(nat, R):
    """gq_mean_rows (the dense side channel's aggregate, ps_quantizer.py:48 for the tensors IdenticalCompressor passes through):
    (+0 + row 0 + row 1 + ...) / R with rows ascending; for an odd R the kernel's four-operation quotient must equal the true
    division on subnormal, huge, infinite and NaN sums as well."""
    n = 4099
    rng = np.random.RandomState(900 + R)
    rows = (rng.standard_normal((R, n)) * 10.0 ** rng.uniform(-44, 37, (1, n))).astype(np.float32)
    rows[:, 0] = -0.0
    rows[0, 1], rows[R - 1, 2] = np.inf, np.nan
    rows[:, 3] = 3.0e38
    with np.errstate(all="ignore"):
        acc = np.zeros(n, np.float32)
        for r in range(R):
            acc = acc + rows[r]
        ref = acc / np.float32(R)
    dev = torch.device("cuda:0")
    out = torch.empty(n, dtype=torch.float32, device=dev)
    nat.mean_rows(torch.from_numpy(rows).to(dev), out)
    got = out.cpu().numpy()
    assert _same(got, ref)
    assert _bits(got[:1])[0] == 0                      # the -0 column comes out as +0
    if R >= 2:
        assert np.isinf(got[3])


@pytest.mark.parametrize("name", QSGD_CASES)
def test_qsgd_matches_reference_golden(nat, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    d, n_bit, random = int(g["dim"]), int(g["n_bit"]), int(g["random"])
    dev = torch.device("cuda:0")
    x = torch.from_numpy(g["x"].reshape(-1)).to(dev)
    Mb = x.numel() // d
    norm = torch.empty(Mb, dtype=torch.float32, device=dev)
    signs = torch.empty(Mb * d, dtype=torch.uint8, device=dev)
    levels = torch.empty(Mb * d, dtype=torch.int32, device=dev)
    r = torch.from_numpy(g["r"].reshape(-1)).to(dev) if random else None
    nat.qsgd_compress(x, d, n_bit, 1 if random else 0, r, 0, norm, signs, levels)
    torch.cuda.synchronize()
    assert _same(norm.cpu().numpy(), g["norm"])          # a NaN element makes its bucket's norm NaN (torch.max)
    assert np.array_equal(signs.cpu().numpy().astype(bool), g["signs"].reshape(-1))
    assert np.array_equal(levels.cpu().numpy(), g["levels"].reshape(-1))
    out = torch.empty(Mb * d, dtype=torch.float32, device=dev)
    nat.qsgd_decode_sum(norm, signs, levels, d, n_bit, out, R=1)
    assert np.array_equal(out.cpu().numpy(), g["decoded"].reshape(-1), equal_nan=True)


def test_full_size_properties(nat, oracle):
    """BASELINE size (25M floats): size-independent properties + a sampled oracle check."""
    dev = torch.device("cuda:0")
    cb_np = _cb(16, 256)
    cb = torch.from_numpy(cb_np).to(dev)
    torch.manual_seed(1234)
    g = torch.randn(25_000_000, device=dev)
    M = g.numel() // 16
    codes = torch.empty(M, dtype=torch.uint8, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    partials = nat.new_workspace(dev, M)
    lb_ub = torch.empty(2, dtype=torch.float32, device=dev)
    levels = torch.empty(M, dtype=torch.uint8, device=dev)
    nat.hsq_encode(g, cb, codes, u, partials)
    nat.hsq_levels(u, 6, 0, None, 0, partials, lb_ub, levels)
    out = torch.empty_like(g)
    nat.hsq_decode_sum(codes, levels, lb_ub, cb, 6, out, R=1)
    torch.cuda.synchronize()
    # lb/ub are the exact extrema of u
    assert lb_ub[0].item() == u.min().item() and lb_ub[1].item() == u.max().item()
    assert int(levels.max()) == 63 and int(levels.min()) == 0
    # u is the projection on the chosen codeword, and no codeword projects larger (fp32 matmul check)
    V = g.view(-1, 16)
    sel = cb[codes.long()]
    proj = (V * sel).sum(1)
    assert torch.allclose(proj, u, rtol=1e-4, atol=1e-5)
    idx = torch.arange(0, M, 97, device=dev)
    best = (V[idx] @ cb.t()).abs().max(1).values
    assert torch.all(u[idx].abs() >= best * (1 - 1e-5))
    # decoded = codeword * dequantised level; the residual is smaller than the input
    assert (g - out).norm() < g.norm()
    # idempotence of the decode and linearity in R: mean of two identical payloads == one payload
    out2 = torch.empty_like(g)
    nat.hsq_decode_sum(torch.cat([codes, codes]), torch.cat([levels, levels]), torch.cat([lb_ub, lb_ub]), cb, 6,
                       out2, R=2)
    assert torch.equal(out, out2)
    # sampled bit-exact check against the oracle on a 1M-element window (offset not tile aligned)
    off = 16 * 123457
    win = g[off:off + 16 * 65536].cpu().numpy()
    rc, ru = oracle.hsq_encode(win, cb_np)
    assert np.array_equal(codes[123457:123457 + 65536].cpu().numpy().astype(np.int32), rc)
    assert np.array_equal(_bits(u[123457:123457 + 65536].cpu().numpy()), _bits(ru))


@pytest.mark.parametrize("d,K,M", [(16, 256, 100003), (8, 32, 5000), (12, 100, 777), (32, 512, 3000), (48, 2048, 500),
                                   (4, 16, 64), (10, 33, 4097), (16, 256, 1)])
def test_pvq_encode_on_the_matrix_cores_matches_the_oracle(nat, oracle, d, K, M):
    """ProbabilisticVectorCompressor encode (intended semantics; parity with the reference unpinned): the MFMA
    kernel (any d <= 104, any K; whole and chunked codebooks, K no multiple of 32, ragged tiles) against the CPU
    restatement for the same draws r -- codes, u and the (min,max) of u bit for bit; and against the VALU
    cross-check kernel where that one is built."""
    rng = np.random.RandomState(d * 131 + K)
    cdag = rng.standard_normal((K, d)).astype(np.float32) * 0.3
    x = (rng.standard_normal(M * d) * 0.05).astype(np.float32)
    if M > 2:
        x[d:2 * d] = 0.0                      # an all-zero subvector: l1 = 0, NaN walk -> code K-1, u = 0
    r = rng.random_sample(M).astype(np.float32)
    r[0] = 0.0
    r[-1] = np.float32(1.0) - np.float32(2.0 ** -24)
    dev = torch.device("cuda:0")
    g, c, rt = torch.from_numpy(x).to(dev), torch.from_numpy(cdag).to(dev), torch.from_numpy(r).to(dev)
    codes = torch.empty(M, dtype=torch.uint8 if K <= 256 else torch.int32, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = nat.new_workspace(dev, M)
    nat.pvq_encode(g, c, codes, u, ws, nat.RANDOM_GIVEN, rt, 0)
    torch.cuda.synchronize()
    oc, ou = oracle.pvq_encode(x, cdag, r)
    assert np.array_equal(codes.cpu().numpy().astype(np.int64), oc.astype(np.int64))
    assert np.array_equal(_bits(u.cpu().numpy()), _bits(ou))
    lb_ub = torch.empty(2, dtype=torch.float32, device=dev)
    levels = torch.empty(M, dtype=torch.uint8, device=dev)
    nat.hsq_levels(u, 6, 0, None, 0, ws, lb_ub, levels)
    assert lb_ub[0].item() == float(ou.min()) and lb_ub[1].item() == float(ou.max())


@pytest.mark.parametrize("d,K,M", [(16, 256, 64 * 900 + 17), (32, 256, 20011), (8, 256, 30001), (16, 96, 9000), (32, 32, 4097),
                                   (8, 224, 5003)])
def test_pvq_one_sweep_kernel_matches_the_oracle(nat, oracle, d, K, M):
    """The one-sweep kernel (pvq.hip: pvq_encode_walk_kernel; d in {8, 16, 32}, K = 32 ... 256 in whole blocks): codes and u bit
    for bit against the CPU restatement, on data with whole zero subvectors, subvectors with one element, scales on both
    sides of the range its three-operation quotient accepts (those lanes go through the wave's exact prefix sum or the
    term-by-term walk), and draws on the edges."""
    rng = np.random.RandomState(7 * d + K)
    cdag = rng.standard_normal((K, d)).astype(np.float32) * 0.3
    x = (rng.standard_normal((M, d)) * 0.05).astype(np.float32)
    x[5::97] = 0.0
    x[7::89, 1:] = 0.0
    x[11::83] *= np.float32(1e-28)
    x[13::79] *= np.float32(1e9)
    x[17::101, 2:] *= np.float32(1e-12)        # quotients far below 2^-29 next to ordinary ones
    r = rng.random_sample(M).astype(np.float32)
    r[:4] = [0.0, np.float32(1e-5), np.float32(1.0) - np.float32(2.0 ** -24), np.float32(0.5)]
    x = x.reshape(-1)
    dev = torch.device("cuda:0")
    g, c, rt = torch.from_numpy(x).to(dev), torch.from_numpy(cdag).to(dev), torch.from_numpy(r).to(dev)
    codes = torch.empty(M, dtype=torch.uint8, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = nat.new_workspace(dev, M)
    nat.pvq_encode(g, c, codes, u, ws, nat.RANDOM_GIVEN, rt, 0)
    torch.cuda.synchronize()
    oc, ou = oracle.pvq_encode(x, cdag, r)
    assert np.array_equal(codes.cpu().numpy().astype(np.int64), oc.astype(np.int64))
    assert np.array_equal(_bits(u.cpu().numpy()), _bits(ou))


@pytest.mark.parametrize("env", [{"GQ_PVQ_EPS": "1e-3"}, {"GQ_PVQ_EPS": "-1e-3"}, {"GQ_PVQ_TWO_SWEEPS": "1"}])
def test_pvq_one_sweep_kernel_rare_paths(env):
    """The same PVQ / residual tests with the one-sweep kernel's window widened to 1e-3 (read once per process, hence the child):
    about half of the lanes are then unsettled and go through the wave's prefix-sum walk (GQ_PVQ_EPS > 0) or the
    term-by-term walk (< 0); and with the two-sweep kernel in its place (GQ_PVQ_TWO_SWEEPS, the cross-check)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", __file__,
                        os.path.join(here, "test_gpu_api.py"), "-k", "(pvq or residual or vector) and not rare_paths"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("K", [256, 250])
def test_pvq_shared_quotient_walk_equals_the_division(nat, oracle, K):
    """The inverse-CDF walk divides every |p_k| by the same l1: the kernel's three-operation quotient (pvq.hip,
    shared_quotient) and its double threshold have to give what `a / l1` and `(float)cum >= thr` give.  Subvectors
    scaled from 1e-30 to 1e8 (both sides of the range the fast form accepts, mixed within a wave), single-spike
    subvectors (terms far below l1), whole zero tiles, draws on and next to 1e-5 (threshold 0 and its neighbours),
    and K = 250 (padded codewords: always the term-by-term form)."""
    d, M = 16, 64 * 700 + 5
    rng = np.random.RandomState(K)
    cdag = rng.standard_normal((K, d)).astype(np.float32) * 0.3
    x = rng.standard_normal((M, d)).astype(np.float32)
    scale = np.float32(10.0) ** rng.randint(-30, 9, size=M).astype(np.float32)
    scale[: 64 * 200] = np.float32(1e-3)                       # plain waves: all lanes fast
    x *= scale[:, None]
    x[64 * 300: 64 * 302] = 0.0                                # two all-zero tiles
    spikes = rng.randint(0, M, size=2000)
    x[spikes, 1:] *= np.float32(1e-35)                         # one large element, the rest (nearly) underflown
    r = rng.random_sample(M).astype(np.float32)
    edge = np.float32(1e-5)
    r[:8] = [0.0, edge, np.nextafter(edge, np.float32(1)), np.nextafter(edge, np.float32(0)), np.float32(2.0 ** -24),
             np.float32(1.0) - np.float32(2.0 ** -24), np.float32(0.5), np.float32(1e-5) + np.float32(2.0 ** -40)]
    x = x.reshape(-1)
    dev = torch.device("cuda:0")
    g, c, rt = torch.from_numpy(x).to(dev), torch.from_numpy(cdag).to(dev), torch.from_numpy(r).to(dev)
    codes = torch.empty(M, dtype=torch.uint8, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = nat.new_workspace(dev, M)
    nat.pvq_encode(g, c, codes, u, ws, nat.RANDOM_GIVEN, rt, 0)
    torch.cuda.synchronize()
    oc, ou = oracle.pvq_encode(x, cdag, r)
    assert np.array_equal(codes.cpu().numpy().astype(np.int64), oc.astype(np.int64))
    assert np.array_equal(_bits(u.cpu().numpy()), _bits(ou))


def test_full_size_compress_and_decode_mean_equal_the_oracle(nat, oracle):
    """BASELINE size, everything against the oracle bit for bit: three ranks' 25 M-element gradients compressed
    (codes, u, lb, ub, levels) and their decode-mean (R = 1 and R = 3) -- rare-event errors (one subvector in
    1e5, as the packed-FMA trap produced) cannot hide at this size."""
    dev = torch.device("cuda:0")
    cb_np = _cb(16, 256)
    cb = torch.from_numpy(cb_np).to(dev)
    M = 25_000_000 // 16
    codes = torch.empty((3, M), dtype=torch.uint8, device=dev)
    levels = torch.empty((3, M), dtype=torch.uint8, device=dev)
    lb_ub = torch.empty((3, 2), dtype=torch.float32, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = nat.new_workspace(dev, M)
    dec = []
    for r in range(3):
        torch.manual_seed(100 + r)
        g = torch.randn(25_000_000, device=dev) * (1e-3 if r == 1 else 1.0)
        nat.hsq_encode(g, cb, codes[r], u, ws)
        nat.hsq_levels(u, 6, 0, None, 0, ws, lb_ub[r], levels[r])
        torch.cuda.synchronize()
        oc, ou = oracle.hsq_encode(g.cpu().numpy(), cb_np)
        lb, ub, lv = oracle.scalar_levels(ou, 6, 0, None)
        assert np.array_equal(codes[r].cpu().numpy(), oc.astype(np.uint8)), r
        assert np.array_equal(_bits(u.cpu().numpy()), _bits(ou)), r
        assert np.array_equal(_bits(lb_ub[r].cpu().numpy()), _bits(np.array([lb, ub], dtype=np.float32).reshape(-1)))
        assert np.array_equal(levels[r].cpu().numpy().astype(np.int64), lv.astype(np.int64)), r
        dec.append(oracle.hsq_decompress(oc, lv, lb, ub, cb_np, 6).reshape(-1))
    out = torch.empty(25_000_000, dtype=torch.float32, device=dev)
    nat.hsq_decode_sum(codes[0], levels[0], lb_ub[0], cb, 6, out, R=1)
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(dec[0]))
    nat.hsq_decode_sum(codes.reshape(-1), levels.reshape(-1), lb_ub.reshape(-1), cb, 6, out, R=3)
    want = ((dec[0] + dec[1]) + dec[2]) / np.float32(3.0)        # stack().mean(0): ascending sum, then / R
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(want))


@pytest.mark.parametrize("R", [8, 16])
def test_full_size_decode_mean_of_eight_and_sixteen_payloads_equals_the_oracle(nat, oracle, R):
    """The kernel ps_quantizer.py:48 becomes on 8 (16) GPUs -- decode-mean over R different payloads -- at BASELINE size,
    bit for bit against the oracle: R gradients of 25 M elements at different scales compressed on the device (the
    compress itself is pinned by the test above), every payload decoded by the ORACLE, summed in payload order from +0
    and divided by R (torch.stack(...).mean(0))."""
    dev = torch.device("cuda:0")
    cb_np = _cb(16, 256)
    cb = torch.from_numpy(cb_np).to(dev)
    M = 25_000_000 // 16
    codes = torch.empty((R, M), dtype=torch.uint8, device=dev)
    levels = torch.empty((R, M), dtype=torch.uint8, device=dev)
    lb_ub = torch.empty((R, 2), dtype=torch.float32, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = nat.new_workspace(dev, M)
    want = np.zeros(25_000_000, dtype=np.float32)
    for r in range(R):
        torch.manual_seed(500 + r)
        g = torch.randn(25_000_000, device=dev) * float(10.0 ** ((r % 5) - 3))
        nat.hsq_encode(g, cb, codes[r], u, ws)
        nat.hsq_levels(u, 6, 0, None, 0, ws, lb_ub[r], levels[r])
        torch.cuda.synchronize()
        lu = lb_ub[r].cpu().numpy()
        want = want + oracle.hsq_decompress(codes[r].cpu().numpy(), levels[r].cpu().numpy().astype(np.int32), lu[0], lu[1], cb_np, 6).reshape(-1)
        del g
    want = want / np.float32(R)
    out = torch.empty(25_000_000, dtype=torch.float32, device=dev)
    nat.hsq_decode_sum(codes.reshape(-1), levels.reshape(-1), lb_ub.reshape(-1), cb, 6, out, R=R)
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(want))


@pytest.mark.parametrize("d", [16, 8, 32, 12, 24])
def test_prefilter_equals_exact_mfma_at_full_size_and_fixup_rate(nat, d):
    """The f16 prefilter path (d = 16, 8 and 32; round 6: 12 and 24, the reference's repaired dimensions, as rows of 12 / 24
    floats through the d = 16 / 32 kernels) must reproduce the exact f32 MFMA kernel bit for bit on 25M
    elements (randn and randn*1e-3), and only a small fraction may need the second pass / exact scan."""
    dev = torch.device("cuda:0")
    cb = torch.from_numpy(_cb(d, 256)).to(dev)
    exact = 1 if d == 16 else 5
    for seed, scale in [(1234, 1.0), (77, 1e-3)]:
        torch.manual_seed(seed)
        g = torch.randn(25_000_000 // d * d, device=dev) * scale
        M = g.numel() // d
        res = {}
        for impl in (exact, 4):
            codes = torch.empty(M, dtype=torch.uint8, device=dev)
            u = torch.empty(M, dtype=torch.float32, device=dev)
            ws = nat.new_workspace(dev, M)
            nat.mark_worklist(ws, M)
            lb_ub = torch.empty(2, dtype=torch.float32, device=dev)
            nat.hsq_encode(g, cb, codes, u, ws, impl=impl)
            levels = torch.empty(M, dtype=torch.uint8, device=dev)
            nat.hsq_levels(u, 6, 0, None, 0, ws, lb_ub, levels)
            torch.cuda.synchronize()
            res[impl] = (codes, u, lb_ub, levels, ws)
        assert torch.equal(res[exact][0], res[4][0]), "codes differ between exact and prefilter kernels"
        assert torch.equal(res[exact][1].view(torch.int32), res[4][1].view(torch.int32)), "u differs bitwise"
        assert torch.equal(res[exact][2].view(torch.int32), res[4][2].view(torch.int32))
        assert torch.equal(res[exact][3], res[4][3])
        n_fix = nat.fixup_count(res[4][4], M)
        # one f16 MFMA per chain and k-step settles ~98.5 % of N(0,1) subvectors (d = 8: 98.4 %), the rest take the second pass
        # (three MFMAs) or an exact scan
        assert 0 < n_fix < M * 0.03, n_fix
        print("scale %g: fix-up worklist %d of %d subvectors (%.4f%%)" % (scale, n_fix, M, 100.0 * n_fix / M))


@pytest.mark.parametrize("d", [16, 8, 32, 12, 24])
@pytest.mark.parametrize("case", ["zeros", "constant", "ties", "huge", "tiny", "mixed_scales"])
def test_prefilter_degenerate_inputs_match_exact(nat, oracle, case, d):
    rng = np.random.RandomState(11)
    cbn = _cb(d, 256)
    M = 5000
    if case == "zeros":
        x = np.zeros(M * d, np.float32)
        x[d * 7:d * 8] = -0.0
    elif case == "constant":
        x = np.tile(rng.standard_normal(d).astype(np.float32), M)
    elif case == "ties":
        a = rng.randint(0, 256, M)
        b = (a + 1 + rng.randint(0, 255, M)) % 256
        x = (cbn[a] + rng.choice([-1.0, 1.0], M)[:, None].astype(np.float32) * cbn[b]).reshape(-1)
    elif case == "huge":
        x = (rng.standard_normal(M * d) * 1e32).astype(np.float32)
    elif case == "tiny":
        x = (rng.standard_normal(M * d) * 1e-30).astype(np.float32)
        x[:d * 100] *= 1e-12   # subnormal products
    else:
        x = (rng.standard_normal(M * d) * np.exp(rng.standard_normal(M * d) * 8)).astype(np.float32)
    x = np.ascontiguousarray(x, np.float32)
    ref_codes, ref_u = oracle.hsq_encode(x, cbn)
    res = gpu_compress(nat, x, cbn, 32, 0, impl=4)
    assert np.array_equal(res["codes"].cpu().numpy().astype(np.int32), ref_codes)
    assert np.array_equal(_bits(res["u"].cpu().numpy()), _bits(ref_u))
    n_fix = nat.fixup_count(res["ws"], M)
    if case == "zeros":
        assert n_fix == 0            # handled inline, never sent to the fix-up kernel
    if case in ("huge", "tiny"):
        assert n_fix > 0             # outside the proven range of the error bound -> exact path


@pytest.mark.parametrize("d", [16, 8, 32])
def test_prefilter_respects_unnormalised_codebooks(nat, oracle, d):
    """The error bound scales with the measured max ||c_k||_1, not an assumed unit norm."""
    rng = np.random.RandomState(12)
    cbn = (_cb(d, 256) * rng.uniform(0.2, 30.0, (256, 1))).astype(np.float32)
    x = rng.standard_normal(d * 20000).astype(np.float32)
    ref_codes, ref_u = oracle.hsq_encode(x, cbn)
    res = gpu_compress(nat, x, cbn, 32, 0, impl=4)
    assert np.array_equal(res["codes"].cpu().numpy().astype(np.int32), ref_codes)
    assert np.array_equal(_bits(res["u"].cpu().numpy()), _bits(ref_u))


def test_prefilter_fuzz_against_exact_kernels():
    """tools/fuzz_prefilter.py: random sizes / scales / sub-dimensions and adversarial subvectors (sparse, one-hot,
    integer-valued, duplicated, exact codeword multiples and sums, subnormal and huge magnitudes, unnormalised
    codebooks): the prefilter encode equals the exact f32 MFMA kernels bit for bit."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_prefilter.py")
    r = subprocess.run([sys.executable, tool, "120", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "120 rounds, 0 mismatching" in r.stdout




def test_prefilter_fuzz_smaller_codebooks_against_exact_kernel():
    """tools/fuzz_prefilter.py ... smallk: K = 4 ... 256 in multiples of 4 (round 6: --k-bit 5 / 6 and K == dim on the kernels
    that score one / two row blocks, any other K on the eight-block kernel over zero rows), the same adversarial inputs plus
    infinities and NaN: codes on every subvector and projections equal the exact f32 MFMA kernel's bit for bit; no code >= K."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_prefilter.py")
    r = subprocess.run([sys.executable, tool, "200", "11", "smallk"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "200 rounds, 0 mismatching" in r.stdout


@pytest.mark.parametrize("d,K", [(16, 64), (16, 32), (16, 16), (8, 8), (32, 32), (32, 64), (24, 64), (12, 32), (8, 64), (16, 4), (16, 132)])
def test_prefilter_smaller_codebooks_match_oracle_at_full_size(nat, oracle, d, K):
    """25 M elements through the prefilter kernels with fewer than 256 codewords (one / two / eight row blocks) against the CPU
    oracle on a sample of tiles and against the exact kernel everywhere."""
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(100 + d + K)
    cbn = rng.standard_normal((K, d)).astype(np.float32)
    cbn /= np.linalg.norm(cbn, axis=1, keepdims=True)
    M = 25_000_000 // d
    torch.manual_seed(77 + K)
    g = torch.randn(M * d, device=dev) * 1e-3
    g[5 * d:6 * d] = 0.0
    cb = torch.from_numpy(cbn).to(dev)
    outs = {}
    for impl in (4, 5):
        codes = torch.empty(M, dtype=torch.uint8, device=dev)
        u = torch.empty(M, dtype=torch.float32, device=dev)
        ws = nat.new_workspace(dev, M)
        nat.hsq_encode(g, cb, codes, u, ws, impl=impl)
        torch.cuda.synchronize()
        outs[impl] = (codes, u)
    assert torch.equal(outs[4][0], outs[5][0]) and torch.equal(outs[4][1].view(torch.int32), outs[5][1].view(torch.int32))
    assert int(outs[4][0].max()) < K
    n = 4096 * d
    for start in (0, (M // 2) * d, M * d - n):
        ref_codes, ref_u = oracle.hsq_encode(g[start:start + n].cpu().numpy(), cbn)
        m0 = start // d
        assert np.array_equal(outs[4][0][m0:m0 + 4096].cpu().numpy().astype(np.int32), ref_codes)
        assert np.array_equal(_bits(outs[4][1][m0:m0 + 4096].cpu().numpy()), _bits(ref_u))


def test_compress_and_decode_replay_from_a_hip_graph(nat):
    """The C ABI keeps no per-thread call state and its kernels reset their own counters: encode + levels + decode-mean
    captured ONCE into a HIP graph (torch.cuda.graph: stream capture of the launches) give, on every replay with new
    contents in the same input buffer, the bits of the eager calls."""
    dev = torch.device("cuda:0")
    cb = torch.from_numpy(_cb(16, 256)).to(dev)
    M = 70_001
    x = torch.empty(M * 16, dtype=torch.float32, device=dev)

    def buffers():
        return dict(codes=torch.empty(M, dtype=torch.uint8, device=dev), u=torch.empty(M, dtype=torch.float32, device=dev),
                    ws=nat.new_workspace(dev, M), lb_ub=torch.empty(2, dtype=torch.float32, device=dev),
                    levels=torch.empty(M, dtype=torch.uint8, device=dev), out=torch.empty(M * 16, dtype=torch.float32, device=dev))

    def run(b):
        nat.hsq_encode(x, cb, b["codes"], b["u"], b["ws"])
        nat.hsq_levels(b["u"], 6, 0, None, 0, b["ws"], b["lb_ub"], b["levels"])
        nat.hsq_decode_sum(b["codes"], b["levels"], b["lb_ub"], cb, 6, b["out"], R=1)

    captured, eager = buffers(), buffers()
    torch.manual_seed(5)
    x.copy_(torch.randn(M * 16, device=dev))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):      # the first calls set kernel attributes: not inside a capture
        run(captured)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with _capture(graph):
        run(captured)
    for trial in range(3):
        x.copy_(torch.randn(M * 16, device=dev) * (10.0 ** -trial))
        graph.replay()
        run(eager)
        torch.cuda.synchronize()
        for k in ("codes", "levels", "lb_ub", "out"):
            assert torch.equal(captured[k].view(torch.uint8), eager[k].view(torch.uint8)), (trial, k)


@pytest.mark.parametrize("R,M,K", [(1, 5003, 64), (6, 5002, 64), (13, 5001, 32)])
def test_hsq_decode_sum_smaller_codebooks(nat, oracle, R, M, K):
    """d = 16 with fewer than 256 codewords goes through the same kernels (the LDS image holds K rows)."""
    test_hsq_decode_sum_matches_oracle_mean(nat, oracle, R, M, K)


@pytest.mark.parametrize("n_bit,random,packed6,M", [(6, 0, False, 70_003), (6, 2, False, 4098), (5, 1, True, 100_001), (6, 0, True, 1_300_002),
                                                    (2, 1, False, 5), (8, 0, False, 64 * 1000), (7, 1, False, 9_999), (6, 0, False, 3)])
def test_fused_levels_decode_equals_the_two_calls(nat, n_bit, random, packed6, M, special=None):
    """gq_hsq_levels_decode == gq_hsq_levels followed by gq_hsq_decode_sum (R = 1): lb_ub, the level section and the decoded
    tensor bit for bit -- deterministic, given draws, device draws; byte and packed levels; ragged and multi-item M."""
    dev = torch.device("cuda:0")
    cb = torch.from_numpy(_cb(16, 256)).to(dev)
    torch.manual_seed(n_bit * 100 + random)
    x = torch.randn(M * 16, device=dev) * 0.03
    if special == "zeros":          # lb == ub: probabilistic_scalar_compressor.py:15-16, all levels 0
        x.zero_()
    elif special == "nan":          # a NaN projection: lb = ub = NaN, every level the byte of INT_MIN, the decode NaN
        x[16 * (M // 2) + 3] = float("nan")
    elif special == "inf":
        x[5] = float("inf")
    codes = torch.empty(M, dtype=torch.uint8, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = nat.new_workspace(dev, M)
    nat.hsq_encode(x, cb, codes, u, ws)
    r = torch.rand(M, device=dev) if random == 1 else None
    nlev = nat.packed6_bytes(M) + 4 if packed6 else M
    res = []
    for fused in (False, True):
        lb_ub = torch.zeros(2, dtype=torch.float32, device=dev)
        levels = torch.zeros(nlev, dtype=torch.uint8, device=dev)
        out = torch.full((M * 16,), 7.0, dtype=torch.float32, device=dev)
        if fused:
            assert nat.hsq_levels_decode(u, n_bit, random, r, 99, ws, lb_ub, levels, codes, cb, out, packed6)
        else:
            nat.hsq_levels(u, n_bit, random, r, 99, ws, lb_ub, levels, packed6)
            if packed6:
                _decode_packed_unaligned(nat, codes, levels, lb_ub, cb, n_bit, out, M)
            else:
                nat.hsq_decode_sum(codes, levels, lb_ub, cb, n_bit, out, R=1)
        torch.cuda.synchronize()
        res.append((lb_ub.clone(), levels.clone(), out.clone()))
    for a, b, name in zip(res[0], res[1], ("lb_ub", "levels", "out")):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), name


def _decode_packed_unaligned(nat, codes, levels, lb_ub, cb, n_bit, out, M):
    """gq_hsq_decode_sum_strided with level_bytes = GQ_LEVELS_PACKED6 on separate buffers."""
    import ctypes
    L = nat.lib()
    rc = L.gq_hsq_decode_sum_strided(ctypes.c_void_p(codes.data_ptr()), 1, ctypes.c_int64(M), ctypes.c_void_p(levels.data_ptr()),
                                     nat.LEVELS_PACKED6, ctypes.c_int64(levels.numel()), ctypes.c_void_p(lb_ub.data_ptr()),
                                     ctypes.c_int64(8), ctypes.c_void_p(cb.data_ptr()), 1, ctypes.c_int64(M), 16, 256, n_bit,
                                     ctypes.c_void_p(out.data_ptr()), None)
    assert rc == 0, L.gq_last_error()


def test_hsq_decode_sum_more_than_1024_payloads(nat, oracle):
    """R above what the pipelined kernels hold (lb, ub) for in LDS goes through the generic d = 16 kernel."""
    test_hsq_decode_sum_matches_oracle_mean(nat, oracle, 1025, 260)


@pytest.mark.parametrize("special", ["zeros", "nan", "inf"])
@pytest.mark.parametrize("packed6", [False, True])
def test_fused_levels_decode_degenerate_inputs(nat, special, packed6):
    """All-zero gradient (lb == ub), a NaN and an infinite element: the fused launch and the two calls still agree on every bit."""
    test_fused_levels_decode_equals_the_two_calls(nat, 6, 0, packed6, 4099, special)


@pytest.mark.parametrize("M", [1, 2, 3, 5, 63, 4097, 70_002])
@pytest.mark.parametrize("R", [1, 3, 8, 13])
def test_decode_and_fused_kernels_write_nothing_past_their_buffers(nat, M, R):
    """Canaries right behind `out` (and behind the level section of the fused launch): the partial last group and the
    re-requests of the last item must not reach past M subvectors."""
    dev = torch.device("cuda:0")
    cb = torch.from_numpy(_cb(16, 256)).to(dev)
    torch.manual_seed(M + R)
    x = torch.randn(M * 16, device=dev) * 0.1
    codes = torch.empty(M, dtype=torch.uint8, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = nat.new_workspace(dev, M)
    nat.hsq_encode(x, cb, codes, u, ws)
    CAN = 256
    big_out = torch.full((M * 16 + CAN,), 12345.0, dtype=torch.float32, device=dev)
    out = big_out[:M * 16]
    for packed6 in (False, True):
        nlev = nat.packed6_bytes(M) if packed6 else M
        big_lv = torch.full((nlev + CAN,), 0xA5, dtype=torch.uint8, device=dev)
        lb_ub = torch.zeros(2, dtype=torch.float32, device=dev)
        if R == 1:
            assert nat.hsq_levels_decode(u, 6, 0, None, 0, ws, lb_ub, big_lv[:nlev + (4 if packed6 else 0)] if packed6 else big_lv[:nlev],
                                         codes, cb, out, packed6)
            torch.cuda.synchronize()
            assert bool((big_lv[nlev:] == 0xA5).all()), "fused launch wrote past the level section"
            assert bool((big_out[M * 16:] == 12345.0).all()), "fused launch wrote past out"
        if not packed6:
            levels = torch.empty(M, dtype=torch.uint8, device=dev)
            nat.hsq_levels(u, 6, 0, None, 0, ws, lb_ub, levels)
            cR, lR, bR = codes.repeat(R), levels.repeat(R), lb_ub.repeat(R)
            big_out.fill_(12345.0)
            nat.hsq_decode_sum(cR, lR, bR, cb, 6, out, R=R)
            torch.cuda.synchronize()
            assert bool((big_out[M * 16:] == 12345.0).all()), "decode-mean wrote past out"
