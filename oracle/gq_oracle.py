"""ctypes bindings + numpy front-end for the CPU parity oracle (oracle/gq_oracle.c).

TEST INFRASTRUCTURE ONLY: the product package never imports this module.
Each function mirrors one step of the reference's hot path; the reference
file:line each follows is cited in gq_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgq_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile oracle/gq_oracle.c with gcc (seconds)."""
    srcs = [os.path.join(_HERE, "gq_oracle.c"), os.path.join(_HERE, "gq_cpu.c")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libgq_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _require_avx2_fma():
    """The oracle is built with -mavx2 -mfma (hardware fmaf, eight codeword chains per 256-bit register): on a host
    without them the first call would die with SIGILL.  Say so instead."""
    try:
        flags = set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = set(line.split(":", 1)[1].split())
                break
    except OSError:
        return      # not Linux: nothing to check against
    missing = [f for f in ("avx2", "fma") if f not in flags]
    if flags and missing:
        raise RuntimeError("oracle/libgq_oracle.so is built with -mavx2 -mfma, but this host's CPU lacks %s (/proc/cpuinfo); "
                           "the CPU checker cannot run here" % " and ".join(missing))


def lib():
    global _lib
    if _lib is None:
        _require_avx2_fma()
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.gq_oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().gq_oracle_num_threads())


def set_num_threads(n):
    lib().gq_oracle_set_num_threads(ctypes.c_int(int(n)))


def hsq_encode(grad, codebook):
    """-> (codes int32[M], u f32[M])."""
    codebook = _f32(codebook)
    K, d = codebook.shape
    g = _f32(grad).reshape(-1)
    assert g.size % d == 0
    M = g.size // d
    codes = np.empty(M, np.int32)
    u = np.empty(M, np.float32)
    lib().gq_oracle_hsq_encode(_p(g, _f32p), _p(codebook, _f32p), ctypes.c_int64(M), ctypes.c_int(d),
                               ctypes.c_int(K), _p(codes, _i32p), _p(u, _f32p))
    return codes, u


def hsq_encode_scalar(grad, codebook):
    """The literal one-codeword-at-a-time form of hsq_encode (what the blocked form must equal bit for bit)."""
    codebook = _f32(codebook)
    K, d = codebook.shape
    g = _f32(grad).reshape(-1)
    assert g.size % d == 0
    M = g.size // d
    codes = np.empty(M, np.int32)
    u = np.empty(M, np.float32)
    lib().gq_oracle_hsq_encode_scalar(_p(g, _f32p), _p(codebook, _f32p), ctypes.c_int64(M), ctypes.c_int(d),
                                      ctypes.c_int(K), _p(codes, _i32p), _p(u, _f32p))
    return codes, u


def minmax(u):
    u = _f32(u).reshape(-1)
    out = np.empty(2, np.float32)
    lib().gq_oracle_minmax(_p(u, _f32p), ctypes.c_int64(u.size), _p(out, _f32p))
    return np.float32(out[0]), np.float32(out[1])


def scalar_levels(u, n_bit, random=0, r=None, lb=None, ub=None):
    """-> (lb, ub, levels int32[M])."""
    u = _f32(u).reshape(-1)
    if lb is None:
        lb, ub = minmax(u)
    if random:
        r = _f32(r).reshape(-1)
        assert r.size == u.size
    levels = np.empty(u.size, np.int32)
    lib().gq_oracle_scalar_levels(_p(u, _f32p), ctypes.c_int64(u.size), ctypes.c_int(n_bit),
                                  ctypes.c_int(1 if random else 0), _p(r, _f32p) if random else None,
                                  ctypes.c_float(lb), ctypes.c_float(ub), _p(levels, _i32p))
    return np.float32(lb), np.float32(ub), levels


def scalar_decode(levels, n_bit, lb, ub):
    levels = np.ascontiguousarray(levels, np.int32).reshape(-1)
    out = np.empty(levels.size, np.float32)
    lib().gq_oracle_scalar_decode(_p(levels, _i32p), ctypes.c_int64(levels.size), ctypes.c_int(n_bit),
                                  ctypes.c_float(lb), ctypes.c_float(ub), _p(out, _f32p))
    return out


def hsq_decode(codes, norms, codebook):
    codebook = _f32(codebook)
    K, d = codebook.shape
    codes = np.ascontiguousarray(codes, np.int32).reshape(-1)
    norms = _f32(norms).reshape(-1)
    out = np.empty(codes.size * d, np.float32)
    lib().gq_oracle_hsq_decode(_p(codes, _i32p), _p(norms, _f32p), _p(codebook, _f32p),
                               ctypes.c_int64(codes.size), ctypes.c_int(d), _p(out, _f32p))
    return out


def hsq_compress(grad, codebook, n_bit, random=0, r=None):
    """Whole compress (encode + min/max + levels) -> dict(codes,u,lb,ub,levels)."""
    codebook = _f32(codebook)
    K, d = codebook.shape
    g = _f32(grad).reshape(-1)
    M = g.size // d
    codes = np.empty(M, np.int32)
    u = np.empty(M, np.float32)
    levels = np.empty(M, np.int32)
    lbub = np.empty(2, np.float32)
    if random:
        r = _f32(r).reshape(-1)
    lib().gq_oracle_hsq_compress(_p(g, _f32p), _p(codebook, _f32p), ctypes.c_int64(M), ctypes.c_int(d),
                                 ctypes.c_int(K), ctypes.c_int(n_bit), ctypes.c_int(1 if random else 0),
                                 _p(r, _f32p) if random else None, _p(codes, _i32p), _p(u, _f32p),
                                 _p(lbub, _f32p), _p(levels, _i32p))
    return dict(codes=codes, u=u, lb=np.float32(lbub[0]), ub=np.float32(lbub[1]), levels=levels)


def hsq_decompress(codes, levels, lb, ub, codebook, n_bit):
    """levels -> norms -> codebook gather * norm (flat f32)."""
    return hsq_decode(codes, scalar_decode(levels, n_bit, lb, ub), codebook)


def mean_users(decoded):
    """decoded: [U, n] -> mean over users (ps_quantizer.py:48)."""
    dec = _f32(decoded)
    U = dec.shape[0]
    n = dec[0].size
    dec = dec.reshape(U, n)
    out = np.empty(n, np.float32)
    lib().gq_oracle_mean_users(_p(dec, _f32p), ctypes.c_int(U), ctypes.c_int64(n), _p(out, _f32p))
    return out


def qsgd_compress(grad, d, n_bit, random=0, r=None):
    g = _f32(grad).reshape(-1)
    assert g.size % d == 0
    Mb = g.size // d
    norm = np.empty(Mb, np.float32)
    signs = np.empty(g.size, np.uint8)
    levels = np.empty(g.size, np.int32)
    if random:
        r = _f32(r).reshape(-1)
    lib().gq_oracle_qsgd_compress(_p(g, _f32p), ctypes.c_int64(Mb), ctypes.c_int(d), ctypes.c_int(n_bit),
                                  ctypes.c_int(1 if random else 0), _p(r, _f32p) if random else None,
                                  _p(norm, _f32p), _p(signs, _u8p), _p(levels, _i32p))
    return norm, signs, levels


def qsgd_decompress(norm, signs, levels, d, n_bit):
    norm = _f32(norm).reshape(-1)
    signs = np.ascontiguousarray(signs, np.uint8).reshape(-1)
    levels = np.ascontiguousarray(levels, np.int32).reshape(-1)
    out = np.empty(levels.size, np.float32)
    lib().gq_oracle_qsgd_decompress(_p(norm, _f32p), _p(signs, _u8p), _p(levels, _i32p),
                                    ctypes.c_int64(norm.size), ctypes.c_int(d), ctypes.c_int(n_bit),
                                    _p(out, _f32p))
    return out


def pvq_encode(grad, c_dagger, r, sub_rows=0):
    """ProbabilisticVectorCompressor encode (pinned by tests/golden/pvq_*.npz) -> (codes, u), or with
    sub_rows > 0 -> (codes, u, l1 f32[M], p f32[sub_rows, K], cumsum f32[sub_rows, K])."""
    cd = _f32(c_dagger)
    K, d = cd.shape
    g = _f32(grad).reshape(-1)
    M = g.size // d
    r = _f32(r).reshape(-1)
    assert r.size == M
    codes = np.empty(M, np.int32)
    u = np.empty(M, np.float32)
    sub_rows = min(int(sub_rows), M)
    l1 = np.empty(M, np.float32)
    p = np.empty((sub_rows, K), np.float32) if sub_rows else None
    cum = np.empty((sub_rows, K), np.float32) if sub_rows else None
    lib().gq_oracle_pvq_encode_ex(_p(g, _f32p), _p(cd, _f32p), ctypes.c_int64(M), ctypes.c_int(d), ctypes.c_int(K),
                                  _p(r, _f32p), _p(codes, _i32p), _p(u, _f32p), _p(l1, _f32p), _p(p, _f32p),
                                  _p(cum, _f32p), ctypes.c_int64(sub_rows))
    if sub_rows:
        return codes, u, l1, p, cum
    return codes, u


def pvq_compress(grad, c_dagger, r, n_bit):
    """ProbabilisticVectorCompressor.compress with deterministic norms (args.random = 0) -> dict."""
    codes, u = pvq_encode(grad, c_dagger, r)
    out = dict(codes=codes, u=u)
    if n_bit != 32:
        lb, ub, levels = scalar_levels(u, n_bit)
        out.update(lb=lb, ub=ub, levels=levels)
    return out


def pvq_decompress(sig, codewords, n_bit):
    norms = sig["u"] if n_bit == 32 else scalar_decode(sig["levels"], n_bit, sig["lb"], sig["ub"])
    return hsq_decode(sig["codes"], norms, codewords)


def residual_compress(grad, codewords1, codewords2, c_dagger, r, n_bit):
    """ResidualCompressor.compress (residual_compressor.py:15-24), deterministic norms:
    stage 1 nearest neighbour -> residual -= decoded (in place, f32) -> stage 2 probabilistic vector."""
    g = _f32(grad).reshape(-1).copy()
    s1 = hsq_compress(g, codewords1, n_bit) if n_bit != 32 else dict(zip(("codes", "u"), hsq_encode(g, codewords1)))
    dec1 = (hsq_decompress(s1["codes"], s1["levels"], s1["lb"], s1["ub"], codewords1, n_bit) if n_bit != 32
            else hsq_decode(s1["codes"], s1["u"], codewords1))
    g = (g - dec1).astype(np.float32)
    s2 = pvq_compress(g, c_dagger, r, n_bit)
    dec2 = pvq_decompress(s2, codewords2, n_bit)
    # residual_compressor.py:26-32: torch.stack([d1, d2]).sum(0) == d1 + d2 in f32
    return s1, s2, dec1, dec2, (dec1 + dec2).astype(np.float32)
