#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference does not travel to the GPU box):

    cd /root/reference && python -B /root/repo/tests/golden/make_golden.py [name-prefix ...]

With name prefixes only the matching fixtures are (re)written and the other MANIFEST entries are
kept (npz archives carry timestamps, so rewriting everything would churn every file).

The reference's compressors read ``./codebooks/...`` relative to the cwd
(compressors/nearest_neighbor_compressor.py:50), hence the cwd requirement.
``-B`` keeps ``__pycache__`` out of the read-only reference tree.

What is written (all data -- inputs and the reference's outputs; no reference
source text):

* ``codebooks/*.fvecs``               raw codebook data files the tests need (copied bytes)
* ``codebook_d{d}_k{K}_normalized.npy`` the reference's post-``normalize`` codebook
* ``hsq_*.npz``                       NearestNeighborCompressor.compress/decompress vectors
* ``psq_*.npz``                       PSQuantizer.record/apply vectors (FCN-like grads)
* ``ring_*.npz``                      RingQuantizer.record/apply vectors (same layout as psq_*)
* ``qsgd_*.npz``                      QSGDCompressor vectors
* ``MANIFEST.json``                   sha256 of every fixture + versions
"""
import hashlib
import json
import os
import shutil
import sys
from argparse import Namespace

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

if os.path.realpath(os.getcwd()) != os.path.realpath(REF):
    sys.exit("run with cwd=%s (the reference opens ./codebooks/...)" % REF)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from compressors import (NearestNeighborCompressor, QSGDCompressor,  # noqa: E402
                         IdenticalCompressor)
from compressors.probabilistic_scalar_compressor import ProbabilisticScalarCompressor  # noqa: E402
from quantizers import Quantizer  # noqa: E402

torch.set_num_threads(8)
ONLY = sys.argv[1:]


def want(name):
    return not ONLY or any(name.startswith(o) for o in ONLY)


def make_args(**kw):
    base = dict(c_dim=16, k_bit=8, n_bit=6, no_cuda=True, random=0, ef=False,
                two_phase=False, scale="exp", num_users=4, mode="ps", cr=256)
    base.update(kw)
    return Namespace(**base)


def f64_top2_gap(cb, x, d):
    """Relative gap between the best and second-best |<c,v>| in float64
    (tests use it to know which subvectors are safely away from a tie)."""
    v = x.reshape(-1, d).astype(np.float64)
    p = np.abs(v @ cb.astype(np.float64).T)
    part = np.partition(p, -2, axis=1)
    top, sec = part[:, -1], part[:, -2]
    with np.errstate(divide="ignore", invalid="ignore"):
        gap = np.where(top > 0, (top - sec) / top, 0.0)
    return gap.astype(np.float64)


def hsq_case(name, x, shape=None, seed_r=None, **argkw):
    """One NearestNeighborCompressor fixture: input, signature, decoded."""
    if not want(name):
        return
    args = make_args(**argkw)
    x = np.ascontiguousarray(x, dtype=np.float32)
    shape = tuple(shape) if shape is not None else x.shape
    t = torch.from_numpy(x.copy()).view(*shape)
    comp = NearestNeighborCompressor(t.numel(), t.shape, args)
    out = dict(x=x.reshape(shape), dim=np.int32(comp.dim), K=np.int32(comp.K),
               n_bit=np.int32(args.n_bit), random=np.int32(args.random))
    M = t.numel() // comp.dim
    if args.random and args.n_bit != 32:
        # The reference draws r = torch.rand(M) from the global CPU generator
        # inside compress (probabilistic_scalar_compressor.py:23); reproduce the
        # same draw by reseeding.
        torch.manual_seed(seed_r)
        r = torch.rand(M).numpy().copy()
        out["r"] = r
        torch.manual_seed(seed_r)
    sig = comp.compress(t)
    norms, codes = sig
    out["codes"] = codes.numpy().copy()
    if args.n_bit != 32:
        lb, ub, l = norms
        out["lb"] = np.float32(lb.item())
        out["ub"] = np.float32(ub.item())
        out["levels"] = l.numpy().copy()
        # the un-quantised projections, from an n_bit=32 twin
        comp32 = NearestNeighborCompressor(t.numel(), t.shape, make_args(**dict(argkw, n_bit=32)))
        u32, codes32 = comp32.compress(t)
        assert torch.equal(codes32, codes)
        out["u"] = u32.numpy().copy()
    else:
        out["u"] = norms.numpy().copy()
    dec = comp.decompress(sig)
    out["decoded"] = dec.numpy().copy()
    cb = comp.codewords.numpy()
    out["gap64"] = f64_top2_gap(cb, x, comp.dim)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    return comp


def qsgd_case(name, x, shape=None, seed_r=None, **argkw):
    if not want(name):
        return
    args = make_args(**argkw)
    x = np.ascontiguousarray(x, dtype=np.float32)
    shape = tuple(shape) if shape is not None else x.shape
    t = torch.from_numpy(x.copy()).view(*shape)
    comp = QSGDCompressor(t.numel(), t.shape, args)
    out = dict(x=x.reshape(shape), dim=np.int32(comp.dim), n_bit=np.int32(args.n_bit),
               random=np.int32(args.random))
    if args.random:
        torch.manual_seed(seed_r)
        out["r"] = torch.rand(t.numel() // comp.dim, comp.dim).numpy().copy()
        torch.manual_seed(seed_r)
    norm, signs, l = comp.compress(t)
    out["norm"] = norm.numpy().copy()
    out["signs"] = signs.numpy().copy()
    out["levels"] = l.numpy().copy()
    out["decoded"] = comp.decompress([norm, signs, l]).numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def psq_case(name, shapes, users, epoch, seed, scale_in=1e-2, steps=1, **argkw):
    """PSQuantizer.record x users + apply, `steps` times, on given parameter shapes."""
    if not want(name):
        return
    args = make_args(num_users=users, **argkw)
    g = torch.Generator().manual_seed(seed)
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
    COMP = {"hsq": NearestNeighborCompressor, "qsgd": QSGDCompressor,
            "sgd": IdenticalCompressor}[argkw.get("quantizer", "hsq")]
    if "quantizer" in argkw:
        delattr(args, "quantizer")
    q = Quantizer(COMP, params, args)
    out = dict(users=np.int32(users), epoch=np.int32(epoch), steps=np.int32(steps),
               n_params=np.int32(len(shapes)))
    for st in range(steps):
        for u in range(users):
            for i, p in enumerate(params):
                grad = torch.randn(p.shape, generator=g) * scale_in
                out["grad_s%d_u%d_p%d" % (st, u, i)] = grad.numpy().copy()
                p.grad = grad.clone()
            q.record(u, epoch=epoch)
        q.apply()
        for i, p in enumerate(params):
            out["agg_s%d_p%d" % (st, i)] = p.grad.data.numpy().copy()
    if args.ef:
        for i, p in enumerate(params):
            for u in range(users):
                out["err_p%d_u%d" % (i, u)] = p.error[u].numpy().copy()
            if args.two_phase:
                out["serr_p%d" % i] = p.server_error.numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def main():
    # ---- codebooks (data) ---------------------------------------------------
    os.makedirs(os.path.join(OUT, "codebooks", "learned_codebook"), exist_ok=True)
    from utils.vecs_io import fvecs_read
    from utils.vec_np import normalize
    for d, K in ([] if ONLY else [(16, 256), (8, 32), (24, 64), (12, 512), (32, 256), (8, 256)]):
        fn = "angular_dim_%d_Ks_%d.fvecs" % (d, K)
        src = os.path.join(REF, "codebooks", "learned_codebook", fn)
        dst = os.path.join(OUT, "codebooks", "learned_codebook", fn)
        shutil.copyfile(src, dst)
        os.chmod(dst, 0o644)
        _, cb = normalize(fvecs_read(src))
        np.save(os.path.join(OUT, "codebook_d%d_k%d_normalized.npy" % (d, K)), cb)

    rng = np.random.RandomState(20240917)

    # ---- HSQ main vectors ---------------------------------------------------
    x1 = rng.standard_normal(65536).astype(np.float32)
    hsq_case("hsq_randn_s1_det", x1, random=0)
    hsq_case("hsq_randn_s1_rand", x1, random=1, seed_r=4321)
    hsq_case("hsq_randn_s1_n32", x1[:16384], n_bit=32)
    hsq_case("hsq_randn_s1_n8_det", x1[:16384], n_bit=8, random=0)
    hsq_case("hsq_randn_s1_n2_rand", x1[:16384], n_bit=2, random=1, seed_r=99)
    x2 = (rng.standard_normal(32768) * 1e-3).astype(np.float32)
    hsq_case("hsq_randn_s1e-3_det", x2, random=0)
    hsq_case("hsq_randn_s1e-3_rand", x2[:16384], random=1, seed_r=777)
    # a 2-D "conv-like" shape
    hsq_case("hsq_shape_64x3x3x3_det", (rng.standard_normal(1728) * 1e-2).astype(np.float32),
             shape=(64, 3, 3, 3), random=0)
    # tiny M (MKL small-shape paths): 1024 elems (64 subvectors), 16 elems (1), 48 (3)
    hsq_case("hsq_small_1024_det", rng.standard_normal(1024).astype(np.float32), random=0)
    hsq_case("hsq_small_16_det", rng.standard_normal(16).astype(np.float32), random=0)
    hsq_case("hsq_small_48_rand", rng.standard_normal(48).astype(np.float32), random=1, seed_r=5)
    # heavy-tailed / mixed scale input
    xh = (rng.standard_normal(32768) * np.exp(rng.standard_normal(32768) * 3)).astype(np.float32)
    hsq_case("hsq_heavytail_det", xh, random=0)
    # subnormal-scale input
    hsq_case("hsq_subnormal_det", (rng.standard_normal(4096) * 1e-41).astype(np.float32), random=0)

    # ---- HSQ edge vectors ---------------------------------------------------
    hsq_case("hsq_zeros_det", np.zeros(4096, np.float32), random=0)
    hsq_case("hsq_zeros_rand", np.zeros(4096, np.float32), random=1, seed_r=1)
    one = rng.standard_normal(16).astype(np.float32)
    hsq_case("hsq_constant_u_det", np.tile(one, 256), random=0)
    hsq_case("hsq_constant_u_rand", np.tile(one, 256), random=1, seed_r=2)
    # crafted ties / near-ties against the d16 K256 codebook
    cb = np.load(os.path.join(OUT, "codebook_d16_k256_normalized.npy"))
    ties = []
    pr = np.random.RandomState(7)
    for _ in range(512):
        a, b = pr.choice(256, 2, replace=False)
        sgn = pr.choice([-1.0, 1.0])
        ties.append(cb[a] + sgn * cb[b])           # |<c_a,v>| == |<c_b,v>| in exact arithmetic
    for _ in range(256):
        a = pr.randint(256)
        ties.append(cb[a] * np.float32(pr.uniform(0.1, 10)))   # on a codeword direction
    for _ in range(128):
        v = np.zeros(16, np.float32)
        v[pr.randint(16)] = pr.choice([-1.0, 1.0]) * 2.0 ** pr.randint(-20, 20)
        ties.append(v)                                  # axis-aligned, power of two
    for _ in range(128):
        a, b, c = pr.choice(256, 3, replace=False)
        ties.append(cb[a] - cb[b] + cb[c])
    ties = np.asarray(ties, np.float32).reshape(-1)
    hsq_case("hsq_ties_det", ties, random=0)

    # ---- other (d, K) --------------------------------------------------------
    hsq_case("hsq_d8_k32_det", rng.standard_normal(8192).astype(np.float32), c_dim=8, k_bit=5, random=0)
    hsq_case("hsq_d8_k256_rand", rng.standard_normal(8192).astype(np.float32), c_dim=8, k_bit=8,
             random=1, seed_r=11)
    # dim repair 16 -> 24 (size 1032 = 24*43, not divisible by 16), K=64
    hsq_case("hsq_d24_k64_repair_det", rng.standard_normal(1032).astype(np.float32), c_dim=16, k_bit=6,
             random=0)
    # k_bit > 8 -> int32 codes
    hsq_case("hsq_d12_k512_det", rng.standard_normal(12 * 700).astype(np.float32), c_dim=12, k_bit=9,
             random=0)
    hsq_case("hsq_d32_k256_det", rng.standard_normal(32 * 512).astype(np.float32), c_dim=32, k_bit=8,
             random=0)

    # ---- QSGD -----------------------------------------------------------------
    xq = (rng.standard_normal(128 * 96) * 1e-2).astype(np.float32)
    xq[128 * 5:128 * 6] = 0.0          # a zero bucket (0/0 -> NaN -> INT_MIN level, decodes 0)
    qsgd_case("qsgd_d128_n2_det", xq, c_dim=128, n_bit=2, random=0)
    qsgd_case("qsgd_d128_n2_rand", xq, c_dim=128, n_bit=2, random=1, seed_r=31)
    qsgd_case("qsgd_d128_n4_rand", xq, c_dim=128, n_bit=4, random=1, seed_r=32)
    qsgd_case("qsgd_repair_1728_rand", (rng.standard_normal(1728)).astype(np.float32),
              shape=(64, 3, 3, 3), c_dim=128, n_bit=2, random=1, seed_r=33)
    # c_dim=0 -> one bucket spanning the tensor (TernGrad-like), qsgd_compressor.py:15-16
    qsgd_case("qsgd_cdim0_n1_rand", (rng.standard_normal(4096)).astype(np.float32), c_dim=0, n_bit=1,
              random=1, seed_r=34)

    # ---- PSQuantizer: a shrunken FCN-like parameter list (models/fcn.py has
    # [256,784],[256],[10,256],[10]); two weight matrices above the 1000-element
    # threshold (ps_quantizer.py:18) and two identity-compressed biases.
    fcn = [(96, 112), (96,), (12, 96), (12,)]
    psq_case("psq_fcn_u4_det", fcn, users=4, epoch=1, seed=101, random=0)
    psq_case("psq_fcn_u4_ef", fcn, users=4, epoch=1, seed=102, random=0, ef=True, steps=2)
    psq_case("psq_fcn_u4_twophase", fcn, users=4, epoch=1, seed=103, random=0, two_phase=True)
    psq_case("psq_fcn_u4_ef_twophase", fcn, users=4, epoch=2, seed=104, random=0, ef=True,
             two_phase=True, steps=2)
    psq_case("psq_fcn_u2_ef_scale0.5", fcn, users=2, epoch=3, seed=105, random=0, ef=True,
             scale="0.5", steps=2)
    psq_case("psq_fcn_u3_qsgd", fcn, users=3, epoch=1, seed=106, random=0, quantizer="qsgd",
             c_dim=128, n_bit=2)

    # ---- RingQuantizer (quantizers/ring_quantizer.py): same record/apply protocol, sum semantics
    psq_case("ring_fcn_u3_det", fcn, users=3, epoch=1, seed=107, random=0, mode="ring")
    psq_case("ring_fcn_u3_ef", fcn, users=3, epoch=2, seed=108, random=0, mode="ring", ef=True, steps=2)
    psq_case("ring_fcn_u2_qsgd_ef", fcn, users=2, epoch=1, seed=109, random=0, mode="ring", ef=True,
             quantizer="qsgd", c_dim=128, n_bit=2, steps=2)

    # ---- manifest ---------------------------------------------------------------
    man = {"torch": torch.__version__, "numpy": np.__version__, "files": {}}
    if ONLY:
        man = json.load(open(os.path.join(OUT, "MANIFEST.json")))
    for root, _, files in os.walk(OUT):
        for fn in sorted(files):
            if fn.endswith((".npz", ".npy", ".fvecs")) and (want(fn) or os.path.relpath(os.path.join(root, fn), OUT) not in man["files"]):
                p = os.path.join(root, fn)
                man["files"][os.path.relpath(p, OUT)] = hashlib.sha256(open(p, "rb").read()).hexdigest()
    json.dump(man, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(man["files"]), "fixtures to", OUT)


if __name__ == "__main__":
    main()
