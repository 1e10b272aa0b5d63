"""Compressor classes with the reference's names, constructor and method contract
(SURVEY.md 8b), backed by the HIP kernels in libgq_hsq.so.

    Compressor(size, shape, args).compress(vec) -> signature
    Compressor(...).decompress(signature)       -> tensor of `shape`

Signatures have the reference's structure and dtypes, so code that inspects them keeps
working:  HSQ  [ (lb, ub, levels int32[M]) | u f32[M] , codes uint8|int32 [M] ]
          QSGD [ norm f32[Mb,1], signs bool(shape), levels int32(shape) ].

No CPU path: `compress` / `decompress` raise if the tensor is not in MI355X HBM or the
HIP library is missing.  Construction (sub-dimension choice, codebook load) is host-side
NumPy exactly as in the reference and works anywhere.

Stochastic rounding (`args.random`, default True in the reference's CLI) needs uniform
draws.  The reference draws them with the CPU generator and copies them to the GPU every
call (probabilistic_scalar_compressor.py:23-25).  `args.gq_rng` (or $GQ_RNG) selects:
    "device"     (default) counter-based generator inside the kernel: same distribution,
                 no host round trip, different draws than the reference;
    "reference"  torch.rand on the CPU generator + H2D copy: bit-identical to the
                 reference for the same torch seed;
    "keyed"      as "device", but in the multi-tensor level kernels the stream of a tensor is keyed by the bits of
                 its (lb, ub) instead of a per-call seed: nothing in the launch changes from step to step, which is
                 what lets a quantizer step replay from a HIP graph (gq_graph, quantizers.py).
"""
import os

import numpy as np
import torch

from . import native
from .codebook import load_codebook, repaired_dim

_seed_counter = [0]


def _rng_mode(args):
    mode = getattr(args, "gq_rng", None) or os.environ.get("GQ_RNG", "device")
    if mode not in ("device", "reference", "keyed"):
        raise ValueError("gq_rng must be 'device', 'reference' or 'keyed', got %r" % (mode,))
    return mode


_seed_source = [None]      # a callable that replaces the process-local stream below while it is set (shared_seeds)


def _next_seed():
    """A fresh 63-bit seed per call for the in-kernel generator, derived from torch's CPU
    seed so that torch.manual_seed() makes runs reproducible."""
    if _seed_source[0] is not None:
        return _seed_source[0]()
    _seed_counter[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_counter[0] * 0xD1B54A32D192ED03) & (2 ** 63 - 1)


class shared_seeds(object):
    """with shared_seeds(source): every _next_seed() inside comes from `source()` instead of this process's torch seed and call
    count.  The parameter-server quantizer wraps the replicated second phase (ps_quantizer.py:52-61) in it with a source all ranks
    share, so that whatever path the re-compress takes (multi-tensor launches without device step words, per-tensor codecs, any
    compressor object) the ranks round identically -- whatever their own torch seeds are."""

    def __init__(self, source):
        self.source = source

    def __enter__(self):
        self.prev, _seed_source[0] = _seed_source[0], self.source

    def __exit__(self, *exc):
        _seed_source[0] = self.prev


def _require_device(t, what):
    if t.device.type != "cuda":
        raise native.GQNativeError(
            "%s: tensor is on %s; gq_amd runs on MI355X only (no CPU fallback). "
            "Run without --no-cuda on a GPU box." % (what, t.device))


class IdenticalCompressor(object):
    """Pass-through for small tensors and `--quantizer sgd` (identical_compressor.py:1-11)."""

    def __init__(self, size=None, shape=None, args=None):
        pass

    @staticmethod
    def compress(vec):
        return vec.clone()

    @staticmethod
    def decompress(signature):
        return signature


class ProbabilisticScalarCompressor(object):
    """Uniform 2^n_bit-level quantiser of a vector between its global min and max
    (probabilistic_scalar_compressor.py:4-33)."""

    def __init__(self, n_bit, args):
        self.n_bit = n_bit
        self.s = 2 ** n_bit
        self.cuda = not args.no_cuda
        self.code_dtype = torch.int32
        self.random = args.random
        self._rng = _rng_mode(args)

    # -- shared with NearestNeighborCompressor ---------------------------------------
    def _levels_from_partials(self, u, partials, levels, lb_ub):
        M = u.numel()
        if not self.random:
            native.hsq_levels(u, self.n_bit, native.RANDOM_OFF, None, 0, partials, lb_ub, levels)
        elif self._rng == "reference":
            r = torch.rand(M)  # CPU generator, as prob_scalar:23
            native.hsq_levels(u, self.n_bit, native.RANDOM_GIVEN, r.to(u.device), 0, partials, lb_ub, levels)
        else:
            native.hsq_levels(u, self.n_bit, native.RANDOM_DEVICE, None, _next_seed(), partials, lb_ub, levels)

    def compress(self, vec):
        _require_device(vec, "ProbabilisticScalarCompressor.compress")
        flat = vec.contiguous().view(-1)
        partials = native.new_workspace(flat.device, 0)
        native.minmax_partials(flat, partials)
        lb_ub = torch.empty(2, dtype=torch.float32, device=flat.device)
        levels = torch.empty(flat.numel(), dtype=self.code_dtype, device=flat.device)
        self._levels_from_partials(flat, partials, levels, lb_ub)
        return lb_ub[0], lb_ub[1], levels.view(vec.shape)

    def decompress(self, signature):
        """probabilistic_scalar_compressor.py:29-33 on the library's decode: float(l) * (ub - lb) / 2^n + lb, each step rounded on
        its own, is what gq_hsq_decode_sum computes for the norm of a subvector -- here times a one-by-one codebook holding 1.0
        (exact).  (Until round 6: three torch elementwise launches.)"""
        lower_bound, upper_bound, l = signature
        _require_device(l, "ProbabilisticScalarCompressor.decompress")
        dev = l.device
        flat = l.contiguous().view(-1)
        if flat.dtype not in (torch.uint8, torch.int16, torch.int32):
            flat = flat.to(torch.int32)
        M = flat.numel()
        unit = getattr(self, "_unit", None)
        if unit is None or unit[0].device != dev or unit[1].numel() < M:
            unit = self._unit = (torch.ones((1, 1), dtype=torch.float32, device=dev), torch.zeros(M, dtype=torch.uint8, device=dev))
        lb_ub = torch.stack([torch.as_tensor(lower_bound, dtype=torch.float32, device=dev).reshape(()),
                             torch.as_tensor(upper_bound, dtype=torch.float32, device=dev).reshape(())])
        out = torch.empty(M, dtype=torch.float32, device=dev)
        native.hsq_decode_sum(unit[1][:M], flat, lb_ub, unit[0], self.n_bit, out, R=1)
        return out.view(l.shape)


class NearestNeighborCompressor(object):
    """HSQ: per d-float subvector, the codeword with the largest |<c, v>| and the signed
    projection, then the scalar quantiser (nearest_neighbor_compressor.py:9-90)."""

    def __init__(self, size, shape, args):
        c_dim = args.c_dim
        k_bit = args.k_bit
        n_bit = args.n_bit
        assert c_dim > 0
        assert k_bit >= 0
        assert n_bit > 0

        self.cuda = not args.no_cuda
        self.size = size
        self.shape = shape
        self.dim = repaired_dim(size, c_dim)
        if c_dim != self.dim:
            print("alternate dimension form {} to {}, size {} shape {}".format(c_dim, self.dim, size, shape))
        assert size % self.dim == 0, \
            "not divisible size {}  c_dim {} self.dim {}".format(size, c_dim, self.dim)

        self.K = self.dim if k_bit <= 0 else 2 ** k_bit
        if self.K == self.dim:
            # random orthogonal codebook from NumPy's global generator, as :46
            from scipy import stats
            codewords = stats.ortho_group.rvs(self.dim).astype(np.float32)
        else:
            codewords = load_codebook(self.dim, self.K)
        self.codewords = torch.from_numpy(np.ascontiguousarray(codewords))
        self.code_dtype = torch.uint8 if k_bit <= 8 else torch.int32
        self.n_bit = n_bit
        self.compressed_norm = n_bit != 32
        if self.compressed_norm:
            self.norm_compressor = ProbabilisticScalarCompressor(n_bit, args)
        self.M = size // self.dim
        if self.cuda and torch.cuda.is_available():
            self.codewords = self.codewords.cuda()

    def _codebook_on(self, device):
        if self.codewords.device != device:
            self.codewords = self.codewords.to(device)
        return self.codewords

    # ---- reference protocol -----------------------------------------------------------
    def compress(self, vec):
        _require_device(vec, "NearestNeighborCompressor.compress")
        dev = vec.device
        flat = vec.contiguous().view(-1)
        assert flat.numel() == self.size
        cb = self._codebook_on(dev)
        M = self.M
        codes = torch.empty(M, dtype=self.code_dtype, device=dev)
        u = torch.empty(M, dtype=torch.float32, device=dev)
        partials = native.new_workspace(dev, M)
        native.hsq_encode(flat, cb, codes, u, partials)
        if not self.compressed_norm:
            return [u, codes]
        lb_ub = torch.empty(2, dtype=torch.float32, device=dev)
        levels = torch.empty(M, dtype=torch.int32, device=dev)
        self.norm_compressor._levels_from_partials(u, partials, levels, lb_ub)
        return [(lb_ub[0], lb_ub[1], levels), codes]

    def decompress(self, signature):
        norms, codes = signature
        _require_device(codes, "NearestNeighborCompressor.decompress")
        dev = codes.device
        cb = self._codebook_on(dev)
        codes = codes.contiguous().view(-1)
        if codes.dtype not in (torch.uint8, torch.int32):
            codes = codes.to(torch.int32)
        out = torch.empty(self.size, dtype=torch.float32, device=dev)
        if self.compressed_norm:
            lb, ub, levels = norms
            lb_ub = torch.stack([lb.reshape(()), ub.reshape(())]).to(device=dev, dtype=torch.float32)
            levels = levels.contiguous().view(-1)
            if levels.dtype not in (torch.uint8, torch.int16, torch.int32):
                levels = levels.to(torch.int32)
            native.hsq_decode_sum(codes, levels, lb_ub, cb, self.n_bit, out, R=1)
        else:
            native.hsq_decode_sum(codes, norms.contiguous().view(-1).float(), None, cb, 32, out, R=1)
        return out.view(self.shape)

    # ---- fused paths used by the quantizers (same results, fewer bytes / launches) ------
    def wire_level_dtype(self):
        """Narrowest level type that holds [0, 2^n_bit] (stochastic) / [0, 2^n_bit - 1]."""
        top = 2 ** self.n_bit - (0 if self.norm_compressor.random else 1)
        return torch.uint8 if top <= 255 else (torch.int16 if top <= 32767 else torch.int32)

    def compress_into(self, vec, codes, levels, lb_ub, u=None, partials=None):
        """compress() writing into caller-owned (wire) buffers; levels may be uint8."""
        _require_device(vec, "NearestNeighborCompressor.compress_into")
        assert self.compressed_norm
        dev = vec.device
        flat = vec.contiguous().view(-1)
        cb = self._codebook_on(dev)
        if u is None:
            u = torch.empty(self.M, dtype=torch.float32, device=dev)
        if partials is None:
            partials = native.new_workspace(dev, self.M)
        native.hsq_encode(flat, cb, codes, u, partials)
        self.norm_compressor._levels_from_partials(u, partials, levels, lb_ub)

    def roundtrip(self, vec):
        """decompress(compress(vec)) without the int32 signature round trip."""
        _require_device(vec, "NearestNeighborCompressor.roundtrip")
        if not self.compressed_norm:
            return self.decompress(self.compress(vec))
        dev = vec.device
        codes = torch.empty(self.M, dtype=self.code_dtype, device=dev)
        levels = torch.empty(self.M, dtype=self.wire_level_dtype(), device=dev)
        lb_ub = torch.empty(2, dtype=torch.float32, device=dev)
        self.compress_into(vec, codes, levels, lb_ub)
        out = torch.empty(self.size, dtype=torch.float32, device=dev)
        native.hsq_decode_sum(codes, levels, lb_ub, self._codebook_on(dev), self.n_bit, out, R=1)
        return out.view(self.shape)


class QSGDCompressor(object):
    """Bucketed max-norm stochastic quantiser (qsgd_compressor.py:4-71)."""

    def __init__(self, size, shape, args):
        self.random = args.random
        self.bit = args.n_bit
        c_dim = args.c_dim
        assert self.bit > 0
        self.cuda = not args.no_cuda
        self.s = 2 ** self.bit
        self.size = size
        self.shape = shape
        self.dim = repaired_dim(size, c_dim)
        if c_dim != self.dim:
            print("alternate dimension form {} to {}, size {} shape {}".format(c_dim, self.dim, size, shape))
        assert self.dim != 0, "0 sub dimension size {}  c_dim {} self.dim {}".format(size, c_dim, self.dim)
        assert size % self.dim == 0, "not divisible size {}  c_dim {} self.dim {}".format(size, c_dim, self.dim)
        self.M = size // self.dim
        self.code_dtype = torch.int32
        self._rng = _rng_mode(args)

    def compress(self, vec):
        _require_device(vec, "QSGDCompressor.compress")
        dev = vec.device
        flat = vec.contiguous().view(-1)
        norm = torch.empty(self.M, dtype=torch.float32, device=dev)
        signs = torch.empty(self.size, dtype=torch.bool, device=dev)
        levels = torch.empty(self.size, dtype=self.code_dtype, device=dev)
        if not self.random:
            native.qsgd_compress(flat, self.dim, self.bit, native.RANDOM_OFF, None, 0, norm, signs, levels)
        elif self._rng == "reference":
            r = torch.rand(self.M, self.dim)  # CPU generator, as qsgd_compressor.py:58
            native.qsgd_compress(flat, self.dim, self.bit, native.RANDOM_GIVEN, r.to(dev).view(-1), 0, norm, signs,
                                 levels)
        else:
            native.qsgd_compress(flat, self.dim, self.bit, native.RANDOM_DEVICE, None, _next_seed(), norm, signs,
                                 levels)
        return [norm.view(self.M, 1), signs.view(self.shape), levels.view(self.shape)]

    def decompress(self, signature):
        norm, signs, l = signature
        assert l.shape == signs.shape
        _require_device(l, "QSGDCompressor.decompress")
        out = torch.empty(self.size, dtype=torch.float32, device=l.device)
        lv = l.contiguous().view(-1)
        if lv.dtype not in (torch.uint8, torch.int32):
            lv = lv.to(torch.int32)
        native.qsgd_decode_sum(norm.contiguous().view(-1), signs.contiguous().view(-1), lv, self.dim, self.bit, out,
                               R=1)
        return out.view(self.shape)

    def roundtrip(self, vec):
        return self.decompress(self.compress(vec))


class ProbabilisticVectorCompressor(object):
    """Unbiased vector quantiser (probabilistic_vector_compressor.py:8-77): sample the codeword with probability
    |p_k| / ||p||_1, p = pinv(C^T) v, magnitude sign(p_k) * ||p||_1, so that E[decode] = v.
    Pinned by the reference's own output (tests/golden/pvq_*.npz, residual_*.npz): the class as it
    stands opens ./codebook/... (the tree has ./codebooks/learned_codebook/..., which is what is used here) and calls
    torch.argmin on a bool tensor (:58), which no torch with bool tensors implements; the fixtures were produced with
    that one operation defined as (index of the first True) - 1 -- the inverse-CDF sample the line's `+ 1` is written
    for -- and every other line running unedited.  gq_pvq_encode reproduces them bit for bit, including torch.cumsum's
    double accumulation of the cumulative probabilities."""

    def __init__(self, size, shape, args):
        c_dim, k_bit, n_bit = args.c_dim, args.k_bit, args.n_bit
        assert c_dim > 0
        assert k_bit > 0
        assert n_bit > 0
        self.cuda = not args.no_cuda
        self.size, self.shape = size, shape
        self.dim = c_dim if c_dim < size else size
        assert size % self.dim == 0, "not divisible size {} dim {}".format(size, self.dim)
        self.K = 2 ** k_bit
        if self.K == self.dim:
            from scipy import stats
            codewords = stats.ortho_group.rvs(self.dim).astype(np.float32)
        else:
            codewords = load_codebook(self.dim, self.K)
        self.codewords = torch.from_numpy(np.ascontiguousarray(codewords))
        self.c_dagger = torch.from_numpy(np.ascontiguousarray(np.linalg.pinv(codewords.T).astype(np.float32)))
        self.code_dtype = torch.uint8 if k_bit <= 8 else torch.int32
        self.n_bit = n_bit
        self.compressed_norm = n_bit != 32
        if self.compressed_norm:
            self.norm_compressor = ProbabilisticScalarCompressor(n_bit, args)
        self.M = size // self.dim
        self._rng = _rng_mode(args)

    def _on(self, device):
        if self.codewords.device != device:
            self.codewords = self.codewords.to(device)
            self.c_dagger = self.c_dagger.to(device)
        return self.codewords, self.c_dagger

    def compress(self, vec, minus=None):
        """minus = (codes1, norm1, codebook1): compress  vec - codebook1[codes1] * norm1  -- the residual of a first
        stage (ResidualCompressor) -- computed inside the kernel's tile staging instead of as a tensor."""
        _require_device(vec, "ProbabilisticVectorCompressor.compress")
        dev = vec.device
        flat = vec.contiguous().view(-1)
        _, cdag = self._on(dev)
        codes = torch.empty(self.M, dtype=self.code_dtype, device=dev)
        u = torch.empty(self.M, dtype=torch.float32, device=dev)
        ws = native.new_workspace(dev, 0)
        r, mode, seed = None, native.RANDOM_DEVICE, 0
        if self._rng == "reference":
            r, mode = torch.rand(self.M).to(dev), native.RANDOM_GIVEN       # CPU generator, as :52
        else:
            seed = _next_seed()
        if minus is None:
            native.pvq_encode(flat, cdag, codes, u, ws, mode, r, seed)
        else:
            codes1, norm1, cb1 = minus
            native.pvq_encode_residual(flat, codes1.contiguous().view(-1), norm1.contiguous().view(-1).float(), cb1, cdag,
                                       codes, u, ws, mode, r, seed)
        if not self.compressed_norm:
            return [u, codes]
        lb_ub = torch.empty(2, dtype=torch.float32, device=dev)
        levels = torch.empty(self.M, dtype=torch.int32, device=dev)
        self.norm_compressor._levels_from_partials(u, ws, levels, lb_ub)
        return [(lb_ub[0], lb_ub[1], levels), codes]

    def decompress(self, signature):
        norms, codes = signature
        _require_device(codes, "ProbabilisticVectorCompressor.decompress")
        dev = codes.device
        cb, _ = self._on(dev)
        codes = codes.contiguous().view(-1)
        if codes.dtype not in (torch.uint8, torch.int32):
            codes = codes.to(torch.int32)
        out = torch.empty(self.size, dtype=torch.float32, device=dev)
        if self.compressed_norm:
            lb, ub, levels = norms
            lb_ub = torch.stack([lb.reshape(()), ub.reshape(())]).to(device=dev, dtype=torch.float32)
            native.hsq_decode_sum(codes, levels.contiguous().view(-1), lb_ub, cb, self.n_bit, out, R=1)
        else:
            native.hsq_decode_sum(codes, norms.contiguous().view(-1).float(), None, cb, 32, out, R=1)
        return out.view(self.shape)


class ResidualCompressor(object):
    """Two stages (residual_compressor.py:7-32): NearestNeighbor on the gradient, the probabilistic vector
    compressor on what stage 1 leaves; decode = sum of the stage decodes.  The residual is never a tensor here:
    stage 2's kernel stages its tiles as  v - codebook1[code1] * norm1  (gq_pvq_encode_residual), so compress is
    stage 1's two launches + stage 2's, with two reads of the gradient instead of a clone, a decode, an in-place
    subtraction and a second encode over a materialised residual.  Pinned against the reference's own output
    (tests/golden/residual_*.npz)."""

    FUSED_MAX_DIM = 104      # the LDS-staged second-stage kernel; wider subvectors take the tensor path

    def __init__(self, size, shape, args):
        self.compressors = [
            NearestNeighborCompressor(size, shape, args),
            ProbabilisticVectorCompressor(size, shape, args),
        ]

    def compress(self, vec):
        first, second = self.compressors
        sig1 = first.compress(vec)
        if first.dim > self.FUSED_MAX_DIM or first.dim != second.dim:
            residuals = vec.clone()
            residuals -= first.decompress(sig1)
            return [sig1, second.compress(residuals)]
        norms, codes1 = sig1
        norm1 = first.norm_compressor.decompress(norms) if first.compressed_norm else norms     # probabilistic_scalar_compressor.py:31-32
        return [sig1, second.compress(vec, minus=(codes1, norm1, first._codebook_on(vec.device)))]

    def decompress(self, signatures):
        decoded = [c.decompress(sig) for sig, c in zip(signatures, self.compressors)]
        # the reference's reduction, not `a + b`: sum() starts from +0, so two -0 entries give +0 (:26-32)
        return torch.stack(decoded, dim=0).sum(dim=0)


# ---- exported for `from compressors import *` in the reference's main.py ----------------
# Not on the accelerated path: plain tensor ops, kept only so
# that main.py's `quantizer_choices` table resolves.

class SignSGDCompressor(object):
    """sign(v) (signsgd_compressor.py:4-12)."""

    def __init__(self, size, shape, args):
        pass

    def compress(self, vec):
        return torch.sign(vec)

    def decompress(self, signature):
        return signature


class TopKSparsificationCompressor(object):
    """Keep the size//cr largest-magnitude entries (topk_sparsification_compressor.py:9-26)."""

    def __init__(self, size, shape, args):
        self.cuda = not args.no_cuda
        self.size = size
        self.shape = shape
        self.users = 1
        self.k = size // args.cr

    def compress(self, vec):
        vec = vec.view(self.users, -1)
        keep = torch.zeros_like(vec)
        idx = torch.topk(torch.abs(vec), k=self.k, dim=1)[1]
        keep.scatter_(1, idx, 1)
        return vec * keep

    def decompress(self, signature):
        return signature.view(self.shape)


class MaureySparsification(object):
    """Maurey sampling (maurey_sparsification.py:4-50): k coordinates drawn with probability |v_i| / ||v||_1, each
    carrying sign(v_i) * ||v||_1 / k.  A name export like the two classes above (the reference does not wire it into
    main.py's table either): same constructor arithmetic for k, the draws come from torch.multinomial on the
    tensor's device instead of a k x size cumulative-sum comparison on the CPU -- the same distribution, other
    random numbers."""

    def __init__(self, size, shape, args):
        self.cr = 32 * args.c_dim // (args.k_bit + args.n_bit)
        bit_for_idx = 32 if size > 65536 else 16
        self.k = max(1, 32 * size // ((bit_for_idx + 1) * self.cr))
        self.cuda = not args.no_cuda
        self.size, self.shape = size, shape

    def compress(self, vec):
        flat = vec.reshape(-1)
        mag = flat.abs()
        l1_norm = mag.sum()
        codes = torch.multinomial(mag / l1_norm, self.k, replacement=True)
        return [l1_norm / self.k, codes, torch.sign(flat[codes])]

    def decompress(self, signature):
        scale, codes, signs = signature
        out = torch.zeros(self.size, dtype=signs.dtype, device=signs.device)
        out.index_add_(0, codes.reshape(-1).long(), signs.reshape(-1))
        return (scale * out).view(self.shape)


__all__ = ["IdenticalCompressor", "QSGDCompressor", "NearestNeighborCompressor", "ProbabilisticScalarCompressor",
           "ProbabilisticVectorCompressor", "ResidualCompressor", "SignSGDCompressor", "TopKSparsificationCompressor",
           "MaureySparsification"]
