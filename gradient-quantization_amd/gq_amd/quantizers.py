"""Quantizer layer with the reference's contract:

    q = Quantizer(Compressor, model.parameters(), args)      # args.mode in {'ps','ring'}
    for user in ...: loss.backward(); q.record(user, epoch=epoch)
    q.apply(); optimizer.step()

PSQuantizer mirrors quantizers/ps_quantizer.py:6-65, re-designed around a real wire:

* `record` compresses every gradient straight into this user's slot of ONE wire buffer
  (codes | levels | lb,ub per tensor; raw f32 for the <=1000-element tensors).  Nothing is
  decoded unless error feedback needs the residual.
* `apply` (alias `aggregate`) all-gathers the wire buffers of all ranks with ONE RCCL
  collective when torch.distributed is initialised (world_size > 1: every rank is one or
  more of the reference's `num_users`), then runs decode+mean on the GPU: payloads are
  summed in (rank, user) order and divided by their count -- the same arithmetic as the
  reference's torch.stack(decoded).mean(0) -- and `param.grad.data` is rebound to it.

The single-process case (no process group) is the reference's simulated-users loop with
identical results; the wire simply never leaves the GPU.

Codecs are the only objects that touch device memory; the product codecs call the HIP
library (gq_amd.native).  `codec_factory` exists so that the host logic above can be
exercised without a GPU by the tests (with the CPU oracle as the checker codec).
"""
import math
import operator
import os

import torch

from . import exchange, native
from .compressors import (IdenticalCompressor, NearestNeighborCompressor, QSGDCompressor, _next_seed,
                          _require_device)

# The C++ walks of the parameter list (csrc/host_ext.cpp -> gq_amd/_gq_host.so, built by build.py): the grads, their
# addresses as one bytes key and the "all plain f32" flag in one pass; `.data =` for all parameters in another.  A step
# whose launches replay from graphs is host-bound, and these walks are most of the host's work.  GQ_HOST_EXT=0: the
# Python walks (same results; tests/test_host_logic.py compares the two).
_HOST = None
if os.environ.get("GQ_HOST_EXT", "1") != "0":
    try:
        from . import _gq_host as _HOST
    except ImportError as _e:      # not built: the Python walks below do the same, slower; say so once
        import warnings
        warnings.warn("gq_amd: the host helper _gq_host.so is not built (%s); run gradient-quantization_amd/build.py" % (_e,))


def _up(x, a=16):
    return (x + a - 1) // a * a


# --------------------------------------------------------------------------------------
# Codecs: how one parameter tensor is written to / read from the wire
# --------------------------------------------------------------------------------------
class DenseCodec(object):
    """IdenticalCompressor tensors (<=1000 elements, ps_quantizer.py:18-19): raw f32 on the wire."""

    align = 4       # dense sections are packed back to back so that ONE cat / ONE mean serves them all

    def __init__(self, compressor, numel, shape):
        self.numel, self.shape = numel, shape
        self.nbytes = numel * 4

    def encode_into(self, grad, wire_user, off, salt):
        wire_user[off:off + self.numel * 4].view(torch.float32).copy_(grad.reshape(-1))

    def roundtrip(self, grad, salt):
        return grad.clone()

    def decode_mean(self, gathered, off, R, plain=False):
        # [R, numel] view of the gathered wire; stack().mean(0) of the reference (plain: the one payload as it is)
        rows = gathered[:, off:off + self.numel * 4].view(torch.float32)
        if plain and R == 1:
            return rows[0].clone().view(self.shape)
        if rows.device.type == "cuda":      # torch's GPU mean multiplies by 1/R and sums in its own order
            out = torch.empty(self.numel, dtype=torch.float32, device=rows.device)
            native.mean_rows(rows, out)
            return out.view(self.shape)
        return rows.mean(dim=0).view(self.shape)


def _kernel_copy(dst, src):
    """dst <- src (int64 device tensors of one size) by an elementwise KERNEL, for use under stream capture: a memcpy node in a
    replayed HIP graph costs ~15 us of every step whether its source is pinned host memory or device memory (84.9 / 83.4 us
    against 69.7 without the node, tools/graph_pieces.py -- the copy engine's hand-over), a kernel node ~2."""
    torch.bitwise_or(src, 0, out=dst)


def _esize(dtype):
    return torch.empty(0, dtype=dtype).element_size()


class GenericCodec(object):
    """Any other compressor class (sign, top-k, user supplied): ships the DECODED tensor.
    Keeps the reference semantics (mean of decompress(compress(g))) without a compact format."""

    def __init__(self, compressor, numel, shape):
        self.c, self.numel, self.shape = compressor, numel, shape
        self.nbytes = _up(numel * 4)

    def roundtrip(self, grad, salt):
        return self.c.decompress(self.c.compress(grad)).reshape(self.shape)

    def encode_into(self, grad, wire_user, off, salt):
        wire_user[off:off + self.numel * 4].view(torch.float32).copy_(self.roundtrip(grad, salt).reshape(-1))

    def decode_mean(self, gathered, off, R, plain=False):
        rows = gathered[:, off:off + self.numel * 4].view(torch.float32)
        if plain and R == 1:
            return rows[0].clone().view(self.shape)
        if rows.device.type == "cuda":      # torch's GPU mean multiplies by 1/R and sums in its own order
            out = torch.empty(self.numel, dtype=torch.float32, device=rows.device)
            native.mean_rows(rows, out)
            return out.view(self.shape)
        return rows.mean(dim=0).view(self.shape)


def aggregate_fma(args=None):
    """Opt-in (args.gq_aggregate = "fma" / $GQ_AGGREGATE=fma): the decode-mean over R >= 2 payloads accumulates with fused
    multiply-adds (GQ_AGGREGATE_FMA: half the arithmetic per payload, aggregate within 1e-6 relative L2 of the bit-exact one;
    the north star grants 1e-5).  Default "exact": the reference's separately rounded product and sum."""
    mode = getattr(args, "gq_aggregate", None) or os.environ.get("GQ_AGGREGATE", "exact")
    if mode not in ("exact", "fma"):
        raise ValueError("gq_aggregate / GQ_AGGREGATE must be 'exact' or 'fma', got %r" % (mode,))
    return mode == "fma"


def wire_levels_mode(args=None, world=1):
    """How byte-sized levels travel: "bytes" (one per level) or "packed6" (four 6-bit levels per three bytes, for the
    configurations whose top level is <= 63 with d = 16, K = 256).  args.gq_wire_levels, else $GQ_WIRE_LEVELS, else
    "auto": packed6 when there is an exchange to shorten (more than one rank), bytes on a single rank -- the packed form is
    bit-identical in its result and costs < 1 % of a single-rank step."""
    mode = getattr(args, "gq_wire_levels", None) or os.environ.get("GQ_WIRE_LEVELS", "auto")
    if mode not in ("bytes", "packed6", "auto"):
        raise ValueError("gq_wire_levels / GQ_WIRE_LEVELS must be 'bytes', 'packed6' or 'auto', got %r" % (mode,))
    if mode == "auto":
        mode = "packed6" if world > 1 else "bytes"
    return mode


class HSQCodec(object):
    """NearestNeighborCompressor on the HIP kernels.  Wire per user:
    codes[M] (uint8 | int32) | levels[M] (uint8/int16/int32, or f32 u when n_bit == 32; packed6: 3 * ceil(M/4) bytes) | lb, ub."""

    def __init__(self, compressor, numel, shape, packed6=False):
        self.c, self.numel, self.shape = compressor, numel, shape
        M = compressor.M
        self.M = M
        self.code_dtype = compressor.code_dtype
        self.level_dtype = compressor.wire_level_dtype() if compressor.compressed_norm else torch.float32
        self.packed6 = bool(packed6) and self.can_pack6(compressor)
        cb = torch.empty(0, dtype=self.code_dtype).element_size()
        self._level_bytes = native.packed6_bytes(M) if self.packed6 else M * torch.empty(0, dtype=self.level_dtype).element_size()
        self.codes_off = 0
        self.levels_off = _up(M * cb)
        self.lbub_off = self.levels_off + _up(self._level_bytes)
        self.nbytes = self.lbub_off + 16
        self._u = None
        self._partials = None

    @staticmethod
    def can_pack6(compressor):
        """The packed form is used for d = 16, K = 256 (what the multi-tensor level / decode kernels with packed levels are
        built for; the per-tensor entry points would take any K <= 256) when no level exceeds 63: n_bit <= 6 without
        stochastic rounding (probabilistic_scalar_compressor.py:18: levels up to 2^n_bit - 1), n_bit <= 5 with it (:25: up
        to 2^n_bit).  Every other configuration keeps one byte (or more) per level."""
        if not compressor.compressed_norm or compressor.dim != 16 or compressor.K != 256:
            return False
        nc = compressor.norm_compressor
        return (1 << nc.n_bit) - (0 if nc.random else 1) <= 63

    def wire_level_kind(self):
        """What the native calls take as the level type of this codec's wire."""
        return native.PACKED6 if self.packed6 else self.level_dtype

    def _views(self, wire_user, off):
        M = self.M
        cb = torch.empty(0, dtype=self.code_dtype).element_size()
        codes = wire_user[off + self.codes_off:off + self.codes_off + M * cb].view(self.code_dtype)
        levels = wire_user[off + self.levels_off:off + self.levels_off + self._level_bytes]
        if not self.packed6:
            levels = levels.view(self.level_dtype)
        lb_ub = wire_user[off + self.lbub_off:off + self.lbub_off + 8].view(torch.float32)
        return codes, levels, lb_ub

    def _scratch(self, dev):
        if self._u is None or self._u.device != dev:
            self._u = torch.empty(self.M, dtype=torch.float32, device=dev)
            self._partials = native.new_workspace(dev, self.M)
        return self._u, self._partials

    def uses_reference_draws(self):
        """True if compress draws r = torch.rand(M) from the CPU generator as the reference does
        (probabilistic_scalar_compressor.py:23-25; args.random with gq_rng = "reference")."""
        nc = getattr(self.c, "norm_compressor", None)
        return bool(self.c.compressed_norm and nc is not None and nc.random and nc._rng == "reference")

    def _levels(self, u, partials, levels, lb_ub, salt, r=None):
        nc = self.c.norm_compressor
        if not nc.random:
            native.hsq_levels(u, nc.n_bit, native.RANDOM_OFF, None, 0, partials, lb_ub, levels, self.packed6)
        elif nc._rng == "reference":
            if r is None:       # the quantizer hands over its slice of ONE torch.rand per record (same stream)
                r = torch.rand(self.M).to(u.device)
            native.hsq_levels(u, nc.n_bit, native.RANDOM_GIVEN, r, 0, partials, lb_ub, levels, self.packed6)
        else:
            native.hsq_levels(u, nc.n_bit, native.RANDOM_DEVICE, None, _next_seed() ^ salt, partials, lb_ub, levels, self.packed6)

    def encode_into(self, grad, wire_user, off, salt, r=None):
        _require_device(grad, "HSQCodec.encode_into")
        dev = grad.device
        flat = grad.contiguous().view(-1)
        codes, levels, lb_ub = self._views(wire_user, off)
        cbk = self.c._codebook_on(dev)
        if self.c.compressed_norm:
            u, partials = self._scratch(dev)
            nc = self.c.norm_compressor
            if nc.random and nc._rng == "reference":
                native.hsq_encode(flat, cbk, codes, u, partials)
                self._levels(u, partials, levels, lb_ub, salt, r)
            else:   # encode + levels in one library call (gq_hsq_compress)
                mode = native.RANDOM_DEVICE if nc.random else native.RANDOM_OFF
                native.hsq_compress(flat, cbk, codes, u, partials, nc.n_bit, mode, None,
                                    (_next_seed() ^ salt) if nc.random else 0, lb_ub, levels, self.packed6)
        else:
            _, partials = self._scratch(dev)
            native.hsq_encode(flat, cbk, codes, levels, partials)  # `levels` section holds f32 u

    def decode_wire(self, wire_user, off, out):
        """Decode this user's own payload (error feedback residual)."""
        self._decode(wire_user.view(1, -1), off, 1, out)

    def encode_decode_into(self, grad, wire_user, off, salt, out, r=None):
        """encode_into + decode_wire: decompress(compress(grad)) with the payload left in the wire (ps_quantizer.py:37).
        Where the library serves it (d = 16, byte codes, byte or packed levels) the level quantiser and the decode are ONE
        launch (gq_hsq_levels_decode); same bits either way."""
        nc = getattr(self.c, "norm_compressor", None)
        if self.c.compressed_norm and self.c.dim == 16 and self.code_dtype == torch.uint8 and grad.device.type == "cuda":
            dev = grad.device
            flat = grad.contiguous().view(-1)
            codes, levels, lb_ub = self._views(wire_user, off)
            if self.packed6 or levels.dtype == torch.uint8:
                cbk = self.c._codebook_on(dev)
                u, partials = self._scratch(dev)
                if not nc.random:
                    mode, rr, seed = native.RANDOM_OFF, None, 0
                elif nc._rng == "reference":
                    mode, rr, seed = native.RANDOM_GIVEN, (r if r is not None else torch.rand(self.M).to(dev)), 0
                else:
                    mode, rr, seed = native.RANDOM_DEVICE, None, _next_seed() ^ salt
                native.hsq_encode(flat, cbk, codes, u, partials)
                if native.hsq_levels_decode(u, nc.n_bit, mode, rr, seed, partials, lb_ub, levels, codes, cbk, out, self.packed6):
                    return
                native.hsq_levels(u, nc.n_bit, mode, rr, seed, partials, lb_ub, levels, self.packed6)
                self.decode_wire(wire_user, off, out)
                return
        self.encode_into(grad, wire_user, off, salt, r)
        self.decode_wire(wire_user, off, out)

    def _decode(self, gathered, off, R, out):
        P = gathered.shape[1]
        cbk = self.c._codebook_on(gathered.device)
        n_bit = self.c.n_bit if self.c.compressed_norm else 32
        if getattr(self, "fma", False) and R >= 2 and self.c.compressed_norm:
            n_bit |= native.AGGREGATE_FMA
        native.hsq_decode_sum_packed(gathered, self.M, cbk, n_bit, out, R,
                                     codes_off=off + self.codes_off, levels_off=off + self.levels_off,
                                     lbub_off=off + self.lbub_off, code_dtype=self.code_dtype,
                                     level_dtype=self.wire_level_kind())
        assert P == gathered.stride(0)

    def roundtrip(self, grad, salt, r=None):
        dev = grad.device
        tmp = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.encode_decode_into(grad, tmp, 0, salt, out, r)
        return out.view(self.shape)

    def decode_mean(self, gathered, off, R, plain=False):
        out = torch.empty(self.numel, dtype=torch.float32, device=gathered.device)
        self._decode(gathered, off, R, out)
        if R == 1 and not plain:
            out.add_(0.0)   # one payload is the plain decompress (-0 kept); the aggregate is a sum that starts from +0
        return out.view(self.shape)


class QSGDCodec(object):
    """QSGDCompressor on the HIP kernels.  Wire per user, packed form (even bucket size and a
    top level that fits 3, 7 or 15 bits):  norm f32[Mb] | one code per element = sign<<(bits-1) | level,
    4-bit codes two per byte.  Otherwise the plain form  norm f32[Mb] | signs u8[n] | levels u8|i32 [n]."""

    def __init__(self, compressor, numel, shape):
        self.c, self.numel, self.shape = compressor, numel, shape
        self.Mb, self.d = compressor.M, compressor.dim
        mode = native.RANDOM_DEVICE if compressor.random else native.RANDOM_OFF
        self.bits = 0
        if self.d % 2 == 0 and (not compressor.random or compressor._rng != "reference"):
            top = 2 ** compressor.bit - (0 if compressor.random else 1)
            self.bits = 4 if top <= 7 else (8 if top <= 127 else (16 if top <= 32767 else 0))
        self.norm_off = 0
        if self.bits:
            self.codes_off = _up(self.Mb * 4)
            self.nbytes = self.codes_off + _up(numel * self.bits // 8)
            self._single = None     # a one-tensor BatchedQSGD, built on first use
        else:
            top = 2 ** compressor.bit
            self.level_dtype = torch.uint8 if top <= 127 else torch.int32
            lb = torch.empty(0, dtype=self.level_dtype).element_size()
            self.signs_off = _up(self.Mb * 4)
            self.levels_off = self.signs_off + _up(numel)
            self.nbytes = self.levels_off + _up(numel * lb)
        self._mode = mode

    # ---- packed form: a single-segment instance of the batched kernels ----------------------
    def _batched1(self, dev):
        if self._single is None or self._single.device != dev:
            self._single = BatchedQSGD([self], [0], [0], dev, 1, self.nbytes)
        return self._single

    # ---- plain form --------------------------------------------------------------------------
    def _views(self, wire_user, off):
        lb = torch.empty(0, dtype=self.level_dtype).element_size()
        norm = wire_user[off + self.norm_off:off + self.norm_off + self.Mb * 4].view(torch.float32)
        signs = wire_user[off + self.signs_off:off + self.signs_off + self.numel]
        levels = wire_user[off + self.levels_off:off + self.levels_off + self.numel * lb].view(self.level_dtype)
        return norm, signs, levels

    def encode_into(self, grad, wire_user, off, salt):
        _require_device(grad, "QSGDCodec.encode_into")
        flat = grad.contiguous().view(-1)
        c = self.c
        if self.bits:
            ok = self._batched1(flat.device).encode([flat], wire_user[off:off + self.nbytes], 0, salt)
            assert ok, "QSGDCodec: gradient storage must be 8-byte aligned"
            return
        norm, signs, levels = self._views(wire_user, off)
        if not c.random:
            native.qsgd_compress(flat, self.d, c.bit, native.RANDOM_OFF, None, 0, norm, signs, levels)
        elif c._rng == "reference":
            r = torch.rand(self.Mb, self.d)
            native.qsgd_compress(flat, self.d, c.bit, native.RANDOM_GIVEN, r.to(flat.device).view(-1), 0, norm, signs,
                                 levels)
        else:
            native.qsgd_compress(flat, self.d, c.bit, native.RANDOM_DEVICE, None, _next_seed() ^ salt, norm, signs,
                                 levels)

    def _decode_rows(self, gathered, off, R, out, plain=False):
        if self.bits:
            rows = gathered[:, off:off + self.nbytes]
            if not rows.is_contiguous():
                rows = rows.contiguous()
            out.copy_(self._batched1(gathered.device).decode_mean(rows, R, plain=plain)[0].view(-1))
            return
        # the plain entry point takes dense [R][...] arrays: gather the three sections
        lb = torch.empty(0, dtype=self.level_dtype).element_size()
        norm = gathered[:, off + self.norm_off:off + self.norm_off + self.Mb * 4].contiguous().view(torch.float32)
        signs = gathered[:, off + self.signs_off:off + self.signs_off + self.numel].contiguous()
        levels = gathered[:, off + self.levels_off:off + self.levels_off + self.numel * lb].contiguous() \
            .view(self.level_dtype)
        native.qsgd_decode_sum(norm.view(-1), signs.view(-1), levels.view(-1), self.d, self.c.bit, out, R=R)

    def roundtrip(self, grad, salt):
        tmp = torch.empty(self.nbytes, dtype=torch.uint8, device=grad.device)
        self.encode_into(grad, tmp, 0, salt)
        out = torch.empty(self.numel, dtype=torch.float32, device=grad.device)
        self._decode_rows(tmp.view(1, -1), 0, 1, out, plain=True)     # decompress(compress(g)): no aggregate
        return out.view(self.shape)

    def decode_wire(self, wire_user, off, out):
        self._decode_rows(wire_user.view(1, -1), off, 1, out, plain=True)

    def decode_mean(self, gathered, off, R, plain=False):
        out = torch.empty(self.numel, dtype=torch.float32, device=gathered.device)
        self._decode_rows(gathered, off, R, out, plain=plain)
        if R == 1 and not plain:
            out.add_(0.0)   # as HSQCodec.decode_mean: torch.stack(...).mean(0) of one payload turns -0 into +0
        return out.view(self.shape)


_DATA_PTR = torch.Tensor.data_ptr
_IS_CONTIGUOUS = torch.Tensor.is_contiguous
_DTYPE_OF = operator.attrgetter("dtype")
_GET_DEVICE = torch.Tensor.get_device      # the device index (-1 for a CPU tensor)
_F32_ONLY = {torch.float32}


class _BatchedBase(object):
    """Shared plumbing of the multi-tensor kernels: a per-step header (segment table with the
    tensors' current device pointers, plus kernel-specific reset values) goes to the device in ONE
    pinned H2D copy; a ring of pinned buffers (one per user slot + one) keeps a copy in flight from
    being overwritten."""

    def _setup(self, table, extra, device, slots, user_bytes, dense=None):
        """dense: [(byte offset in one user's wire, elements), ...] of the identity-compressed tensors this group's compress
        launch also copies into the wire (the quantizer gives them to its first group), or None."""
        self.nseg = table.shape[0]
        self.device = device
        self.user_bytes = user_bytes
        self._table_words = self.nseg * 8
        host = torch.cat([table.view(-1), extra.view(-1)]) if extra is not None else table.view(-1).clone()
        self.ndense = len(dense) if dense else 0
        self._dense_at = int(host.numel())      # the dense table's first word in the header
        if self.ndense:
            dt = torch.zeros((self.ndense, 3), dtype=torch.int64)
            for k, (off, numel) in enumerate(dense):
                dt[k, 1], dt[k, 2] = off, numel
            host = torch.cat([host, dt.view(-1)])
        self._host = [host.clone().pin_memory() for _ in range(slots + 1)]
        self._host_np = [h[:self._table_words].view(self.nseg, 8).numpy() for h in self._host]   # views of the pinned tables
        self._host_dense_np = [h[self._dense_at:].view(self.ndense, 3).numpy() for h in self._host] if self.ndense else None
        self._last_dptrs = None
        self._zeros = [0] * self.nseg
        self._resets = extra is not None    # the header also carries per-step reset values (min / max accumulators)
        self._acc_init = extra.view(-1).to(device) if extra is not None else None     # the accumulators' empty state, on the device
        self._acc_clean = False             # the device accumulators are in that state right now (see _graph_tables)
        self._last_ptrs = self._last_eptrs = None
        self._events = [None] * (slots + 1)
        self._dev = torch.empty_like(host, device=device)
        self._tmp_wire = None
        self.ready = False      # the device header has been written at least once
        self._outs, self._out_views, self._out_turn = [None, None], [None, None], 0
        self._layout = table.clone()    # host copy of the segment table without pointers (decode needs no pointers)
        self._parts = {}                # (first tensor, end) -> launch descriptor of one chunk of a split / pipelined decode
        self.rng_pairs = None           # this group's { seed, step } pairs, one per user slot (PSQuantizer._rng_pairs_for)

    def _out_buffer(self, device, advance=True):
        """Decode target + its per-tensor views.  Two buffers used in turn (the mean and its two-phase
        re-decode never alias; last step's gradients stay intact for one more apply) and the 76+
        slice/view objects of a model are built once instead of every step.  advance=False: the buffer
        of the previous call again (second part of a split decode)."""
        if not advance:
            k = self._out_turn ^ 1
            return self._outs[k], self._out_views[k]
        k = self._out_turn
        self._out_turn ^= 1
        if self._outs[k] is None or self._outs[k].device != device:
            out = torch.empty(self.out_floats, dtype=torch.float32, device=device)
            self._outs[k] = out
            self._out_views[k] = [out[o:o + cd.numel].view(cd.shape) for o, cd in zip(self.out_off, self.codecs)]
        return self._outs[k], self._out_views[k]

    def dense_table_dev(self):
        return self._dev[self._dense_at:].view(self.ndense, 3) if self.ndense else None

    # ---- launches under stream capture: a graph's own tables, accumulators reset behind their last reader -----------------
    # A captured record reads the segment / dense tables from a device copy that belongs to the graph (nobody rewrites
    # it), so a replay needs no header copy in front of the encode -- any node there, memcpy or kernel, cost ~7 us of every
    # step (profiles/r04_graph_pieces.txt).  What the header copy also did, resetting the accumulators the kernels fold into
    # ((min, max) per tensor; wide QSGD buckets' norms), is a small kernel BEHIND the group's last launch instead: a graph
    # leaves them clean for the next replay, an eager step leaves them used (`_acc_clean`), and whoever replays a graph
    # after an eager step cleans them first (ensure_clean).
    def _graph_tables(self, graph_header, dense):
        self._batch.set_table(graph_header[:self._table_words])
        self._batch.set_dense(graph_header[self._dense_at:].view(self.ndense, 3) if (dense is not None and self.ndense) else None,
                              self.ndense)

    def _graph_tables_done(self, defer=None):
        """defer (a list): the reset is left to the caller -- (accumulators, their empty state) is appended -- who folds it
        into a launch that runs anyway behind this group's last one (the aggregate's gq_mean_rows in a whole-step graph)."""
        if self._resets:
            if defer is not None:
                defer.append((self._dev[self._table_words:self._dense_at], self._acc_init))
            else:
                _kernel_copy(self._dev[self._table_words:self._dense_at], self._acc_init)
        self._batch.set_table(self._dev[:self._table_words])

    def _graph_tables_abort(self):
        """A launch failed between _graph_tables and _graph_tables_done (an invalidated capture, a launch error): the callers
        fall back to eager launches, which must not read their pointers from the graph's header.  The descriptor goes back to
        the shared device header and the next eager encode re-validates and re-sends it (no reset kernel: the stream may be
        in a broken capture; `_acc_clean = False` makes the next replay clean the accumulators first)."""
        self._batch.set_table(self._dev[:self._table_words])
        self._batch.set_dense(self.dense_table_dev(), self.ndense)
        self._last_ptrs = self._last_eptrs = self._last_dptrs = None
        self._acc_clean = False

    def ensure_clean(self):
        if self._resets and not self._acc_clean:
            _kernel_copy(self._dev[self._table_words:self._dense_at], self._acc_init)
            self._acc_clean = True

    def _upload(self, tensors, slot, align, errs=None, dense=None):
        """Column 0 of the segment table <- the tensors' device pointers; column 7 <- the error
        buffers' (error-feedback kernels) or 0.  False if any tensor cannot be addressed that way.
        A header that also carries the reset values of the kernels' min / max accumulators (HSQ, wide-bucket QSGD)
        goes to the device every time; when the pointers are the ones of the last upload (gradients that keep their storage from
        step to step) the pinned copy is sent as it is, without checking and rewriting the table."""
        ptrs = list(map(_DATA_PTR, tensors))
        eptrs = list(map(_DATA_PTR, errs)) if errs is not None else self._zeros
        dptrs = list(map(_DATA_PTR, dense)) if dense is not None else None
        # the fast path still checks what the kernels assume about every tensor: a gradient replaced by a strided view or
        # another dtype AT THE SAME ADDRESS (channels_last, the caching allocator handing the block out again) must not
        # ride on the last upload's validation.  (map() over the C-level accessors: ~6 us for 76 tensors; a Python-level
        # list of (dtype, is_contiguous) tuples cost 20.)
        if (self.ready and ptrs == self._last_ptrs and eptrs == self._last_eptrs and dptrs == self._last_dptrs
                and all(map(_IS_CONTIGUOUS, tensors)) and set(map(_DTYPE_OF, tensors)) == _F32_ONLY
                and (dense is None or (all(map(_IS_CONTIGUOUS, dense)) and set(map(_DTYPE_OF, dense)) == _F32_ONLY))):
            if not self._resets:
                return True     # nothing but the table in this header, and the device copy still holds it
            self._dev.copy_(self._host[self._last_slot], non_blocking=True)     # unchanged since its last copy
            self._events[self._last_slot].record()     # a later rewrite of this pinned buffer waits for this copy too
            return True
        # (the same facts for a new set of pointers, from the C-level accessors: a Python loop over
        # `g.device != ... or g.dtype != ...` cost 40 us for 76 tensors, most of it building torch.device objects)
        dev_index = self.device.index if self.device.index is not None else torch._C._cuda_getDevice()
        for ts, ps, al in ((tensors, ptrs, align), (errs or (), eptrs if errs is not None else (), align), (dense or (), dptrs or (), 4)):
            if not ts:
                continue
            if (not all(map(_IS_CONTIGUOUS, ts)) or set(map(_DTYPE_OF, ts)) != _F32_ONLY
                    or set(map(_GET_DEVICE, ts)) != {dev_index} or any(p % al for p in ps)):
                return False
        if dense is not None and len(dense) != self.ndense:
            return False
        if len(eptrs) != len(ptrs):
            return False
        slot %= len(self._host)
        if self._events[slot] is not None:
            self._events[slot].synchronize()       # the previous copy out of this pinned buffer
        tab = self._host_np[slot]
        tab[:, 0] = ptrs
        tab[:, 7] = eptrs
        if dptrs is not None:
            self._host_dense_np[slot][:, 0] = dptrs
        self._dev.copy_(self._host[slot], non_blocking=True)
        self.ready = True
        self._last_ptrs, self._last_eptrs, self._last_slot, self._last_dptrs = ptrs, eptrs, slot, dptrs
        if self._events[slot] is None:
            self._events[slot] = torch.cuda.Event()
        self._events[slot].record()
        return True

    def _counter_seed(self, slot):
        """GQ_RANDOM_DEVICE_COUNTER: the address of this group's { seed, step } pair of user slot `slot`, or None when the
        quantizer gave the group no pairs (a codec used on its own) or not enough of them."""
        if self.rng_pairs is None or not 0 <= slot < self.rng_pairs.shape[0]:
            return None
        return self.rng_pairs.data_ptr() + 16 * slot

    def _range(self, lo, hi):
        """Launch descriptor of the tensors [lo, hi) of the group (one chunk of a split / pipelined decode, PSQuantizer.apply
        under GQ_EXCHANGE=split|pipelined).  lo == 0: the same device table, fewer items; otherwise a table of its own, built
        once (segment and item indices restart at zero).  None when the range is empty."""
        if lo >= hi:
            return None
        ent = self._parts.get((lo, hi))
        if ent is None:
            i_lo = int(self._layout[lo, 2])
            i_hi = int(self._layout[hi, 2]) if hi < self.nseg else self._nitems
            if lo == 0:
                ent = self._batch.part(self._dev[:self._table_words], self._item_seg, hi, i_hi)
            else:
                tab = self._layout[lo:hi].clone()
                tab[:, 2] -= i_lo
                items = (self._item_seg[i_lo:i_hi] - lo).contiguous()
                ent = self._batch.part(tab.view(-1).to(self.device), items, hi - lo, i_hi - i_lo)
            self._parts[(lo, hi)] = ent
        return ent

    def decode_mean(self, gathered, R, part=None, plain=False, tail=None):
        """Mean of the R payloads of `gathered` for every tensor of the group (views of one output buffer).
        part = (lo, hi, first): only the tensors [lo, hi) of the group -- the chunks of a split / pipelined exchange land in
        the same buffer, `first` on the first of them (it takes the next output buffer, the others write into it too).
        plain: the decompress of ONE payload as the reference returns it (a -0 stays -0) instead of the aggregate."""
        if not self.ready:      # a rank that decodes before it has encoded anything (ring hop, late joiner)
            self.upload_layout()
        out, views = self._out_buffer(gathered.device, advance=part is None or part[2])
        batch = self._batch if part is None else self._range(part[0], part[1])
        if batch is not None:
            kw = {"tail": tail} if tail is not None else {}      # (BatchedHSQ only: see takes_tail)
            if getattr(self, "fma", False) and R >= 2 and not plain:
                batch.decode(gathered, R, out, fma=True, **kw)      # (BatchedHSQ only: the quantizer sets `fma` on its HSQ groups)
            else:
                batch.decode(gathered, R, out, plain=plain, **kw)
        return views

    takes_tail = False      # the group's decode-mean launch can take the aggregate's small per-step work along (native.StepTail)

    def upload_layout(self):
        """Device header with the layout columns only (no tensor pointers): enough for decode_mean,
        which a ring rank may need before it has encoded anything."""
        self._last_ptrs = self._last_eptrs = self._last_dptrs = None
        self._dev.copy_(self._host[0], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[0] = ev
        self.ready = True

    def roundtrip(self, tensors, slot, salt, errs=None, ef_scale=None, draws=None, rng_slot=None):
        """decompress(compress(t)) for every batched tensor in a few launches; None if not batchable.
        With `errs`: t <- t + ef_scale*err in place first and err <- t - decoded afterwards.
        rng_slot: the { seed, step } pair the stochastic rounding draws from (default: the one of `slot`)."""
        if self._tmp_wire is None:
            self._tmp_wire = torch.zeros((1, self.user_bytes), dtype=torch.uint8, device=self.device)
        if not self.encode(tensors, self._tmp_wire[0], slot, salt, errs, ef_scale, draws=draws, rng_slot=rng_slot):
            return None
        return self.decode_mean(self._tmp_wire, 1, plain=True)     # decompress(compress(t)) (ps_quantizer.py:52-61): a -0 stays -0


class BatchedHSQ(_BatchedBase):
    """All NearestNeighborCompressor tensors that share a codebook are encoded by ONE encode + ONE levels
    launch and decoded by ONE decode-mean launch (per-tensor lb / ub, identical results).  The reference
    walks the parameter list in Python (ps_quantizer.py:33,47); ResNet-50 has 76 such tensors.
    The library decides which kernels serve the group's shape (include/gq_hsq.h, gq_hsq_batched_path): K = 256 with
    d = 8 / 16 / 32 and byte-sized codes the prefilter encode and the specialised levels / decode kernels, larger
    codebooks of those dimensions the paged prefilter, every other shape exact scoring."""

    takes_tail = True      # gq_hsq_decode_sum_batched_tail

    @staticmethod
    def eligible(codec):
        c = getattr(codec, "c", None)
        if type(codec) is not HSQCodec or c.K == c.dim:   # K == d: a random codebook per tensor
            return False
        return native.hsq_batched_path(c.dim, c.K, codec.code_dtype) != 0

    @staticmethod
    def group_key(codec):
        return (codec.c.dim, codec.c.K, _esize(codec.code_dtype), _esize(codec.level_dtype), int(codec.c.n_bit), int(codec.packed6))

    def __init__(self, codecs, offsets, idxs, device, slots, user_bytes, dense=None):
        self.idxs = list(idxs)
        self.codecs = [codecs[i] for i in self.idxs]
        c0 = self.codecs[0].c
        self.n_bit = c0.n_bit                                  # 32: the projections travel as f32 (no level quantiser)
        self.random = bool(c0.compressed_norm and c0.norm_compressor.random)
        self.keyed = bool(self.random and c0.norm_compressor._rng == "keyed")   # draws keyed by (lb, ub): a launch that never changes
        self.counter = bool(self.random and c0.norm_compressor._rng == "device")  # draws keyed by a device step word: likewise, and fresh every step
        self.reference_draws = self.codecs[0].uses_reference_draws()     # the reference's CPU draws, handed in per record
        self._r_index = self._r_flat = None
        self.codebook = c0._codebook_on(device)
        nseg = len(self.idxs)
        table = torch.zeros((nseg, 8), dtype=torch.int64)
        tile_seg = []
        tile, out_off = 0, 0
        self.out_off = []
        for s, (i, cd) in enumerate(zip(self.idxs, self.codecs)):
            ntile = (cd.M + 63) // 64
            table[s, 1], table[s, 2] = cd.M, tile
            table[s, 3] = offsets[i] + cd.codes_off
            table[s, 4] = offsets[i] + cd.levels_off
            table[s, 5] = offsets[i] + cd.lbub_off
            table[s, 6] = out_off
            tile_seg += [s] * ntile
            tile += ntile
            self.out_off.append(out_off)
            out_off += cd.numel
        self.ntiles, self.out_floats = tile, out_off
        self.tile_seg = torch.tensor(tile_seg, dtype=torch.int32, device=device)
        self._item_seg, self._nitems = self.tile_seg, tile
        init = torch.empty((nseg, 2), dtype=torch.int32)
        init[:, 0], init[:, 1] = -1, 0            # 0xFFFFFFFF / 0: identities of the mapped min / max
        self._setup(table, init.view(torch.int64), device, slots, user_bytes, dense)
        self.u_flat = torch.empty(self.ntiles * 64, dtype=torch.float32, device=device)
        cd0 = self.codecs[0]
        self.code_dtype, self.level_dtype = cd0.code_dtype, cd0.level_dtype
        self.align = 16 if c0.dim % 4 == 0 else 4
        self.ws = native.new_workspace(device, self.ntiles * 64)
        # ONE launch descriptor for the group (gq_hsq_batch): the library picks the kernels -- prefilter (K = 256,
        # d = 8 / 16 / 32), the same with the pages of a larger codebook resident, or exact scoring for every other shape
        self._batch = native.HSQBatch(self._dev[:self._table_words], self.tile_seg, self.nseg, self.ntiles, self.codebook,
                                      self.code_dtype, cd0.wire_level_kind(), self.n_bit, self.u_flat,
                                      self._dev[self._table_words:self._dense_at].view(torch.int32), self.ws)
        self.profile_slot = -1      # measurement hook (bench.py): the NEXT encode's dispatch is timed into this slot

    def _given_draws(self, draws):
        """draws = (r_all on the device, {parameter index: offset of its M draws}): the reference's
        torch.rand(M) per tensor, drawn by the quantizer in ONE call per record.  Laid out like u_flat for
        the level kernels (one gather through an index built once)."""
        r_all, offsets = draws
        if self._r_index is None:
            idx = torch.zeros(self.ntiles * 64, dtype=torch.int64)
            for s, (i, cd) in enumerate(zip(self.idxs, self.codecs)):
                first = int(self._layout[s, 2]) * 64
                idx[first:first + cd.M] = torch.arange(offsets[i], offsets[i] + cd.M)
            self._r_index = idx.to(self.device)
            self._r_flat = torch.empty(self.ntiles * 64, dtype=torch.float32, device=self.device)
        torch.index_select(r_all, 0, self._r_index, out=self._r_flat)
        return self._r_flat

    def graphable(self):
        """True when nothing in this group's launches changes from record to record for fixed gradient addresses (no
        per-call seed, no host-side draws): the launches can be nodes of a HIP graph (PSQuantizer, gq_graph)."""
        return (not self.random or self.keyed or (self.counter and self.rng_pairs is not None)) and not self.reference_draws \
            and self._batch.path != 0

    def encode(self, tensors, wire_user, slot, salt, errs=None, ef_scale=None, draws=None, graph_header=None, dense=None, defer_reset=None,
               rng_slot=None, skip_levels=False):
        """Compress `tensors` (one per batched parameter, in order) into one user's wire.
        skip_levels (whole-step capture at one rank and one user): only the encode is launched; the level launch is left to
        decode_mean(..., fused_levels=True), which runs it together with the decode (gq_hsq_levels_decode_batched).
        dense: the identity-compressed tensors (the quantizer's, in its order) that the level launch also copies into the wire.
        Returns False (nothing launched) when a tensor is not a contiguous, 16-byte aligned f32
        tensor on this device: the caller then takes the per-tensor path for this step.
        With `errs` (error feedback, ps_quantizer.py:34-39) the same launches also do
        t += ef_scale*err (in place, before encoding) and err = t - decoded (in place, after).
        graph_header (stream capture): a device copy of the header of exactly these tensors that nobody rewrites; it is
        copied instead of the shared pinned buffers (no events, no validation: the caller has just run the same call eagerly)."""
        if self.reference_draws and draws is None:
            return False
        if self._batch.path == 0:       # e.g. more than 384 tensors of d = 8 / 32 and no exact kernel for the shape
            return False
        if graph_header is not None:
            self._graph_tables(graph_header, dense)
        elif not self._upload(tensors, slot, self.align, errs, dense):
            return False
        else:
            self._batch.set_dense(self.dense_table_dev() if dense is not None else None, self.ndense)
            self._acc_clean = False
        ef = ef_scale if errs is not None else None
        rng_slot = slot if rng_slot is None else rng_slot
        try:
            self._batch.encode(wire_user, ef, self.profile_slot)
            self.profile_slot = -1
            if self.n_bit == 32:
                mode, seed, r_flat = native.RANDOM_OFF, 0, None
            elif self.reference_draws:
                mode, seed, r_flat = native.RANDOM_GIVEN, 0, self._given_draws(draws)
            elif self.keyed:
                mode, seed, r_flat = native.RANDOM_DEVICE_KEYED, (salt * 0x2545F4914F6CDD1D + 0x5851F42D4C957F2D) & (2 ** 63 - 1), None
            elif self.counter and self._counter_seed(rng_slot) is not None:
                mode, seed, r_flat = native.RANDOM_DEVICE_COUNTER, self._counter_seed(rng_slot), None
            elif self.random:
                mode, seed, r_flat = native.RANDOM_DEVICE, _next_seed() ^ salt, None
            else:
                mode, seed, r_flat = native.RANDOM_OFF, 0, None
            if skip_levels:
                self._pending_levels = (wire_user, mode, seed, r_flat, errs is not None)
            else:
                self._batch.levels(wire_user, mode, seed, r_flat, write_error=errs is not None)
        except BaseException:
            if graph_header is not None:
                self._graph_tables_abort()
            raise
        if graph_header is not None and not skip_levels:
            self._graph_tables_done(defer_reset)
        elif graph_header is not None:
            # the level launch is still to come and reads the graph's own tables: the descriptor goes back to the shared header
            # behind it (levels_decode); the accumulators' reset is handed to the caller now -- it rides in that same launch
            if self._resets:
                assert defer_reset is not None, "skip_levels is for the whole-step capture, which folds the resets into its last launch"
                defer_reset.append((self._dev[self._table_words:self._dense_at], self._acc_init))
        return True

    def fusable_levels(self):
        """The level launch and the decode of the one payload can be ONE launch (native.HSQBatch.levels_decode)."""
        c0 = self.codecs[0]
        return (self._batch.path == native.BATCH_PREFILTER and self.n_bit != 32 and not c0.packed6
                and self.level_dtype in (torch.uint8, torch.int16) and not getattr(self, "fma", False))

    def levels_decode(self, plain, tail):
        """The launch encode(..., skip_levels=True) left out + the decode of that payload (+ tail) -> the output views."""
        wire_user, mode, seed, r_flat, write_error = self._pending_levels
        self._pending_levels = None
        out, views = self._out_buffer(wire_user.device)
        try:
            self._batch.levels_decode(wire_user, mode, seed, r_flat, write_error, out, plain=plain, tail=tail)
        except BaseException:
            self._graph_tables_abort()
            raise
        self._batch.set_table(self._dev[:self._table_words])      # (the reset itself rode in the launch: tail.reset)
        return views


class BatchedQSGD(_BatchedBase):
    """All packed-form QSGD tensors in ONE gq_qsgd_compress_batched / gq_qsgd_decode_sum_batched launch.
    Tensors with WIDE buckets (TernGrad's `--c-dim 0`: the tensor is one bucket; any bucket above WIDE_MIN
    elements) form their own group on the chunked kernels (gq_qsgd_wide_*: bucket norms, codes, decode)."""

    takes_tail = True      # gq_qsgd_decode_sum_batched_tail (the library runs gq_mean_rows behind a decode path without the in-kernel form)
    WIDE_MIN = 4096

    @staticmethod
    def eligible(codec):
        return type(codec) is QSGDCodec and codec.bits != 0

    @staticmethod
    def group_key(codec):
        return (codec.bits, codec.c.bit, int(codec.d > BatchedQSGD.WIDE_MIN))

    def __init__(self, codecs, offsets, idxs, device, slots, user_bytes, dense=None):
        self.idxs = list(idxs)
        self.codecs = [codecs[i] for i in self.idxs]
        c0 = self.codecs[0]
        self.n_bit, self.bits, self.random = c0.c.bit, c0.bits, bool(c0.c.random)
        self.keyed = bool(self.random and c0.c._rng == "keyed")
        self.counter = bool(self.random and c0.c._rng == "device")
        self.wide = c0.d > self.WIDE_MIN
        assert all(cd.bits == self.bits and cd.c.bit == self.n_bit and (cd.d > self.WIDE_MIN) == self.wide
                   for cd in self.codecs)
        nseg = len(self.idxs)
        table = torch.zeros((nseg, 8), dtype=torch.int64)
        item_seg = []
        item, out_off, word = 0, 0, 0
        self.out_off = []
        for s, (i, cd) in enumerate(zip(self.idxs, self.codecs)):
            # items: buckets, or (wide) chunks of native.QSGD_WIDE_CHUNK elements of a bucket
            items = cd.Mb * (-(-cd.d // native.QSGD_WIDE_CHUNK)) if self.wide else cd.Mb
            table[s, 1], table[s, 2] = cd.d, item
            table[s, 3] = offsets[i] + cd.norm_off
            table[s, 4] = offsets[i] + cd.codes_off
            table[s, 5] = out_off
            table[s, 6] = word if self.wide else cd.Mb     # wide: the tensor's first word in norm_bits
            item_seg += [s] * items
            item += items
            word += (cd.Mb + 31) & ~31 if self.wide else 0     # a tensor's norm words start on their own 128-byte line
            self.out_off.append(out_off)
            out_off += (cd.numel + 3) & ~3          # tensors start 16-byte aligned in `out` (dwordx4 stores)
        self.nbuckets, self.out_floats = item, out_off
        self.bucket_seg = torch.tensor(item_seg, dtype=torch.int32, device=device)
        self._item_seg, self._nitems = self.bucket_seg, item
        # wide: max |v| per bucket is folded into words that the per-step header resets to zero
        extra = torch.zeros((word + 1) // 2, dtype=torch.int64) if self.wide else None
        self._setup(table, extra, device, slots, user_bytes, dense)
        self._batch = native.QSGDBatch(self._dev[:self._table_words], self.bucket_seg, self.nseg, self.nbuckets, self.n_bit,
                                       self.bits, self.wide,
                                       self._dev[self._table_words:self._dense_at].view(torch.int32) if self.wide else None)

    def graphable(self):
        """The compress launch takes a fresh seed per record when it rounds stochastically: only the deterministic form
        can be a HIP graph node (see BatchedHSQ.graphable)."""
        return not self.random or self.keyed or (self.counter and self.rng_pairs is not None)

    def encode(self, tensors, wire_user, slot, salt, errs=None, ef_scale=None, draws=None, graph_header=None, dense=None, defer_reset=None,
               rng_slot=None):
        """With `errs`: error feedback in the same launch (t += ef_scale*err, err = t - decoded, both in place).
        graph_header, dense, rng_slot: see BatchedHSQ.encode."""
        rng_slot = slot if rng_slot is None else rng_slot
        if graph_header is not None:
            self._graph_tables(graph_header, dense)
        elif not self._upload(tensors, slot, 8, errs, dense):
            return False
        else:
            self._batch.set_dense(self.dense_table_dev() if dense is not None else None, self.ndense)
            self._acc_clean = False
        if self.keyed:      # gq_rng = "keyed": every bucket's draws keyed by its norm, the seed never changes
            mode, seed = native.RANDOM_DEVICE_KEYED, (salt * 0x2545F4914F6CDD1D + 0x5851F42D4C957F2D) & (2 ** 63 - 1)
        elif self.counter and self._counter_seed(rng_slot) is not None:      # gq_rng = "device": keyed by the slot's device step word
            mode, seed = native.RANDOM_DEVICE_COUNTER, self._counter_seed(rng_slot)
        else:
            mode = native.RANDOM_DEVICE if self.random else native.RANDOM_OFF
            seed = (_next_seed() ^ salt) if self.random else 0
        try:
            self._batch.compress(wire_user, mode, seed, ef_scale if errs is not None else None)
        except BaseException:
            if graph_header is not None:
                self._graph_tables_abort()
            raise
        if graph_header is not None:
            self._graph_tables_done(defer_reset)
        return True


def default_codec_factory(compressor, numel, shape, packed6=False):
    if isinstance(compressor, IdenticalCompressor):
        return DenseCodec(compressor, numel, shape)
    if isinstance(compressor, NearestNeighborCompressor):
        return HSQCodec(compressor, numel, shape, packed6)
    if isinstance(compressor, QSGDCompressor):
        return QSGDCodec(compressor, numel, shape)
    return GenericCodec(compressor, numel, shape)


# --------------------------------------------------------------------------------------
# Parameter-server quantizer
# --------------------------------------------------------------------------------------
def _ef_scale(args, epoch):
    # ps_quantizer.py:28-31
    if args.scale == 'exp':
        return 2 / (math.exp(-epoch) + 1) - 1
    return float(args.scale)


def _dist_world(process_group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(process_group), dist.get_rank(process_group)
    return 1, 0


class PSQuantizer(object):
    def __init__(self, Compressor, parameters, args, process_group=None, codec_factory=None):
        self.parameters = list(parameters)
        self.num_layers = len(self.parameters)
        self.args = args
        self.error_feedback = args.ef
        self.two_phase = args.two_phase
        self.process_group = process_group
        factory = codec_factory or default_codec_factory
        self.aggregate_fma = aggregate_fma(args)     # opt-in fused accumulation of the decode-mean (R >= 2 only)
        self.wire_levels = wire_levels_mode(args, _dist_world(process_group)[0])      # "bytes" | "packed6" (6-bit levels where the configuration allows)
        if self.wire_levels == "packed6":
            base_factory = factory
            factory = lambda comp, n, shape: base_factory(comp, n, shape, packed6=True)
        self.compressors = []
        self.codecs = []
        for param in self.parameters:
            param_size = param.flatten().shape[0]
            comp = Compressor(param_size, param.shape, args) if param_size > 1000 else IdenticalCompressor()
            self.compressors.append(comp)
            self.codecs.append(factory(comp, param_size, param.shape))
            if self.aggregate_fma and isinstance(self.codecs[-1], HSQCodec):
                self.codecs[-1].fma = True
            if self.error_feedback:
                param.error = [torch.zeros_like(param) for _ in range(args.num_users)]
            if self.error_feedback and self.two_phase:
                param.server_error = torch.zeros_like(param)
        # wire layout of one user: 16-byte aligned sections for the compressed tensors first, then ONE
        # packed region with the raw f32 of all identity-compressed (<= 1000 element) tensors
        self.offsets = [0] * self.num_layers
        off = 0
        self.dense_idx = [i for i, c in enumerate(self.codecs) if type(c) is DenseCodec]
        for i, c in enumerate(self.codecs):
            if i not in self.dense_idx:
                self.offsets[i] = off
                off = _up(off + c.nbytes)
        self.dense_off = off
        for i in self.dense_idx:
            self.offsets[i] = off
            off += self.codecs[i].nbytes
        self.dense_bytes = off - self.dense_off
        self._dense_mean, self._dense_views, self._dense_turn = [None, None], [None, None], 0
        self._dense_in = {}                 # (wire pointer, slot) -> views of the slot's dense region
        self.user_bytes = _up(off)          # one user's payload (all tensors)
        # tensors served by multi-tensor kernels: (class, parameter indices), built at the first record()
        self._groups = []
        for cls in (BatchedHSQ, BatchedQSGD):
            keyed = {}
            for i, c in enumerate(self.codecs):
                if cls.eligible(c):
                    keyed.setdefault(cls.group_key(c), []).append(i)
            for key in sorted(keyed):
                idx = keyed[key]
                if len(idx) >= 2 and not getattr(args, "gq_no_batch", False):
                    self._groups.append([cls, idx, None])
        self.batch_idx = [i for g in self._groups for i in g[1]]
        self._pick_dense = operator.itemgetter(*self.dense_idx) if len(self.dense_idx) >= 2 else None
        self._pick_group = {}
        # gq_graph (args.gq_graph or $GQ_GRAPH=1): the device work of a record() for a set of gradient addresses seen before
        # is replayed as ONE HIP graph launch (the step is a launch-bound loop: ten launches and copies against ~85 us of
        # kernels).  Needs launches whose arguments do not change between records: deterministic rounding or gq_rng="keyed".
        g = getattr(args, "gq_graph", None)
        self.use_graphs = bool(int(os.environ.get("GQ_GRAPH", "1"))) if g is None else bool(g)
        # gq_rng = "device" (the default): the multi-tensor launches draw from streams keyed by { seed, step } pairs in device
        # memory, one pair per (tensor group, user slot) (GQ_RANDOM_DEVICE_COUNTER); every aggregate adds one to the step
        # words.  The launches' arguments never change -- they replay from a HIP graph -- and the draws are fresh every
        # step whatever the gradients are (the reference draws per call: probabilistic_scalar_compressor.py:22-26).
        self._rng_state = None
        self._ticket = None          # gq_hsq_levels_decode_batched's last-workgroup counter (one device word, zero between launches)
        self._rec_graphs = {}        # (slot, user, scale, gradient addresses) -> [sightings, graph or None, keep-alive]
        self._apply_graphs = {}      # (users recorded, wire, output-buffer turns) -> [sightings, graph or None, decoded list]
        # One rank, one user per step (args.num_users == 1, no process group): record() is always followed by the apply() of
        # exactly that payload, so the two replay as ONE graph from record() -- compress and decode-mean launches back to
        # back, one graph launch less per step (5 us of ~70, tools/graph_pieces.py) -- and apply() only rebinds the gradients.
        # $GQ_FUSE_STEP=0 keeps the two graphs.
        self._step_graphs = {}       # (record key, apply key) -> [sightings, graph or None, decoded list]
        self._fused = None           # the decoded list of a step whose record() has already replayed its apply()
        self._fuse_steps = type(self) is PSQuantizer and os.environ.get("GQ_FUSE_STEP", "1") != "0"
        # gq_rng = "reference": the reference draws r = torch.rand(M) per compressed tensor, in parameter order, from
        # the CPU generator (probabilistic_scalar_compressor.py:23).  torch.rand is one sequential stream, so ONE
        # torch.rand(sum of M) per record (and one per two-phase apply) gives every tensor the same numbers; the
        # multi-tensor kernels and the per-tensor path both take their slices from it.
        self._draw_off, n = {}, 0
        for i, c in enumerate(self.codecs):
            if isinstance(c, HSQCodec) and c.uses_reference_draws():
                self._draw_off[i] = n
                n += c.M
        self._draw_total = n
        self._draw_host = None
        self._assembled = {}
        self._plan = None
        self.capacity = max(1, int(args.num_users))
        self.recorded = 0                   # record() calls since the last apply()
        self._wire = None
        self._ex = None                     # exchange.WireExchange when torch.distributed is initialised
        self.exchange_mode = exchange.configured_mode()    # $GQ_EXCHANGE: allgather | direct | split | auto
        # split exchange: bytes [0, cut) travel (and are decoded) first; the cut lies on a tensor boundary near the
        # middle of the compressed part of the wire
        bounds = [self.offsets[i] for i in range(self.num_layers) if i not in self.dense_idx]
        later = [o for o in bounds if o >= self.dense_off // 2]
        self.cut = min(later) if later else 0
        # pipelined exchange ($GQ_EXCHANGE=pipelined, opt-in): $GQ_PIPELINE_CHUNKS byte ranges (default 4) of about equal size
        # with their boundaries on tensor boundaries; the identity-compressed tensors ride in the last one
        nchunks = max(2, int(os.environ.get("GQ_PIPELINE_CHUNKS", "4")))
        self.cuts = []
        for k in range(1, nchunks):
            inner = [o for o in bounds if o > 0]
            if inner:
                c = min(inner, key=lambda o: (abs(o - self.dense_off * k // nchunks), o))      # the tensor boundary nearest to k / nchunks
                if c not in self.cuts:
                    self.cuts.append(c)
        self.cuts.sort()

    # ---- buffers -------------------------------------------------------------------------
    def _ensure_wire(self, device, slots):
        """This rank's [capacity, user_bytes] wire.  Under torch.distributed it is this rank's block of rows of
        the exchange buffer (gq_amd.exchange): the kernels write where the collective reads."""
        world, rank = _dist_world(self.process_group)
        if (self._wire is None or self._wire.device != device or self._wire.shape[0] < slots
                or (world > 1) != (self._ex is not None)):
            cap = max(self.capacity, slots)
            old = self._wire
            if world > 1:
                self._ex = exchange.WireExchange(world, rank, cap, self.user_bytes, device, self.process_group)
                new = self._ex.local
            else:
                self._ex = None
                new = torch.zeros((cap, self.user_bytes), dtype=torch.uint8, device=device)
            if old is not None and old.device == device:
                new[:min(cap, old.shape[0])].copy_(old[:cap])
            self._wire = new
            self.capacity = cap
        return self._wire

    def wire_bytes_per_user(self):
        return self.user_bytes

    RNG_SLOTS = 17      # { seed, step } pairs per group: 16 user slots + the two-phase re-compress (the last one)
    TWO_PHASE_RNG_SLOT = RNG_SLOTS - 1

    def _rng_pairs_for(self, device, group_index):
        """This group's rows of the quantizer's { seed, step } array (made on first use; seeds from torch's seed, the
        rank, the group and the slot; steps start at 0).  The two-phase slot's seed leaves the rank out: the second
        phase runs replicated on every rank and must round identically everywhere (ps_quantizer.py:52-61 runs it once,
        on the server); the step words advance in lockstep on all ranks."""
        if self._rng_state is None or self._rng_state.device != device:
            world, rank = _dist_world(self.process_group)
            n = max(1, len(self._groups)) * self.RNG_SLOTS
            host = torch.zeros((n, 2), dtype=torch.int64)
            base = _next_seed()
            for i in range(n):
                r = 0 if i % self.RNG_SLOTS == self.TWO_PHASE_RNG_SLOT else rank
                host[i, 0] = ((base ^ ((r * 1000003 + i + 1) * 0x9E3779B97F4A7C15)) & (2 ** 63 - 1))
            self._rng_state = host.to(device)
        return self._rng_state[group_index * self.RNG_SLOTS:(group_index + 1) * self.RNG_SLOTS]

    def _draws(self, device):
        """One torch.rand for all reference-parity tensors of this record / two-phase apply -> (device tensor, offsets)."""
        if not self._draw_total:
            return None
        if device.type != "cuda":
            return torch.rand(self._draw_total), self._draw_off
        if self._draw_host is None:
            self._draw_host = [torch.empty(self._draw_total).pin_memory() for _ in range(2)]
            self._draw_turn = 0
            self._draw_events = [None, None]
        k = self._draw_turn
        self._draw_turn ^= 1
        if self._draw_events[k] is not None:
            self._draw_events[k].synchronize()      # the previous copy out of this pinned buffer
        torch.rand(self._draw_total, out=self._draw_host[k])     # straight into pinned memory (a fresh 6 MB tensor per
        dev_r = self._draw_host[k].to(device, non_blocking=True)  # record cost 20 ms of page faults on the GPU box)
        ev = torch.cuda.Event()
        ev.record()
        self._draw_events[k] = ev
        return dev_r, self._draw_off

    # ---- reference protocol -------------------------------------------------------------
    def record(self, user, epoch):
        scale = _ef_scale(self.args, epoch)
        scan = _HOST.scan_grads(self.parameters) if _HOST is not None else None     # (grads, addresses as bytes, all plain f32)
        all_grads = scan[0] if scan is not None else [p.grad for p in self.parameters]     # (p.grad.data builds an alias tensor per access: ~1 us each)
        dev = all_grads[0].device
        slot = self.recorded
        wire = self._ensure_wire(dev, slot + 1)[slot]
        world, rank = _dist_world(self.process_group)
        salt = ((rank * 1000003 + user) * 0x9E3779B1) & (2 ** 62 - 1)
        skip = set()
        draws = self._draws(dev)
        # gq_graph: a record whose gradient addresses were seen before replays its device work as ONE graph launch
        graph_key = None
        if (self.use_graphs and dev.type == "cuda" and not self._draw_total and slot < self.TWO_PHASE_RNG_SLOT
                and all(g[2] is not None and g[2].ready and g[2].graphable() for g in self._groups)
                and not torch.cuda.is_current_stream_capturing()):      # (inside a caller's own capture the launches are simply recorded)
            graph_key = (slot, user, self._wire.data_ptr(), scan[1] if scan is not None else tuple(map(_DATA_PTR, all_grads)))
            if self.error_feedback:     # the residual buffers' addresses are in the header too (a per-tensor step replaces them)
                graph_key += (scale, tuple(p.error[user].data_ptr() for p in self.parameters))
            ent = self._rec_graphs.get(graph_key)
            plain_f32 = ent is not None and ent[1] is not None and (
                scan[2] if scan is not None else (all(map(_IS_CONTIGUOUS, all_grads)) and set(map(_DTYPE_OF, all_grads)) == _F32_ONLY))
            step_key = None
            if (plain_f32 and self._fuse_steps and world == 1 and slot == 0 and self.capacity == 1 and not self.two_phase
                    and self._plan is not None and not self._plan[2]):      # (the plan: everything decodes through multi-tensor launches)
                step_key = (graph_key, self._apply_key(self._wire[:1]))
                fent = self._step_graphs.get(step_key)
                if fent is not None and fent[1] is not None:      # compress + decode-mean of this step in one launch
                    for g in self._groups:
                        g[2].ensure_clean()
                    fent[1].replay()
                    for g in self._groups:
                        g[2]._last_ptrs = None
                        g[2]._out_turn ^= 1
                    if len(self.dense_idx) >= 2:
                        self._dense_turn ^= 1
                    self._fused = fent[2]
                    self.recorded += 1
                    return
            if plain_f32:
                for g in self._groups:
                    g[2].ensure_clean()
                ent[1].replay()
                for g in self._groups:
                    g[2]._last_ptrs = None      # the device header now holds this graph's table: the next eager call re-sends its own
                self.recorded += 1
                if step_key is not None:
                    fent = self._graph_entry(self._step_graphs, step_key)
                    if fent is not None and fent[0] >= 2 and fent[1] is None:
                        self._capture_step(fent, ent[2], all_grads, wire, slot, user, salt, scale, dev)
                return
        skip = self._record_launches(all_grads, wire, slot, user, salt, scale, draws, dev)
        if len(skip) == self.num_layers:     # the usual case: everything went through the multi-tensor launches
            self.recorded += 1
            if graph_key is not None:
                ent = self._graph_entry(self._rec_graphs, graph_key)
                if ent is not None and ent[0] >= 2 and ent[1] is None:      # the second sighting: worth a capture
                    self._capture_record(ent, all_grads, wire, slot, user, salt, scale, dev)
            return
        for i, param in enumerate(self.parameters):
            if i in skip:
                continue
            codec, off = self.codecs[i], self.offsets[i]
            grad = param.grad.data
            if self.error_feedback:
                # ps_quantizer.py:35-39:  grad += scale*error ; error = grad - decoded
                if grad.device.type == "cuda" and grad.is_contiguous() and grad.dtype == torch.float32:
                    native.axpy_inplace(grad, param.error[user].contiguous(), scale)
                else:
                    grad.add_(scale * param.error[user])
                if hasattr(codec, "encode_decode_into"):
                    decoded = torch.empty(grad.numel(), dtype=torch.float32, device=grad.device)
                    codec.encode_decode_into(grad, wire, off, salt, decoded, **self._slice(draws, i))
                    decoded = decoded.view(param.shape)
                elif hasattr(codec, "decode_wire"):
                    codec.encode_into(grad, wire, off, salt, **self._slice(draws, i))
                    decoded = torch.empty(grad.numel(), dtype=torch.float32, device=grad.device)
                    codec.decode_wire(wire, off, decoded)
                    decoded = decoded.view(param.shape)
                else:
                    decoded = codec.roundtrip(grad, salt)
                if grad.device.type == "cuda" and grad.is_contiguous():
                    err = torch.empty_like(grad)
                    native.sub(grad, decoded.contiguous(), err)
                    param.error[user].data = err
                else:
                    param.error[user].data = grad - decoded
            else:
                codec.encode_into(grad, wire, off, salt, **self._slice(draws, i))
        self.recorded += 1

    def _make_group(self, grp, dev):
        """The multi-tensor launch object of one group, built the same way whoever needs it first (a record, or a ring rank
        that decodes before it has encoded anything): dense copy table, draws' { seed, step } pairs, aggregate form."""
        cls, idxs = grp[0], grp[1]
        # the first group's compress launch also copies the identity-compressed tensors into the wire
        dense = ([(self.offsets[i], self.codecs[i].numel) for i in self.dense_idx]
                 if (grp is self._groups[0] and len(self.dense_idx) >= 2) else None)
        obj = grp[2] = cls(self.codecs, self.offsets, idxs, dev, self.capacity, self.user_bytes, dense=dense)
        obj.fma = bool(self.aggregate_fma and cls is BatchedHSQ)
        if getattr(obj, "counter", False) and len(self._groups) * self.RNG_SLOTS <= 256:
            obj.rng_pairs = self._rng_pairs_for(dev, self._groups.index(grp))
        return obj

    def _record_launches(self, all_grads, wire, slot, user, salt, scale, draws, dev, headers=None, defer_resets=None, fuse_levels=False):
        """The multi-tensor launches of a record (+ the dense tensors' copy into the wire) -> the set of parameters served.
        headers (stream capture): one device-resident header per group, see BatchedHSQ.encode.
        fuse_levels (whole-step capture, _can_fuse_levels): the group's level launch is left to the aggregate's decode."""
        skip = set()
        skip_groups = []
        for grp in (self._groups if dev.type == "cuda" else []):
            cls, idxs, obj = grp
            if obj is None:
                obj = self._make_group(grp, dev)
            pick = self._pick_group.get(id(grp))      # operator.itemgetter over the group's indices, built once
            if pick is None:
                pick = self._pick_group[id(grp)] = operator.itemgetter(*idxs)     # (a group has at least two tensors)
            grads = list(pick(all_grads))
            # error feedback (ps_quantizer.py:35,39) rides in the same launches: grad += scale*error
            # before the encode, error = grad - decoded after it, both in place
            errs = [self.parameters[i].error[user] for i in idxs] if self.error_feedback else None
            hdr = headers[len(skip_groups)] if headers is not None else None
            skip_groups.append(obj)
            dense = list(self._pick_dense(all_grads)) if obj.ndense else None
            kw = {"skip_levels": True} if fuse_levels else {}
            if obj.encode(grads, wire, slot, salt, errs, scale, draws=draws, graph_header=hdr, dense=dense, defer_reset=defer_resets, **kw):
                skip.update(idxs)
                if dense is not None:
                    skip.update(self.dense_idx)      # (copied by that launch)
        if len(self.dense_idx) >= 2 and self.dense_idx[0] not in skip:
            # all small tensors with one concatenation straight into the packed wire region.  Under
            # error feedback their residual is identically zero (decoded == grad), so nothing else to do.
            key = (self._wire.data_ptr(), slot)
            views = self._dense_in.get(key)
            if views is None:       # parameter-shaped views of this slot's dense region, built once
                region = wire[self.dense_off:self.dense_off + self.dense_bytes].view(torch.float32)
                views, o = [], 0
                for i in self.dense_idx:
                    n = self.codecs[i].numel
                    views.append(region[o:o + n].view(self.codecs[i].shape))
                    o += n
                self._dense_in = {k: v for k, v in self._dense_in.items() if k[0] == key[0]}   # drop a replaced wire's
                self._dense_in[key] = views
            with torch.no_grad():
                torch._foreach_copy_(views, list(self._pick_dense(all_grads)))
            skip.update(self.dense_idx)
        return skip

    @staticmethod
    def _graph_entry(cache, key, max_captured=48, max_counting=64):
        """[sightings, graph or None, keep-alive] of `key`, its sighting counted.  A few address sets recur (the allocator
        hands the same blocks out again); captured graphs are never evicted -- once max_captured of them exist, new sets keep
        their eager launches (None) instead of displacing one another capture by capture -- and among the entries that
        only count sightings the oldest goes first."""
        ent = cache.get(key)
        if ent is None:
            counting = [k for k, e in cache.items() if e[1] is None]
            if len(cache) - len(counting) >= max_captured:
                return None
            if len(counting) >= max_counting:
                cache.pop(counting[0])
            ent = cache[key] = [0, None, None]
        ent[0] += 1
        return ent

    def _capture_record(self, ent, all_grads, wire, slot, user, salt, scale, dev):
        """Stream-capture the launches the record just made eagerly, with copies of the headers it has just sent
        (the shared pinned buffers are rewritten by later records, a graph's memcpy node reads its source at every replay)."""
        try:
            # (device-resident: the graph's copy node is device-to-device.  A host-to-device node -- pinned memory over PCIe --
            # cost 15 us of every replayed step, tools/graph_pieces.py; the 5 KB header per captured graph is nothing)
            headers = [g[2]._host[g[2]._last_slot].to(dev) for g in self._groups]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):     # (other threads -- RCCL's watchdog -- may call into HIP meanwhile)
                self._record_launches(all_grads, wire, slot, user, salt, scale, None, dev, headers=headers)
        except Exception as e:      # a capture that fails leaves the eager path as it was (this record has already run eagerly)
            self.use_graphs = False
            import warnings
            warnings.warn("gq_graph: capturing a record failed (%s); continuing with eager launches" % (e,))
            return
        ent[1], ent[2] = graph, headers

    def _capture_step(self, fent, headers, all_grads, wire, slot, user, salt, scale, dev):
        """One graph for a whole step: the record's launches (headers: the device copies its own graph keeps) and the
        decode-mean launches the following apply() would make, captured from record() after this record has run.  Nothing
        executes here; the output-buffer turns the capture advances are put back for the apply() that is still to come."""
        after = ([g[2]._out_turn for g in self._groups], self._dense_turn)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                resets = []      # the groups' accumulator resets ride in the step's last launch
                fuse = self._can_fuse_levels()      # one rank, one user: level launch + decode of that payload as ONE launch
                self._record_launches(all_grads, wire, slot, user, salt, scale, None, dev, headers=headers, defer_resets=resets,
                                      fuse_levels=fuse)
                decoded = self._decode_all(self._wire[:1], False, (), resets=resets, fused_levels=fuse)
            fent[1], fent[2] = graph, decoded
        except Exception as e:      # the two-graph replay keeps working
            self._fuse_steps = False
            import warnings
            warnings.warn("gq_graph: capturing a whole step failed (%s); record and apply keep their own graphs" % (e,))
        finally:
            for g, t in zip(self._groups, after[0]):
                g[2]._out_turn = t
            self._dense_turn = after[1]

    def _can_fuse_levels(self):
        """A whole step of one rank and one user whose tensors all go through ONE HSQ group (+ the dense tensors riding in
        its launches): encode, then gq_hsq_levels_decode_batched.  $GQ_FUSE_LEVELS=0 keeps the three launches."""
        if os.environ.get("GQ_FUSE_LEVELS", "1") == "0" or len(self._groups) != 1:
            return False
        if self.error_feedback and os.environ.get("GQ_FUSE_LEVELS") != "ef":
            # measured (profiles/r05_experiments.txt, 2): with error feedback the one launch moves three streams (updated gradient in,
            # residual and decoded tensor out) and is SLOWER than the level launch + the decode (0.1233 against 0.1158 ms per step)
            return False
        obj = self._groups[0][2]
        return (isinstance(obj, BatchedHSQ) and obj.fusable_levels() and (not self.dense_idx or (len(self.dense_idx) >= 2 and obj.ndense))
                and os.environ.get("GQ_STEP_TAIL", "1") != "0")

    def _apply_key(self, gathered):
        """What an apply()'s captured launches depend on: payload count, the wire, and which output buffers are next."""
        return (gathered.shape[0], gathered.data_ptr(), tuple(g[2]._out_turn for g in self._groups), self._dense_turn,
                tuple(0 if o is None else o.data_ptr() for g in self._groups for o in g[2]._outs),
                tuple(0 if m is None else m.data_ptr() for m in self._dense_mean))

    def _slice(self, draws, i):
        """This parameter's share of the record's draws as a keyword for the codec (nothing for the other codecs)."""
        if draws is None or i not in self._draw_off:
            return {}
        o = self._draw_off[i]
        return {"r": draws[0][o:o + self.codecs[i].M]}

    def _decode_all(self, gathered, two_phase, pending=(), plain=False, resets=None, fused_levels=False):
        """Mean of the R = gathered.shape[0] user payloads for every parameter (ps_quantizer.py:47-61),
        as a list of tensors in parameter order.  `pending`: the transfers that fill `gathered`
        (exchange.WireExchange.start) -- one, or one per byte range for a split / pipelined exchange, in which case the
        tensors of a range are decoded while the ranges behind it are still in flight."""
        R = gathered.shape[0]
        done = {}
        pending = list(pending)
        chunked = len(pending) >= 2      # split / pipelined: byte ranges of the wire, each decoded as soon as it has arrived
        on_gpu = gathered.device.type == "cuda"
        # a group that did not encode in multi-tensor form this run (unaligned tensors, too many of them) is not
        # `ready`: its tensors take the per-tensor decode below
        key = (on_gpu,) + tuple(g[2] is not None and g[2].ready for g in self._groups)
        if self._plan is None or self._plan[0] != key:      # who decodes what: rebuilt only when a group's state changes
            groups = [g for g in (self._groups if on_gpu else []) if g[2] is not None and g[2].ready]
            batched = set(i for g in groups for i in g[1])
            dense = set(self.dense_idx) if len(self.dense_idx) >= 2 else set()
            single = [i for i in range(self.num_layers) if i not in batched and i not in dense]
            self._plan = (key, groups, single, {})
        _, groups, single, seg_ranges = self._plan
        single = list(single)
        group_views = {}

        # The aggregate's small per-step work -- the mean of the identity-compressed tensors' rows, one step of the draws'
        # { seed, step } words, the accumulators' reset (whole-step capture) -- rides in the LAST multi-tensor decode launch
        # that can take it (native.StepTail: gq_hsq_decode_sum_batched_tail): one kernel and one boundary less per step.
        # Not with a chunked exchange (several decode launches per group), not with two-phase (its re-compress draws from
        # the step words and folds into the accumulators AFTER this decode).
        step_rng = self._rng_state is not None and on_gpu     # one step of the device draws per aggregate (GQ_RANDOM_DEVICE_COUNTER)
        dense_job = None
        if len(self.dense_idx) >= 2:
            # identity tensors: two-phase / error feedback leave them unchanged (roundtrip == clone)
            rows = gathered[:, self.dense_off:self.dense_off + self.dense_bytes].view(torch.float32)
            k = self._dense_turn
            self._dense_turn ^= 1
            if self._dense_mean[k] is None or self._dense_mean[k].device != rows.device:
                mean = torch.empty(rows.shape[1], dtype=torch.float32, device=rows.device)
                views, o = [], 0
                for i in self.dense_idx:
                    n = self.codecs[i].numel
                    views.append(mean[o:o + n].view(self.codecs[i].shape))
                    o += n
                self._dense_mean[k], self._dense_views[k] = mean, views
            dense_job = (rows, k)
        tail, tail_group = None, -1
        if on_gpu and not chunked and not two_phase and os.environ.get("GQ_STEP_TAIL", "1") != "0":
            takers = [gi for gi, g in enumerate(groups) if g[2].takes_tail]
            mean_in_tail = dense_job is not None and not (plain and R == 1)
            if takers and (mean_in_tail or step_rng or resets):
                tail_group = takers[-1]
                if self._ticket is None or self._ticket.device != gathered.device:
                    self._ticket = torch.zeros(native.TICKET_WORDS, dtype=torch.int32, device=gathered.device)
                tail = native.StepTail(rows=dense_job[0] if mean_in_tail else None,
                                       out=self._dense_mean[dense_job[1]] if mean_in_tail else None,
                                       rng_state=self._rng_state if step_rng else None, reset=resets.pop(0) if resets else None,
                                       ticket=self._ticket)
                step_rng = False
                if mean_in_tail:
                    dense_job = (None, dense_job[1])      # (done by the decode launch)

        def decode_range(lo, hi, first):
            """The tensors whose wire section starts in [lo, hi) (None: all of them)."""
            for gi, (cls, idxs, obj) in enumerate(groups):
                part = None
                if lo is not None:
                    part = seg_ranges.get((gi, lo, hi))
                    if part is None:      # a group's tensors are in wire order: the range is a run of them
                        part = seg_ranges[(gi, lo, hi)] = (sum(1 for i in idxs if self.offsets[i] < lo),
                                                          sum(1 for i in idxs if self.offsets[i] < hi))
                    part = part + (first,)
                if fused_levels and lo is None and getattr(obj, "_pending_levels", None) is not None:
                    group_views[gi] = obj.levels_decode(plain, tail if gi == tail_group else None)      # levels + decode (+ tail): one launch
                else:
                    group_views[gi] = obj.decode_mean(gathered, R, part, plain=plain, tail=tail if gi == tail_group else None)
            for i in single:
                if lo is None or lo <= self.offsets[i] < hi:
                    done[i] = self.codecs[i].decode_mean(gathered, self.offsets[i], R, plain=plain)

        if not chunked:
            if pending:
                pending.pop(0).wait()
            decode_range(None, None, True)
        else:
            for k, pnd in enumerate(pending):
                pnd.wait()
                decode_range(pnd.lo, self.user_bytes if pnd.hi is None else pnd.hi, k == 0)
        draws2 = self._draws(gathered.device) if two_phase else None     # the second phase compresses again: new draws
        sources = []        # the lists of output views this call's result is assembled from (persistent objects, see below)
        for gi, (cls, idxs, obj) in enumerate(groups):
            gs = group_views[gi]
            if two_phase:
                # ps_quantizer.py:52-61, replicated on every rank (salt 0, same call count); with error
                # feedback g += server_error and server_error = g - decoded happen inside the launches
                serr = [self.parameters[i].server_error for i in idxs] if self.error_feedback else None
                dec = obj.roundtrip(list(gs), self.capacity, 0, serr, 1.0, draws=draws2, rng_slot=self.TWO_PHASE_RNG_SLOT)
                if dec is None:     # not batchable this step: per-tensor second phase below
                    for i, g in zip(idxs, gs):
                        done[i] = g
                        single.append(i)
                    continue
                gs = dec
            sources.append((idxs, gs))
        if dense_job is not None:
            rows, k = dense_job
            if rows is None:
                pass                                     # the mean rode in a decode launch (tail)
            elif plain and R == 1:
                self._dense_mean[k].copy_(rows[0])      # the ring's hop: the payload as it is (a -0 stays -0)
            elif rows.device.type == "cuda":
                # stack().mean(0) with the CPU's arithmetic (true division); the same launch steps the draws' step words
                native.mean_rows(rows, self._dense_mean[k], rng_state=self._rng_state if step_rng else None,
                                 reset=resets.pop(0) if resets else None)
                step_rng = False
            else:
                torch.mean(rows, dim=0, out=self._dense_mean[k])   # stack().mean(0) of the reference, all at once
            sources.append((self.dense_idx, self._dense_views[k]))
        if step_rng:        # no identity-compressed tensors to average (or the ring's plain hop): a launch of its own
            native.rng_step(self._rng_state, reset=resets.pop(0) if resets else None)
        for dst, src in (resets or ()):      # (whole-step capture) what no launch of the aggregate took along
            _kernel_copy(dst, src)
        if not single and not done:
            # everything came out of multi-tensor launches: the result is a fixed interleaving of a few PERSISTENT view lists
            # (two output buffers per group used in turn, their per-tensor views built once), so the parameter-ordered
            # list is assembled once per combination and reused (161 dictionary stores + look-ups per step otherwise)
            key = tuple(id(v) for _, v in sources)
            hit = self._assembled.get(key)
            if hit is not None and all(a is b for a, (_, b) in zip(hit[0], sources)):
                return hit[1]
            for idxs, vs in sources:
                for i, v in zip(idxs, vs):
                    done[i] = v
            out = [done[i] for i in range(self.num_layers)]
            if len(self._assembled) > 8:
                self._assembled.clear()
            self._assembled[key] = ([v for _, v in sources], out)      # (holding the lists keeps their ids from being reused)
            return out
        for idxs, vs in sources:
            for i, v in zip(idxs, vs):
                done[i] = v
        if two_phase:
            for i in single:
                # ps_quantizer.py:52-61 -- identical on every rank (salt 0, same call count)
                param, codec, g = self.parameters[i], self.codecs[i], done[i]
                kw = self._slice(draws2, i)
                if self.error_feedback:
                    g = g + param.server_error
                    decoded = codec.roundtrip(g, 0, **kw)
                    param.server_error = g - decoded
                    g = decoded
                else:
                    g = codec.roundtrip(g, 0, **kw)
                done[i] = g
        return [done[i] for i in range(self.num_layers)]

    def apply(self, refresh_grads=False):
        """ps_quantizer.py:46-65.  refresh_grads: accepted for callers of earlier versions; `param.grad` is always evaluated here."""
        if self.recorded == 0:
            return
        world, rank = _dist_world(self.process_group)
        if world > 1:
            # ONE exchange per step: every rank's [users, bytes] block, rank-major (gq_amd.exchange)
            ex = self._ex
            if self.exchange_mode == "auto":
                def step(mode):
                    buf, pend = ex.start(mode, self.recorded, self.cut)
                    self._decode_all(buf, False, pend)
                self.exchange_mode = ex.autotune(
                    step, preflight=lambda mode: ex.start(mode, self.recorded, self.cut, dry_run=True))
            gathered, pending = ex.start(self.exchange_mode, self.recorded, self.cut, cuts=self.cuts)
        else:
            gathered, pending = self._wire[:self.recorded], ()
        decoded = None
        graph_key = None
        fused, self._fused = self._fused, None
        if fused is not None and self.recorded == 1 and world == 1:
            decoded = fused      # record() has replayed this step's decode-mean already (self._step_graphs)
        elif (self.use_graphs and len(pending) <= 1 and not self.two_phase and gathered.device.type == "cuda"
                and all(g[2] is not None and g[2].ready for g in self._groups) and not torch.cuda.is_current_stream_capturing()):
            # gq_graph: the decode-mean launches (+ the dense tensors' mean) of an apply that has been seen with these buffers
            # before replay as ONE graph launch; the two output buffers are used in turn, so two graphs alternate.  With
            # several ranks the exchange stays outside: its one transfer is waited for first (the split transport, whose
            # decode is interleaved with its second transfer, keeps its eager launches)
            for pnd in pending:
                pnd.wait()
            pending = ()
            graph_key = self._apply_key(gathered)
            ent = self._apply_graphs.get(graph_key)
            if ent is not None and ent[1] is not None:
                ent[1].replay()
                for g in self._groups:
                    g[2]._out_turn ^= 1
                if len(self.dense_idx) >= 2:
                    self._dense_turn ^= 1
                decoded = ent[2]
        if decoded is None:
            decoded = self._decode_all(gathered, self.two_phase, pending)
            if graph_key is not None and self._plan is not None and not self._plan[2]:     # (no per-tensor decodes in the plan)
                ent = self._graph_entry(self._apply_graphs, graph_key)
                if ent is not None and ent[0] >= 2 and ent[1] is None:
                    after = ([g[2]._out_turn for g in self._groups], self._dense_turn)
                    try:
                        for g, t in zip(self._groups, graph_key[2]):      # the capture re-issues the launches of THIS apply
                            g[2]._out_turn = t
                        self._dense_turn = graph_key[3]
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, capture_error_mode="thread_local"):     # (other threads -- RCCL's watchdog -- may call into HIP meanwhile)
                            again = self._decode_all(gathered, False, ())
                        if len(again) == len(decoded) and all(a is b for a, b in zip(again, decoded)):
                            ent[1], ent[2] = graph, decoded
                    except Exception as e:      # this apply has already run eagerly
                        self.use_graphs = False
                        import warnings
                        warnings.warn("gq_graph: capturing an apply failed (%s); continuing with eager launches" % (e,))
                    finally:
                        for g, t in zip(self._groups, after[0]):
                            g[2]._out_turn = t
                        self._dense_turn = after[1]
        # ps_quantizer.py:63 `param.grad.data = g`: `param.grad` is evaluated HERE, at apply time -- a caller that replaced a
        # parameter's .grad object after the last record() gets the mean in the object it holds now.  The C++ helper reads
        # p.grad() without building Python objects (the 161 attribute look-ups were the step's largest host cost, which is
        # why earlier rounds rebound the objects record() had seen); without the helper the look-ups are paid.
        if _HOST is not None and hasattr(_HOST, "set_grad_data") and type(decoded) is list and len(decoded) == len(self.parameters):
            _HOST.set_grad_data(self.parameters, decoded)
        else:
            for p, g in zip(self.parameters, decoded):
                p.grad.data = g
        self.recorded = 0

    aggregate = apply


# --------------------------------------------------------------------------------------
# Ring quantizer (the reference's other --mode; sequential by construction)
# --------------------------------------------------------------------------------------
class RingQuantizer(PSQuantizer):
    """quantizers/ring_quantizer.py:7-49: user k adds user k-1's decoded running sum to its own
    gradient and re-compresses; the result is the LAST user's decode (a sum, not a mean).

    Built on PSQuantizer's wire and multi-tensor kernels: one record() is [grad += running] + the
    parameter-server record (error feedback included, ring_quantizer.py:33-40 == ps_quantizer.py:34-39)
    + a decode of the wire just written.  Under torch.distributed the ring is real: the users are
    numbered rank-major, the compressed wire (not the decoded sum) travels rank -> rank+1 as ONE
    point-to-point message over xGMI when a rank's users are done, and the last rank broadcasts the
    final wire, which every rank decodes.  The chain is sequential by construction (each hop
    re-compresses the sum of everything before it), so it costs `world` encode latencies."""

    def __init__(self, Compressor, parameters, args, process_group=None, codec_factory=None):
        two_phase = args.two_phase
        args.two_phase = False           # ring_quantizer.py has no second phase and no server residual
        try:
            super().__init__(Compressor, parameters, args, process_group, codec_factory)
        finally:
            args.two_phase = two_phase
        self.two_phase = False
        self.running = None              # decoded running sum, one tensor per parameter
        self._inbox = None

    def record(self, user, epoch):
        world, rank = _dist_world(self.process_group)
        dev = self.parameters[0].grad.device
        if self.running is None and rank > 0:
            # first local user of a rank > 0: the running sum arrives compressed from the previous rank
            import torch.distributed as dist
            if self._inbox is None or self._inbox.device != dev:
                self._inbox = torch.empty((1, self.user_bytes), dtype=torch.uint8, device=dev)
            dist.recv(self._inbox.view(-1), src=self._peer(rank - 1), group=self.process_group)
            self._ready_for_wire(dev)
            self.running = self._decode_all(self._inbox, False, plain=True)
        if self.running is not None:     # ring_quantizer.py:31-32 (user != 0)
            torch._foreach_add_([p.grad.data for p in self.parameters], list(self.running))
        self.recorded = 0                # every user re-uses wire slot 0
        super().record(user, epoch)
        self.running = self._decode_all(self._wire[:1], False, plain=True)

    def _peer(self, group_rank):
        import torch.distributed as dist
        return group_rank if self.process_group is None else dist.get_global_rank(self.process_group, group_rank)

    def _ready_for_wire(self, dev):
        """A rank may have to decode a wire before it has encoded anything: the multi-tensor kernels'
        segment tables (the layout part; pointers are not needed for a decode) must exist."""
        for grp in (self._groups if dev.type == "cuda" else []):
            if grp[2] is None:
                self._make_group(grp, dev)
            if not grp[2].ready:
                grp[2].upload_layout()

    def apply(self):
        world, rank = _dist_world(self.process_group)
        if world > 1:
            import torch.distributed as dist
            if self.recorded == 0 and self.running is None:
                return
            dev = self._wire.device
            if rank < world - 1:
                dist.send(self._wire[0], dst=self._peer(rank + 1), group=self.process_group)
            final = self._wire[:1] if rank == world - 1 else self._inbox_for(dev)
            dist.broadcast(final.view(-1), src=self._peer(world - 1), group=self.process_group)
            if rank != world - 1:
                self.running = self._decode_all(final, False, plain=True)
        if self.running is not None:     # ring_quantizer.py:45-46
            for param, g in zip(self.parameters, self.running):
                param.grad.data = g
        self.running = None
        self.recorded = 0

    def _inbox_for(self, dev):
        if self._inbox is None or self._inbox.device != dev:
            self._inbox = torch.empty((1, self.user_bytes), dtype=torch.uint8, device=dev)
        return self._inbox

    aggregate = apply


def Quantizer(Compressor, parameters, args, **kw):
    """quantizers/base_quantizer.py:5-10."""
    if args.mode == 'ps':
        return PSQuantizer(Compressor, parameters, args, **kw)
    elif args.mode == 'ring':
        return RingQuantizer(Compressor, parameters, args, **kw)
    assert False, "mode {} not recognized".format(args.mode)


__all__ = ["Quantizer", "PSQuantizer", "RingQuantizer"]
