"""Codecs: how one parameter tensor -- or all tensors of a model that share a compressor -- is written to and read
from the wire (gq_amd.quantizers owns the wire, the exchange and the step; the codecs are the only objects that touch
device memory, through gq_amd.native).

    DenseCodec                 IdenticalCompressor tensors (<= 1000 elements, ps_quantizer.py:18-19): raw f32
    HSQCodec / QSGDCodec       one tensor per launch (the reference's per-tensor loop, nearest_neighbor_compressor.py:63-90,
                               qsgd_compressor.py:42-71)
    BatchedHSQ / BatchedQSGD   every tensor of a model in one launch per stage (descriptor tables, HIP-graph friendly)
    GenericCodec               any other compressor object: its own compress / decompress, tensors on the wire as they are

`codec_factory` arguments of the quantizers exist so that the host logic can be exercised without a GPU by the tests
(with the CPU oracle as the checker codec: tests/oracle_codec.py).
"""
import operator
import os

import torch

from . import exchange, native
from .compressors import (IdenticalCompressor, NearestNeighborCompressor, QSGDCompressor, _next_seed,
                          _require_device)


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

    def encode_decode_into(self, grad, wire_user, off, salt, out):
        """encode_into + the decode of what was written (error feedback, ps_quantizer.py:37: the residual is zero)."""
        self.encode_into(grad, wire_user, off, salt)
        out.copy_(grad.reshape(-1))

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

    def encode_decode_into(self, grad, wire_user, off, salt, out):
        """ONE decompress(compress(grad)) (ps_quantizer.py:37): it is both what travels and what the residual is taken against
        (a compressor that rounds stochastically must not be asked twice)."""
        self.encode_into(grad, wire_user, off, salt)
        out.copy_(wire_user[off:off + self.numel * 4].view(torch.float32))

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
    configurations whose top level is <= 63 with d = 16, K <= 256).  args.gq_wire_levels, else $GQ_WIRE_LEVELS, else
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
        """The packed form is used for d = 16, K <= 256 (what the multi-tensor level / decode kernels with packed levels are
        built for; the per-tensor entry points would take any K <= 256) when no level exceeds 63: n_bit <= 6 without
        stochastic rounding (probabilistic_scalar_compressor.py:18: levels up to 2^n_bit - 1), n_bit <= 5 with it (:25: up
        to 2^n_bit).  Every other configuration keeps one byte (or more) per level."""
        if not compressor.compressed_norm or compressor.dim != 16 or compressor.K > 256:
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
    pinned H2D copy; a ring of pinned buffers keeps a copy in flight from being overwritten."""

    UPLOAD_RING = 8

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
        # a ring of pinned copies of the header: an upload rewrites the OLDEST one, so the copy it has to wait for was queued
        # UPLOAD_RING uploads ago (round 6: indexed by the user slot, a one-user training loop rewrote the same buffer every step
        # and waited for the previous step's copy -- the host could never run ahead of the device)
        self._host = [host.clone().pin_memory() for _ in range(max(slots + 1, self.UPLOAD_RING))]
        self._up_turn = 0
        self._host_np = [h[:self._table_words].view(self.nseg, 8).numpy() for h in self._host]   # views of the pinned tables
        self._host_dense_np = [h[self._dense_at:].view(self.ndense, 3).numpy() for h in self._host] if self.ndense else None
        self._last_dptrs = None
        self._zeros = [0] * self.nseg
        self._resets = extra is not None    # the header also carries per-step reset values (min / max accumulators)
        self._acc_init = extra.view(-1).to(device) if extra is not None else None     # the accumulators' empty state, on the device
        self._acc_clean = False             # the device accumulators are in that state right now (see _graph_tables)
        self._last_ptrs = self._last_eptrs = None
        self._events = [None] * len(self._host)
        self._dev = torch.empty_like(host, device=device)
        self._tmp_wire = None
        self.ready = False      # the device header has been written at least once
        self._outs, self._out_views, self._out_turn = [None, None], [None, None], 0
        self._out_gen = 0               # output-buffer allocations so far (PSQuantizer._apply_key_for remembers the buffers' addresses per count)
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
            self._out_gen += 1
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
        slot = self._up_turn      # (the user slot plays no part: any pinned buffer whose last copy is done will do)
        self._up_turn = (slot + 1) % len(self._host)
        if self._events[slot] is not None:
            self._events[slot].synchronize()       # the previous copy out of this pinned buffer (UPLOAD_RING uploads ago)
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

    def upload(self, tensors, slot, errs=None, dense=None):
        """The header of these tensors to the device (pointers, accumulator resets, dense copy table), as the eager encode sends
        it -- in front of the replay of an address-free graph.  False: a tensor cannot be addressed that way."""
        if not self._upload(tensors, slot, self.align, errs, dense):
            return False
        self._acc_clean = False
        return True

    def _counter_seed(self, slot, reserved=False):
        """GQ_RANDOM_DEVICE_COUNTER: the address of this group's { seed, step } pair of user slot `slot`, or None when the
        quantizer gave the group no pairs (a codec used on its own) or not enough of them.  The LAST pair is reserved for the
        two-phase re-compress (its seed is the same on every rank): a record() of a user slot that high gets None -- a fresh
        per-call seed -- unless the caller asks for the reserved pair by name (reserved=True)."""
        if self.rng_pairs is None or not 0 <= slot < self.rng_pairs.shape[0] - (0 if reserved else 1):
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

    def roundtrip(self, tensors, slot, salt, errs=None, ef_scale=None, draws=None, rng_slot=None, graph_header=None, defer_reset=None):
        """decompress(compress(t)) for every batched tensor in a few launches; None if not batchable.
        With `errs`: t <- t + ef_scale*err in place first and err <- t - decoded afterwards.
        rng_slot: the { seed, step } pair the stochastic rounding draws from (default: the one of `slot`).
        graph_header, defer_reset (stream capture of a two-phase apply): see encode."""
        if self._tmp_wire is None:
            self._tmp_wire = torch.zeros((1, self.user_bytes), dtype=torch.uint8, device=self.device)
        kw = {"graph_header": graph_header, "defer_reset": defer_reset} if graph_header is not None else {}
        if not self.encode(tensors, self._tmp_wire[0], slot, salt, errs, ef_scale, draws=draws, rng_slot=rng_slot, **kw):
            return None
        return self.decode_mean(self._tmp_wire, 1, plain=True)     # decompress(compress(t)) (ps_quantizer.py:52-61): a -0 stays -0


class BatchedHSQ(_BatchedBase):
    """All NearestNeighborCompressor tensors that share a codebook are encoded by ONE encode + ONE levels
    launch and decoded by ONE decode-mean launch (per-tensor lb / ub, identical results).  The reference
    walks the parameter list in Python (ps_quantizer.py:33,47); ResNet-50 has 76 such tensors.
    The library decides which kernels serve the group's shape (include/gq_hsq.h, gq_hsq_batched_path): K <= 256 with
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
        # ONE launch descriptor for the group (gq_hsq_batch): the library picks the kernels -- prefilter (K <= 256,
        # d = 8 / 12 / 16 / 24 / 32), the same with the pages of a larger codebook resident, or exact scoring for every other shape
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
               rng_slot=None, skip_levels=False, table_current=False):
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
        elif table_current:
            # (capture of an ADDRESS-FREE graph, PSQuantizer._capture_generic: the launches read the shared device header, which
            # the caller refreshes by upload() in front of every replay -- pointers and accumulator resets -- as an eager step does)
            self._batch.set_table(self._dev[:self._table_words])
            self._batch.set_dense(self.dense_table_dev() if dense is not None else None, self.ndense)
        elif not self._upload(tensors, slot, self.align, errs, dense):
            return False
        else:
            self._batch.set_dense(self.dense_table_dev() if dense is not None else None, self.ndense)
            self._acc_clean = False
        ef = ef_scale if errs is not None else None
        counter_seed = self._counter_seed(slot) if rng_slot is None else self._counter_seed(rng_slot, reserved=True)
        try:
            self._batch.encode(wire_user, ef, self.profile_slot)
            self.profile_slot = -1
            if self.n_bit == 32:
                mode, seed, r_flat = native.RANDOM_OFF, 0, None
            elif self.reference_draws:
                mode, seed, r_flat = native.RANDOM_GIVEN, 0, self._given_draws(draws)
            elif self.keyed:
                mode, seed, r_flat = native.RANDOM_DEVICE_KEYED, (salt * 0x2545F4914F6CDD1D + 0x5851F42D4C957F2D) & (2 ** 63 - 1), None
            elif self.counter and counter_seed is not None:
                mode, seed, r_flat = native.RANDOM_DEVICE_COUNTER, counter_seed, None
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
    Tensors with WIDE buckets (TernGrad's `--c-dim 0`: the tensor is one bucket; any bucket of WIDE_MIN elements
    or more; see place_lone_buckets) form their own group on the chunked kernels (gq_qsgd_wide_*: bucket norms, codes, decode)."""

    takes_tail = True      # gq_qsgd_decode_sum_batched_tail (the library runs gq_mean_rows behind a decode path without the in-kernel form)
    WIDE_MIN = 1024        # buckets from here on go to the chunked kernels (place_lone_buckets)
    LONE_MIN = 256         # ... and a tensor that is ONE bucket may from here on
    NARROW_MAX = 4096      # the largest bucket the bucketed kernel is asked to walk

    @staticmethod
    def eligible(codec):
        return type(codec) is QSGDCodec and codec.bits != 0

    @staticmethod
    def is_wide(codec):
        return bool(getattr(codec, "_wide", codec.d >= BatchedQSGD.WIDE_MIN))

    @staticmethod
    def group_key(codec):
        return (codec.bits, codec.c.bit, int(BatchedQSGD.is_wide(codec)))

    @staticmethod
    def place_lone_buckets(codecs):
        """Which tensors go to the chunked (wide) kernels -- a chunk of 1,024 elements per wave, the bucket's max in a pass of
        its own -- and which to the bucketed kernel, which gives a bucket to 16 lanes (in registers up to 256 elements; above
        that the 16 lanes walk the bucket twice, an element pair per lane and trip):
          * buckets of at least WIDE_MIN = 1,024 elements are wide (whole chunks; ResNet-50 at c_dim 2048: 0.092 against 0.112 ms
            per step with the bucketed kernel's walk);
          * a tensor that IS one bucket of more than 256 elements joins them when there are wide tensors of its code format
            already, or two such tensors: for the bucketed kernel it is a latency chain, not throughput (TernGrad's
            1,024 ... 4,096-element tensors, ~60 buckets of the ResNet-50 list, kept one launch busy for 100 us: half of the step);
          * but a single wide tensor next to bucketed ones stays with them (up to 4,096 elements): a group of one is not
            batched at all (c_dim 512: the one 1,728-element tensor cost 0.12 ms per step as a group of its own)."""
        lone, wide, narrow = {}, {}, {}
        for cd in codecs:
            if not BatchedQSGD.eligible(cd):
                continue
            cd._wide = cd.d >= BatchedQSGD.WIDE_MIN
            fmt = (cd.bits, cd.c.bit)
            if cd._wide:
                wide.setdefault(fmt, []).append(cd)
            elif cd.Mb == 1 and cd.d > BatchedQSGD.LONE_MIN:
                lone.setdefault(fmt, []).append(cd)
            else:
                narrow.setdefault(fmt, []).append(cd)
        for fmt, cds in lone.items():
            if fmt in wide or len(cds) >= 2:
                for cd in cds:
                    cd._wide = True
                wide.setdefault(fmt, []).extend(cds)
            else:
                narrow.setdefault(fmt, []).extend(cds)
        for fmt, cds in wide.items():
            if len(cds) == 1 and narrow.get(fmt) and cds[0].d <= BatchedQSGD.NARROW_MAX:
                cds[0]._wide = False

    def __init__(self, codecs, offsets, idxs, device, slots, user_bytes, dense=None):
        self.idxs = list(idxs)
        self.codecs = [codecs[i] for i in self.idxs]
        c0 = self.codecs[0]
        self.n_bit, self.bits, self.random = c0.c.bit, c0.bits, bool(c0.c.random)
        self.keyed = bool(self.random and c0.c._rng == "keyed")
        self.counter = bool(self.random and c0.c._rng == "device")
        self.wide = self.is_wide(c0)
        assert all(cd.bits == self.bits and cd.c.bit == self.n_bit and self.is_wide(cd) == self.wide for cd in self.codecs)
        self.align = 8
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
        # the bucket width most elements have: the bucketed kernels give a bucket d / 8 lanes (4 ... 16)
        by_width = {}
        for cd in self.codecs:
            by_width[cd.d] = by_width.get(cd.d, 0) + cd.numel
        hint = 0 if self.wide else max(by_width, key=by_width.get)
        self._batch = native.QSGDBatch(self._dev[:self._table_words], self.bucket_seg, self.nseg, self.nbuckets, self.n_bit,
                                       self.bits, self.wide,
                                       self._dev[self._table_words:self._dense_at].view(torch.int32) if self.wide else None,
                                       bucket_hint=hint)

    def graphable(self):
        """The compress launch takes a fresh seed per record when it rounds stochastically: only the deterministic form
        can be a HIP graph node (see BatchedHSQ.graphable)."""
        return not self.random or self.keyed or (self.counter and self.rng_pairs is not None)

    def encode(self, tensors, wire_user, slot, salt, errs=None, ef_scale=None, draws=None, graph_header=None, dense=None, defer_reset=None,
               rng_slot=None, table_current=False):
        """With `errs`: error feedback in the same launch (t += ef_scale*err, err = t - decoded, both in place).
        graph_header, dense, rng_slot: see BatchedHSQ.encode."""
        counter_seed = self._counter_seed(slot) if rng_slot is None else self._counter_seed(rng_slot, reserved=True)
        if graph_header is not None:
            self._graph_tables(graph_header, dense)
        elif table_current:      # (see BatchedHSQ.encode)
            self._batch.set_table(self._dev[:self._table_words])
            self._batch.set_dense(self.dense_table_dev() if dense is not None else None, self.ndense)
        elif not self._upload(tensors, slot, 8, errs, dense):
            return False
        else:
            self._batch.set_dense(self.dense_table_dev() if dense is not None else None, self.ndense)
            self._acc_clean = False
        if self.keyed:      # gq_rng = "keyed": every bucket's draws keyed by its norm, the seed never changes
            mode, seed = native.RANDOM_DEVICE_KEYED, (salt * 0x2545F4914F6CDD1D + 0x5851F42D4C957F2D) & (2 ** 63 - 1)
        elif self.counter and counter_seed is not None:      # gq_rng = "device": keyed by the slot's device step word
            mode, seed = native.RANDOM_DEVICE_COUNTER, counter_seed
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
