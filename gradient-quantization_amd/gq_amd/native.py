"""ctypes binding of libgq_hsq.so (include/gq_hsq.h) for torch tensors.

This is the ONLY compute backend of the package: there is no CPU or eager-PyTorch
fallback.  If the library is missing, or a tensor is not on a HIP device, the calls
raise.  Tensors are passed as raw device pointers; work is enqueued on torch's
current HIP stream.
"""
import ctypes
import os

import torch

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("GQ_LIB_PATH") or os.path.join(_PKG_DIR, "libgq_hsq.so")

GQ_MAX_PARTIALS = 1024
WS_LOG_FIRST = 2 * GQ_MAX_PARTIALS + 4      # f32-sized words in front of the workspace's log: (min, max) pairs, 4 flags (include/gq_hsq.h)
GQ_FIXUP_PARTIALS = 256
RANDOM_OFF, RANDOM_GIVEN, RANDOM_DEVICE, RANDOM_DEVICE_KEYED, RANDOM_DEVICE_COUNTER = 0, 1, 2, 3, 4
ENCODE_AUTO, ENCODE_MFMA_D16K256, ENCODE_MFMA_GENERIC, ENCODE_VALU, ENCODE_PREFILTER_D16K256 = 0, 1, 2, 3, 4
ENCODE_MFMA_LDS = 5
TICKET_WORDS = 544      # GQ_TICKET_WORDS: int32 words of gq_step_tail.ticket (StepTail(ticket=...))

EXPORTS = [     # every entry point include/gq_hsq.h declares (tests/test_host_logic.py compares the two lists)
    "gq_abi_version", "gq_last_error", "gq_device_info", "gq_hsq_workspace_bytes", "gq_profile_read",
    "gq_hsq_encode", "gq_hsq_encode_ex", "gq_hsq_levels", "gq_minmax_partials", "gq_hsq_decode_sum", "gq_hsq_decode_sum_strided",
    "gq_hsq_levels_decode",
    "gq_hsq_batched_path", "gq_hsq_encode_batched", "gq_hsq_levels_batched", "gq_hsq_decode_sum_batched", "gq_hsq_decode_sum_batched_tail",
    "gq_hsq_levels_decode_batched",
    "gq_axpy_inplace", "gq_sub", "gq_mean_rows", "gq_qsgd_compress", "gq_qsgd_decode_sum", "gq_qsgd_code_bits",
    "gq_qsgd_compress_batched", "gq_qsgd_decode_sum_batched", "gq_qsgd_decode_sum_batched_tail", "gq_pvq_encode",
    "gq_launch_plan_create", "gq_launch_plan_run", "gq_launch_plan_destroy",
]
ABI_VERSION = 5
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP = -1, -2, -3      # GQ_ERR_* of include/gq_hsq.h

_lib = None


class GQNativeError(RuntimeError):
    pass


def lib():
    """Load libgq_hsq.so; fail loudly if it was not built (python gradient-quantization_amd/build.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GQNativeError(
                "libgq_hsq.so not found at %s -- build it with `python gradient-quantization_amd/build.py` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.gq_last_error.restype = ctypes.c_char_p
        L.gq_abi_version.restype = ctypes.c_int
        L.gq_hsq_workspace_bytes.restype = ctypes.c_size_t
        L.gq_launch_plan_destroy.restype = None
        for name in EXPORTS:
            getattr(L, name)  # AttributeError if the library is stale
        if L.gq_abi_version() != ABI_VERSION:
            raise GQNativeError("%s has ABI version %d, this binding is written for %d: rebuild it"
                                % (LIB_PATH, L.gq_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


CALLS = [0]      # entry-point calls that went through _check (each is one launch, or two for the wide QSGD compress): bench.py's `launches`


def _check(rc, what):
    CALLS[0] += 1
    if rc != 0:
        raise GQNativeError("%s failed (%d): %s" % (what, rc, lib().gq_last_error().decode()))


def _dev_ptr(t, dtype=None, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if t.device.type != "cuda":
        raise GQNativeError("%s is on %s: the HIP kernels need a tensor in MI355X HBM (no CPU fallback)"
                            % (name, t.device))
    if t.device.index != torch._C._cuda_getDevice():
        # the launch goes to the CURRENT device's current stream (see _stream) and is sized by its CU count
        raise GQNativeError("%s is on %s but the current device is cuda:%d: wrap the call in torch.cuda.device(...) "
                            "or call torch.cuda.set_device first" % (name, t.device, torch._C._cuda_getDevice()))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    """The current HIP stream of the current device as a raw handle (the launches go where PyTorch's would)."""
    try:    # the raw getter skips building a torch.cuda.Stream object (~10 us per call, four calls per step)
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_CODE_BYTES = {torch.uint8: 1, torch.int32: 4}
AGGREGATE_FMA = 0x100   # GQ_AGGREGATE_FMA: OR-ed into n_bit (per-tensor decodes) / bit 1 of `plain` (multi-tensor decode)
LEVELS_PACKED6 = -6     # GQ_LEVELS_PACKED6: four 6-bit levels per three bytes (include/gq_hsq.h)
PACKED6 = "packed6"     # stands in for a level dtype wherever one is passed: the section is uint8[3 * ceil(M / 4)]
_LEVEL_BYTES = {torch.uint8: 1, torch.int16: 2, torch.int32: 4, torch.float32: 0, PACKED6: LEVELS_PACKED6}


def packed6_bytes(M):
    """Bytes of a GQ_LEVELS_PACKED6 section (what the writers touch).  A section that a decode READS must be followed by one
    more readable byte (include/gq_hsq.h): `packed6_alloc_bytes` for a section in a buffer of its own."""
    return 3 * ((int(M) + 3) // 4)


def packed6_alloc_bytes(M):
    return packed6_bytes(M) + 1


def workspace_floats(M):
    nbytes = int(lib().gq_hsq_workspace_bytes(ctypes.c_int64(int(M))))
    return (nbytes + 3) // 4


def new_workspace(device, M=0):
    """Encode workspace for M subvectors: (min,max) partials | counters | worklist (f32-typed storage).
    Zero-initialised: the counter block must be zero before the first use (include/gq_hsq.h)."""
    return torch.zeros(workspace_floats(M), dtype=torch.float32, device=device)


def fixup_count(workspace, M):
    """Number of subvectors the last prefilter encode recomputed exactly (syncs).  The kernel
    zeroes its counter when it ends, so this counts the entries it left in the log: call it on a
    workspace whose log was filled with -1 beforehand (tests do)."""
    wl = workspace[WS_LOG_FIRST:WS_LOG_FIRST + M].view(torch.int32)
    return int((wl >= 0).sum().item())


def mark_worklist(workspace, M):
    workspace[WS_LOG_FIRST:WS_LOG_FIRST + M].view(torch.int32).fill_(-1)


def profile_read(slot):
    """Duration (ms) of the encode dispatch that was handed `slot` (hsq_encode(profile_slot=...), HSQBatch.encode); waits for it."""
    ms = ctypes.c_float(0.0)
    _check(lib().gq_profile_read(ctypes.c_int(slot), ctypes.byref(ms)), "gq_profile_read")
    return ms.value


def device_info(device=0):
    cus = ctypes.c_int(0)
    arch = ctypes.create_string_buffer(128)
    _check(lib().gq_device_info(ctypes.c_int(device), ctypes.byref(cus), arch, ctypes.c_size_t(128)), "gq_device_info")
    return cus.value, arch.value.decode()


def hsq_encode(grad, codebook, codes, u, partials, impl=ENCODE_AUTO, profile_slot=-1):
    """grad f32 [M*d] -> codes (uint8|int32 [M]), u f32 [M]; `partials` = new_workspace(device, M).
    profile_slot >= 0: the d16/K256 prefilter kernel's dispatch carries that slot's start/stop events (profile_read)."""
    K, d = codebook.shape
    M = grad.numel() // d
    assert grad.numel() == M * d and codes.numel() == M and u.numel() == M
    assert partials.numel() >= workspace_floats(M), "encode workspace too small: use native.new_workspace(device, M)"
    rc = lib().gq_hsq_encode_ex(_dev_ptr(grad, torch.float32, "grad"), _dev_ptr(codebook, torch.float32, "codebook"),
                                ctypes.c_int64(M), ctypes.c_int(d), ctypes.c_int(K), _dev_ptr(codes, None, "codes"),
                                ctypes.c_int(_CODE_BYTES[codes.dtype]), _dev_ptr(u, torch.float32, "u"),
                                _dev_ptr(partials, torch.float32, "partials"), ctypes.c_int(impl), ctypes.c_int(profile_slot),
                                _stream())
    _check(rc, "gq_hsq_encode")


def hsq_compress(grad, codebook, codes, u, workspace, n_bit, random_mode, r, seed, lb_ub, levels, packed6=False):
    """The whole compress (nearest_neighbor_compressor.py:63-78): gq_hsq_encode, then gq_hsq_levels, on the same stream."""
    hsq_encode(grad, codebook, codes, u, workspace)
    hsq_levels(u, n_bit, random_mode, r, seed, workspace, lb_ub, levels, packed6)


def hsq_levels(u, n_bit, random_mode, r, seed, partials, lb_ub, levels, packed6=False):
    """packed6: `levels` is the uint8 section of 3 * ceil(M / 4) bytes (four 6-bit levels per three bytes)."""
    M = u.numel()
    assert lb_ub.numel() == 2
    if packed6:
        assert levels.dtype == torch.uint8 and levels.numel() >= packed6_bytes(M)
    else:
        assert levels.numel() == M
    rp = _dev_ptr(r, torch.float32, "r") if r is not None else ctypes.c_void_p(0)
    if r is not None:
        assert r.numel() == M
    rc = lib().gq_hsq_levels(_dev_ptr(u, torch.float32, "u"), ctypes.c_int64(M), ctypes.c_int(n_bit),
                             ctypes.c_int(random_mode), rp, ctypes.c_uint64(seed & (2 ** 64 - 1)),
                             _dev_ptr(partials, torch.float32, "partials"), _dev_ptr(lb_ub, torch.float32, "lb_ub"),
                             _dev_ptr(levels, None, "levels"),
                             ctypes.c_int(LEVELS_PACKED6 if packed6 else _LEVEL_BYTES[levels.dtype]), _stream())
    _check(rc, "gq_hsq_levels")


def hsq_levels_decode(u, n_bit, random_mode, r, seed, partials, lb_ub, levels, codes, codebook, out, packed6=False):
    """gq_hsq_levels followed by gq_hsq_decode_sum (R = 1) as ONE launch: decompress(compress(g)) from the encode's
    codes and projections.  True when the library served it (d = 16, K <= 256, byte codes, byte or packed levels, aligned
    buffers); False -- nothing launched -- when the caller has to make the two calls."""
    M = u.numel()
    K, d = codebook.shape
    if (d != 16 or K > 256 or codes.dtype != torch.uint8 or (not packed6 and levels.dtype != torch.uint8)
            or u.data_ptr() % 4 or codes.data_ptr() % 4 or (not packed6 and levels.data_ptr() % 4) or out.data_ptr() % 16
            or codebook.data_ptr() % 16):
        return False
    assert lb_ub.numel() == 2 and codes.numel() == M and out.numel() == M * d
    if packed6:
        assert levels.dtype == torch.uint8 and levels.numel() >= packed6_bytes(M)
    else:
        assert levels.numel() == M
    rp = _dev_ptr(r, torch.float32, "r") if r is not None else ctypes.c_void_p(0)
    if r is not None:
        assert r.numel() == M
    rc = lib().gq_hsq_levels_decode(_dev_ptr(u, torch.float32, "u"), ctypes.c_int64(M), ctypes.c_int(n_bit),
                                    ctypes.c_int(random_mode), rp, ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                    _dev_ptr(partials, torch.float32, "partials"), _dev_ptr(lb_ub, torch.float32, "lb_ub"),
                                    _dev_ptr(levels, None, "levels"), ctypes.c_int(LEVELS_PACKED6 if packed6 else 1),
                                    _dev_ptr(codes, None, "codes"), _dev_ptr(codebook, torch.float32, "codebook"),
                                    ctypes.c_int(K), _dev_ptr(out, torch.float32, "out"), _stream())
    if rc == ERR_UNSUPPORTED:
        return False
    _check(rc, "gq_hsq_levels_decode")
    return True


def minmax_partials(v, partials):
    _check(lib().gq_minmax_partials(_dev_ptr(v, torch.float32, "v"), ctypes.c_int64(v.numel()),
                                    _dev_ptr(partials, torch.float32, "partials"), _stream()), "gq_minmax_partials")


def hsq_decode_sum(codes, levels, lb_ub, codebook, n_bit, out, R=1):
    """codes [R,M], levels [R,M] (int) or f32 norms [R,M], lb_ub [R,2] -> out f32 [M*d] (mean over R)."""
    K, d = codebook.shape
    M = codes.numel() // R
    assert codes.numel() == R * M and levels.numel() == R * M and out.numel() == M * d
    lvl_bytes = _LEVEL_BYTES[levels.dtype]
    if lvl_bytes:
        assert lb_ub is not None and lb_ub.numel() == 2 * R
        lp = _dev_ptr(lb_ub, torch.float32, "lb_ub")
    else:
        lp = ctypes.c_void_p(0)
    rc = lib().gq_hsq_decode_sum(_dev_ptr(codes, None, "codes"), ctypes.c_int(_CODE_BYTES[codes.dtype]),
                                 _dev_ptr(levels, None, "levels"), ctypes.c_int(lvl_bytes), lp,
                                 _dev_ptr(codebook, torch.float32, "codebook"), ctypes.c_int(R), ctypes.c_int64(M),
                                 ctypes.c_int(d), ctypes.c_int(K), ctypes.c_int(n_bit if lvl_bytes else 0),
                                 _dev_ptr(out, torch.float32, "out"), _stream())
    _check(rc, "gq_hsq_decode_sum")


def hsq_decode_sum_packed(wire, M, codebook, n_bit, out, R, codes_off=0, levels_off=None, lbub_off=None,
                          code_dtype=torch.uint8, level_dtype=torch.uint8):
    """Decode the all-gathered wire buffer `wire` (uint8 [R, P]; per rank: codes at codes_off,
    levels at levels_off, (lb, ub) f32 at lbub_off) and average over the R ranks -- no repack."""
    K, d = codebook.shape
    assert wire.dtype == torch.uint8 and wire.dim() == 2 and wire.shape[0] == R and wire.is_contiguous()
    P = wire.shape[1]
    cb_, lb_ = _CODE_BYTES[code_dtype], _LEVEL_BYTES[level_dtype]
    assert out.numel() == M * d
    if level_dtype == PACKED6:      # every group is fetched as one 32-bit word: a byte behind the section must exist (gq_hsq.h)
        assert levels_off + packed6_bytes(M) + 1 <= P, "a packed level section must be followed by one more byte of its row"
    base = wire.data_ptr()
    _dev_ptr(wire, torch.uint8, "wire")
    rc = lib().gq_hsq_decode_sum_strided(
        ctypes.c_void_p(base + codes_off), ctypes.c_int(cb_), ctypes.c_int64(P),
        ctypes.c_void_p(base + levels_off), ctypes.c_int(lb_), ctypes.c_int64(P),
        ctypes.c_void_p(base + lbub_off), ctypes.c_int64(P),
        _dev_ptr(codebook, torch.float32, "codebook"), ctypes.c_int(R), ctypes.c_int64(M), ctypes.c_int(d),
        ctypes.c_int(K), ctypes.c_int(n_bit), _dev_ptr(out, torch.float32, "out"), _stream())
    _check(rc, "gq_hsq_decode_sum_strided")


class _HSQBatchStruct(ctypes.Structure):      # gq_hsq_batch (include/gq_hsq.h)
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("d", ctypes.c_int32), ("K", ctypes.c_int32),
                ("code_bytes", ctypes.c_int32), ("level_bytes", ctypes.c_int32), ("n_bit", ctypes.c_int32),
                ("nseg", ctypes.c_int32), ("profile_slot", ctypes.c_int32), ("ntiles", ctypes.c_int64),
                ("seg_table", ctypes.c_void_p), ("tile_seg", ctypes.c_void_p), ("codebook", ctypes.c_void_p),
                ("u_flat", ctypes.c_void_p), ("seg_minmax", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                ("dense_table", ctypes.c_void_p), ("ndense", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_NAN = float("nan")
BATCH_PREFILTER, BATCH_PAGED, BATCH_EXACT = 1, 2, 3


class HSQBatch(object):
    """The multi-tensor HSQ launches of one group of tensors that share a codebook: the gq_hsq_batch descriptor is
    filled ONCE (every pointer and size that does not change from step to step), a step's calls only hand over the
    wire and the stream -- ~3 us of marshalling per launch (the quantizer step is host-bound).  Which kernels serve
    the shape is the library's decision (gq_hsq_batched_path); `path` is 0 when none does."""

    def __init__(self, seg_table, tile_seg, nseg, ntiles, codebook, code_dtype, level_dtype, n_bit, u_flat=None,
                 seg_minmax=None, workspace=None):
        self.L = lib()
        self.keep = (seg_table, tile_seg, codebook, u_flat, seg_minmax, workspace)     # the pointers below stay valid
        self.device = seg_table.device.index
        K, d = int(codebook.shape[0]), int(codebook.shape[1])
        if workspace is not None:
            assert workspace.numel() >= workspace_floats(ntiles * 64)
        opt = lambda t, dt, nm: _dev_ptr(t, dt, nm).value if t is not None else None
        self.s = _HSQBatchStruct(ctypes.sizeof(_HSQBatchStruct), d, K, _CODE_BYTES[code_dtype], _LEVEL_BYTES[level_dtype],
                                 int(n_bit), int(nseg), -1, int(ntiles), _dev_ptr(seg_table, torch.int64, "seg_table").value,
                                 _dev_ptr(tile_seg, torch.int32, "tile_seg").value,
                                 _dev_ptr(codebook, torch.float32, "codebook").value, opt(u_flat, torch.float32, "u_flat"),
                                 opt(seg_minmax, torch.int32, "seg_minmax"), opt(workspace, torch.float32, "workspace"), None, 0, 0)
        self.ref = ctypes.byref(self.s)
        self.path = int(self.L.gq_hsq_batched_path(self.ref))

    def set_table(self, seg_table):
        """Another copy of the segment table for the launches that follow (a captured graph's own, which nobody rewrites)."""
        self.keep_table = seg_table
        self.s.seg_table = _dev_ptr(seg_table, torch.int64, "seg_table").value

    def set_dense(self, dense_table, ndense):
        """dense_table (int64 [ndense, 3] on the device: source pointer, byte offset in one user's wire, elements): the level
        launch also copies the uncompressed tensors into the wire.  None: no copies."""
        self.keep_dense = dense_table
        self.s.dense_table = _dev_ptr(dense_table, torch.int64, "dense_table").value if dense_table is not None else None
        self.s.ndense = int(ndense) if dense_table is not None else 0

    def part(self, seg_table, tile_seg, nseg, ntiles):
        """The same configuration over another table (the head / tail of a split decode)."""
        codebook = self.keep[2]
        code_dtype = {v: k for k, v in _CODE_BYTES.items()}[self.s.code_bytes]
        level_dtype = {v: k for k, v in _LEVEL_BYTES.items()}[self.s.level_bytes]
        return HSQBatch(seg_table, tile_seg, nseg, ntiles, codebook, code_dtype, level_dtype, self.s.n_bit)

    def _wire(self, wire):
        if wire.device.index != self.device or wire.device.index != torch._C._cuda_getDevice():
            raise GQNativeError("wire is on %s, this batch's launches are for cuda:%d (the current device must match)"
                                % (wire.device, self.device))
        return ctypes.c_void_p(wire.data_ptr())

    def encode(self, wire, ef_scale=None, profile_slot=-1):
        """ef_scale given: the error-feedback form (seg_table[:, 7] = error buffers, grads updated in place)."""
        self.s.profile_slot = profile_slot
        rc = self.L.gq_hsq_encode_batched(self.ref, self._wire(wire), ctypes.c_float(_NAN if ef_scale is None else ef_scale),
                                          _stream())
        self.s.profile_slot = -1
        _check(rc, "gq_hsq_encode_batched")

    def levels(self, wire, random_mode, seed, r_flat=None, write_error=False):
        """r_flat: the reference's draws laid out like u_flat (random_mode = RANDOM_GIVEN).  write_error: also
        error = grad - decode(wire) into the buffers of seg_table[:, 7] (after an encode with ef_scale)."""
        rp = _dev_ptr(r_flat, torch.float32, "r_flat") if r_flat is not None else ctypes.c_void_p(0)
        rc = self.L.gq_hsq_levels_batched(self.ref, self._wire(wire), ctypes.c_int(random_mode),
                                          ctypes.c_uint64(seed & (2 ** 64 - 1)), rp, ctypes.c_int(1 if write_error else 0),
                                          _stream())
        _check(rc, "gq_hsq_levels_batched")

    def levels_decode(self, wire, random_mode, seed, r_flat, write_error, out, plain=False, tail=None):
        """gq_hsq_levels_decode_batched: levels (+ residual) of the one payload `wire`, its decode into `out` and the step's tail
        (StepTail with rows inside `wire` and a ticket) in ONE launch."""
        rp = _dev_ptr(r_flat, torch.float32, "r_flat") if r_flat is not None else ctypes.c_void_p(0)
        rc = self.L.gq_hsq_levels_decode_batched(self.ref, self._wire(wire), ctypes.c_int(random_mode), ctypes.c_uint64(seed & (2 ** 64 - 1)), rp,
                                                 ctypes.c_int(1 if write_error else 0), _dev_ptr(out, torch.float32, "out"),
                                                 ctypes.c_int(1 if plain else 0), tail.ref if tail is not None else None, _stream())
        _check(rc, "gq_hsq_levels_decode_batched")

    def decode(self, gathered, R, out, plain=False, fma=False, tail=None):
        """Mean of the R payloads (plain: the decompress of ONE payload as the reference returns it, a -0 stays -0).
        fma: GQ_AGGREGATE_FMA for this launch (opt-in; not with plain).  tail (StepTail): the aggregate's small per-step
        work rides in the same launch (gq_hsq_decode_sum_batched_tail)."""
        assert gathered.dtype == torch.uint8 and gathered.dim() == 2 and gathered.shape[0] == R and gathered.stride(1) == 1
        stride = int(gathered.stride(0)) if R > 1 else int(gathered.shape[1])
        if tail is not None:
            rc = self.L.gq_hsq_decode_sum_batched_tail(self.ref, self._wire(gathered), ctypes.c_int64(stride), ctypes.c_int(R),
                                                       _dev_ptr(out, torch.float32, "out"),
                                                       ctypes.c_int(1 if plain else (2 if fma else 0)), tail.ref, _stream())
            _check(rc, "gq_hsq_decode_sum_batched_tail")
            return
        rc = self.L.gq_hsq_decode_sum_batched(self.ref, self._wire(gathered), ctypes.c_int64(stride), ctypes.c_int(R),
                                              _dev_ptr(out, torch.float32, "out"),
                                              ctypes.c_int(1 if plain else (2 if fma else 0)), _stream())
        _check(rc, "gq_hsq_decode_sum_batched")


def hsq_batched_path(d, K, code_dtype, nseg=1):
    """Which multi-tensor encode serves (d, K, code width, number of tensors): BATCH_PREFILTER / PAGED / EXACT or 0."""
    s = _HSQBatchStruct(ctypes.sizeof(_HSQBatchStruct), int(d), int(K), _CODE_BYTES[code_dtype], 1, 6, int(nseg), -1, 1, 1, 1, 1,
                        None, None, None, None, 0, 0)      # the path depends on the shape only; the (non-null) pointers are not read
    return int(lib().gq_hsq_batched_path(ctypes.byref(s)))


class _StepTailStruct(ctypes.Structure):      # gq_step_tail (include/gq_hsq.h)
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("rows_R", ctypes.c_int32), ("rows", ctypes.c_void_p),
                ("row_stride_bytes", ctypes.c_int64), ("n", ctypes.c_int64), ("out", ctypes.c_void_p),
                ("rng_state", ctypes.c_void_p), ("reset_dst", ctypes.c_void_p), ("reset_src", ctypes.c_void_p),
                ("rng_pairs", ctypes.c_int32), ("reset_words", ctypes.c_int32), ("ticket", ctypes.c_void_p)]


class StepTail(object):
    """What gq_mean_rows does, as a rider of a decode-mean launch: rows ([R, n] float32 view, rows may be strided) -> out[n];
    rng_state (int64 [pairs, 2]) stepped; reset = (dst, src) int64 tensors copied src -> dst.  Any part may be None."""

    def __init__(self, rows=None, out=None, rng_state=None, reset=None, ticket=None):
        """ticket (int32 [1], zero): gq_hsq_levels_decode_batched's last-workgroup counter (HSQBatch.levels_decode)."""
        self.keep = (rows, out, rng_state, reset, ticket)
        R, n, stride, rp, op = 1, 0, 0, None, None
        if rows is not None:
            assert rows.dtype == torch.float32 and rows.dim() == 2 and rows.stride(1) == 1
            R, n = int(rows.shape[0]), int(rows.shape[1])
            stride = int(rows.stride(0)) * 4 if R > 1 else n * 4
            rp, op = rows.data_ptr(), _dev_ptr(out, torch.float32, "out").value
        sp, pairs = (rng_state.data_ptr(), int(rng_state.shape[0])) if rng_state is not None else (None, 0)
        rd, rs, rw = _reset_args(reset)
        self.s = _StepTailStruct(ctypes.sizeof(_StepTailStruct), R, rp, stride, n, op, sp, rd.value, rs.value, pairs, rw.value,
                                 _dev_ptr(ticket, torch.int32, "ticket").value if ticket is not None else None)
        self.ref = ctypes.byref(self.s)


def _reset_args(reset):
    """reset = (dst, src): int64 device tensors of one size (accumulators, their empty state), or None."""
    if reset is None:
        return ctypes.c_void_p(0), ctypes.c_void_p(0), ctypes.c_int(0)
    dst, src = reset
    assert dst.dtype == torch.int64 and src.dtype == torch.int64 and dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
    return _dev_ptr(dst, torch.int64, "reset_dst"), _dev_ptr(src, torch.int64, "reset_src"), ctypes.c_int(int(dst.numel()))


def mean_rows(rows, out, rng_state=None, reset=None):
    """out[i] = (+0 + rows[0, i] + ... + rows[R-1, i]) / R for a [R, n] float32 view whose rows may be strided (the dense
    region of the gathered wire): torch.stack(...).mean(0) with the CPU's arithmetic.  rng_state (int64 [pairs, 2], the
    { seed, step } words of RANDOM_DEVICE_COUNTER): the same launch adds one to every step word.  reset = (dst, src): it
    also copies src over dst (the next step's accumulators back to their empty state)."""
    assert rows.dtype == torch.float32 and rows.dim() == 2 and rows.stride(1) == 1
    R, n = int(rows.shape[0]), int(rows.shape[1])
    stride = int(rows.stride(0)) * 4 if R > 1 else n * 4
    if rows.device.index != torch._C._cuda_getDevice():
        raise GQNativeError("rows are on %s but the current device is cuda:%d" % (rows.device, torch._C._cuda_getDevice()))
    sp, pairs = (ctypes.c_void_p(rng_state.data_ptr()), int(rng_state.shape[0])) if rng_state is not None else (ctypes.c_void_p(0), 0)
    _check(lib().gq_mean_rows(ctypes.c_void_p(rows.data_ptr()), ctypes.c_int64(stride), ctypes.c_int(R), ctypes.c_int64(n),
                                   _dev_ptr(out, torch.float32, "out"), sp, ctypes.c_int(pairs), *_reset_args(reset), _stream()), "gq_mean_rows")


def rng_step(rng_state, reset=None):
    """step += 1 in every { seed, step } pair of rng_state (int64 [pairs, 2]; RANDOM_DEVICE_COUNTER); reset: see mean_rows."""
    assert rng_state.dtype == torch.int64 and rng_state.dim() == 2 and rng_state.shape[1] == 2 and rng_state.is_contiguous()
    _check(lib().gq_mean_rows(ctypes.c_void_p(0), ctypes.c_int64(0), ctypes.c_int(1), ctypes.c_int64(0), ctypes.c_void_p(0),
                              ctypes.c_void_p(rng_state.data_ptr()), ctypes.c_int(int(rng_state.shape[0])), *_reset_args(reset), _stream()),
           "gq_mean_rows (step)")



class _QSGDBatchStruct(ctypes.Structure):     # gq_qsgd_batch (include/gq_hsq.h)
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("n_bit", ctypes.c_int32), ("bits", ctypes.c_int32),
                ("wide", ctypes.c_int32), ("nseg", ctypes.c_int32), ("bucket_hint", ctypes.c_int32), ("nitems", ctypes.c_int64),
                ("seg_table", ctypes.c_void_p), ("item_seg", ctypes.c_void_p), ("norm_bits", ctypes.c_void_p),
                ("dense_table", ctypes.c_void_p), ("ndense", ctypes.c_int32), ("reserved2", ctypes.c_int32)]


class QSGDBatch(object):
    """The multi-tensor QSGD launches on the packed wire (buckets, or chunks of wide buckets: `wide`)."""

    def __init__(self, seg_table, item_seg, nseg, nitems, n_bit, bits, wide=False, norm_bits=None, bucket_hint=0):
        """bucket_hint: the bucket width most elements have (0: unknown) -- picks the lanes the bucketed kernels give a bucket."""
        self.L = lib()
        self.keep = (seg_table, item_seg, norm_bits)
        self.s = _QSGDBatchStruct(ctypes.sizeof(_QSGDBatchStruct), int(n_bit), int(bits), 1 if wide else 0, int(nseg), int(bucket_hint),
                                  int(nitems), _dev_ptr(seg_table, torch.int64, "seg_table").value,
                                  _dev_ptr(item_seg, torch.int32, "item_seg").value,
                                  _dev_ptr(norm_bits, torch.int32, "norm_bits").value if norm_bits is not None else None, None, 0, 0)
        self.ref = ctypes.byref(self.s)

    def set_table(self, seg_table):
        """As HSQBatch.set_table."""
        self.keep_table = seg_table
        self.s.seg_table = _dev_ptr(seg_table, torch.int64, "seg_table").value

    def set_dense(self, dense_table, ndense):
        """As HSQBatch.set_dense: the compress launch also copies the uncompressed tensors into the wire."""
        self.keep_dense = dense_table
        self.s.dense_table = _dev_ptr(dense_table, torch.int64, "dense_table").value if dense_table is not None else None
        self.s.ndense = int(ndense) if dense_table is not None else 0

    def part(self, seg_table, item_seg, nseg, nitems):
        return QSGDBatch(seg_table, item_seg, nseg, nitems, self.s.n_bit, self.s.bits, bool(self.s.wide), self.keep[2], self.s.bucket_hint)

    def compress(self, wire, random_mode, seed, ef_scale=None):
        """ef_scale given: error feedback in the same pass (seg_table[:, 7] = error buffers)."""
        rc = self.L.gq_qsgd_compress_batched(self.ref, _dev_ptr(wire, torch.uint8, "wire"), ctypes.c_int(random_mode),
                                             ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                             ctypes.c_float(_NAN if ef_scale is None else ef_scale), _stream())
        _check(rc, "gq_qsgd_compress_batched")

    def decode(self, gathered, R, out, plain=False, tail=None):
        """tail (StepTail): gq_qsgd_decode_sum_batched_tail."""
        assert gathered.dtype == torch.uint8 and gathered.dim() == 2 and gathered.shape[0] == R and gathered.is_contiguous()
        if tail is not None:
            rc = self.L.gq_qsgd_decode_sum_batched_tail(self.ref, _dev_ptr(gathered, torch.uint8, "gathered"),
                                                        ctypes.c_int64(gathered.shape[1]), ctypes.c_int(R),
                                                        _dev_ptr(out, torch.float32, "out"), ctypes.c_int(1 if plain else 0), tail.ref, _stream())
            _check(rc, "gq_qsgd_decode_sum_batched_tail")
            return
        rc = self.L.gq_qsgd_decode_sum_batched(self.ref, _dev_ptr(gathered, torch.uint8, "gathered"),
                                               ctypes.c_int64(gathered.shape[1]), ctypes.c_int(R),
                                               _dev_ptr(out, torch.float32, "out"), ctypes.c_int(1 if plain else 0), _stream())
        _check(rc, "gq_qsgd_decode_sum_batched")


class _PVQStage1(ctypes.Structure):           # gq_pvq_stage1
    _fields_ = [("codes1", ctypes.c_void_p), ("code1_bytes", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("norm1", ctypes.c_void_p), ("codebook1", ctypes.c_void_p)]


def pvq_encode(grad, c_dagger, codes, u, workspace, random_mode, r, seed, stage1=None):
    """stage1 = (codes1, norm1, codebook1): the encode of  grad - codebook1[codes1] * norm1  without materialising it
    (the ResidualCompressor's second stage)."""
    K, d = c_dagger.shape
    M = grad.numel() // d
    assert grad.numel() == M * d and codes.numel() == M and u.numel() == M
    rp = _dev_ptr(r, torch.float32, "r") if r is not None else ctypes.c_void_p(0)
    st1 = ctypes.c_void_p(0)
    if stage1 is not None:
        codes1, norm1, codebook1 = stage1
        assert codes1.numel() == M and norm1.numel() == M and codebook1.shape[1] == d
        st1 = ctypes.byref(_PVQStage1(_dev_ptr(codes1, None, "codes1").value, _CODE_BYTES[codes1.dtype], 0,
                                      _dev_ptr(norm1, torch.float32, "norm1").value,
                                      _dev_ptr(codebook1, torch.float32, "codebook1").value))
    rc = lib().gq_pvq_encode(_dev_ptr(grad, torch.float32, "grad"), _dev_ptr(c_dagger, torch.float32, "c_dagger"),
                             ctypes.c_int64(M), ctypes.c_int(d), ctypes.c_int(K), ctypes.c_int(random_mode), rp,
                             ctypes.c_uint64(seed & (2 ** 64 - 1)), _dev_ptr(codes, None, "codes"),
                             ctypes.c_int(_CODE_BYTES[codes.dtype]), _dev_ptr(u, torch.float32, "u"),
                             _dev_ptr(workspace, torch.float32, "workspace"), st1, _stream())
    _check(rc, "gq_pvq_encode")


def pvq_encode_residual(grad, codes1, norm1, codebook1, c_dagger, codes, u, workspace, random_mode, r, seed):
    pvq_encode(grad, c_dagger, codes, u, workspace, random_mode, r, seed, stage1=(codes1, norm1, codebook1))


class LaunchPlan(object):
    """gq_launch_plan_*: the kernel nodes of a captured graph (torch.cuda.CUDAGraph(keep_graph=True)) as plain launches on torch's
    current stream.  `replay()` like the graph's own; raises GQNativeError at construction when the graph is not one chain of
    kernel launches (the caller keeps the graph's replay)."""

    def __init__(self, graph):
        self.graph = graph          # the owner of the launches' argument storage
        self.L = lib()
        self.plan, n = ctypes.c_void_p(0), ctypes.c_int(0)
        rc = self.L.gq_launch_plan_create(ctypes.c_void_p(int(graph.raw_cuda_graph())), ctypes.byref(self.plan), ctypes.byref(n))
        if rc != 0:
            raise GQNativeError("gq_launch_plan_create failed (%d): %s" % (rc, self.L.gq_last_error().decode()))
        self.nodes = n.value
        self._run = self.L.gq_launch_plan_run

    def replay(self):
        rc = self._run(self.plan, _stream())
        if rc != 0:
            raise GQNativeError("gq_launch_plan_run failed (%d): %s" % (rc, self.L.gq_last_error().decode()))

    def __del__(self):
        try:
            if self.plan:
                self.L.gq_launch_plan_destroy(self.plan)
                self.plan = ctypes.c_void_p(0)
        except Exception:
            pass


def qsgd_code_bits(n_bit, random_mode):
    return int(lib().gq_qsgd_code_bits(ctypes.c_int(n_bit), ctypes.c_int(random_mode)))


QSGD_WIDE_CHUNK = 1024     # GQ_QSGD_WIDE_CHUNK (include/gq_hsq.h)


def axpy_inplace(grad, err, scale):
    assert grad.numel() == err.numel()
    _check(lib().gq_axpy_inplace(_dev_ptr(grad, torch.float32, "grad"), _dev_ptr(err, torch.float32, "err"),
                                 ctypes.c_float(scale), ctypes.c_int64(grad.numel()), _stream()), "gq_axpy_inplace")


def sub(grad, decoded, err):
    assert grad.numel() == decoded.numel() == err.numel()
    _check(lib().gq_sub(_dev_ptr(grad, torch.float32, "grad"), _dev_ptr(decoded, torch.float32, "decoded"),
                        _dev_ptr(err, torch.float32, "err"), ctypes.c_int64(grad.numel()), _stream()), "gq_sub")


def qsgd_compress(grad, d, n_bit, random_mode, r, seed, norm, signs, levels):
    Mb = grad.numel() // d
    assert grad.numel() == Mb * d and norm.numel() == Mb and signs.numel() == Mb * d and levels.numel() == Mb * d
    rp = _dev_ptr(r, torch.float32, "r") if r is not None else ctypes.c_void_p(0)
    sg = signs.view(torch.uint8) if signs.dtype == torch.bool else signs
    rc = lib().gq_qsgd_compress(_dev_ptr(grad, torch.float32, "grad"), ctypes.c_int64(Mb), ctypes.c_int(d),
                                ctypes.c_int(n_bit), ctypes.c_int(random_mode), rp,
                                ctypes.c_uint64(seed & (2 ** 64 - 1)), _dev_ptr(norm, torch.float32, "norm"),
                                _dev_ptr(sg, torch.uint8, "signs"), _dev_ptr(levels, None, "levels"),
                                ctypes.c_int(_LEVEL_BYTES[levels.dtype]), _stream())
    _check(rc, "gq_qsgd_compress")


def qsgd_decode_sum(norm, signs, levels, d, n_bit, out, R=1):
    Mb = norm.numel() // R
    assert signs.numel() == R * Mb * d and levels.numel() == R * Mb * d and out.numel() == Mb * d
    sg = signs.view(torch.uint8) if signs.dtype == torch.bool else signs
    rc = lib().gq_qsgd_decode_sum(_dev_ptr(norm, torch.float32, "norm"), _dev_ptr(sg, torch.uint8, "signs"),
                                  _dev_ptr(levels, None, "levels"), ctypes.c_int(_LEVEL_BYTES[levels.dtype]),
                                  ctypes.c_int(R), ctypes.c_int64(Mb), ctypes.c_int(d), ctypes.c_int(n_bit),
                                  _dev_ptr(out, torch.float32, "out"), _stream())
    _check(rc, "gq_qsgd_decode_sum")
