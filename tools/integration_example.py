"""INTEGRATION.md section 3, runnable: the ctypes stubs a maintainer would put into the reference, checked against this
package's own classes.  Part 1: NearestNeighborCompressor.compress through gq_hsq_encode + gq_hsq_levels.  Part 2: the
multi-tensor launches (gq_hsq_batch descriptor, gq_step_tail) that replace PSQuantizer's per-parameter loops, three tensors +
two small dense ones, two users, against decompress(compress()) of the per-tensor class and torch.stack().mean(0).
    python tools/integration_example.py        (prints one line per check; exit code 1 if any differs)"""
import ctypes, torch, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
from gq_amd.codebook import load_codebook
_gq = ctypes.CDLL(os.path.join(ROOT, "gradient-quantization_amd", "libgq_hsq.so"))
_gq.gq_last_error.restype = ctypes.c_char_p
_gq.gq_hsq_workspace_bytes.restype = ctypes.c_size_t
P = lambda t: ctypes.c_void_p(t.data_ptr())
ok = True
def check(what, cond):
    global ok
    ok = ok and bool(cond)
    print("%-72s %s" % (what, "equal" if cond else "DIFFERS"))

# ---------------------------------------------------------------------------------------------- part 1: one tensor
class C: pass
self = C(); self.dim = 16; self.K = 256; self.code_dtype = torch.uint8; self.compressed_norm = True; self.n_bit = 6
self.codewords = torch.from_numpy(load_codebook(16, 256)).cuda()
def compress(self, vec):                                   # drop-in body (nearest_neighbor_compressor.py:63-78)
    vec = vec.contiguous().view(-1)
    M = vec.numel() // self.dim
    codes = torch.empty(M, dtype=self.code_dtype, device=vec.device)
    u = torch.empty(M, dtype=torch.float32, device=vec.device)
    part = torch.zeros(_gq.gq_hsq_workspace_bytes(ctypes.c_int64(M)) // 4 + 1, dtype=torch.float32, device=vec.device)   # encode workspace, zeroed once
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = _gq.gq_hsq_encode(P(vec), P(self.codewords), ctypes.c_int64(M), self.dim, self.K,
                           P(codes), codes.element_size(), P(u), P(part), st)
    assert rc == 0, _gq.gq_last_error()
    if not self.compressed_norm:
        return [u, codes]
    lb_ub = torch.empty(2, dtype=torch.float32, device=vec.device)
    l = torch.empty(M, dtype=torch.int32, device=vec.device)
    rc = _gq.gq_hsq_levels(P(u), ctypes.c_int64(M), self.n_bit, 0, None, ctypes.c_uint64(0),
                           P(part), P(lb_ub), P(l), 4, st)
    assert rc == 0, _gq.gq_last_error()
    return [(lb_ub[0], lb_ub[1], l), codes]
x = torch.randn(4096 * 16, device="cuda")
(lb, ub, l), codes = compress(self, x)
torch.cuda.synchronize()
from gq_amd.compressors import NearestNeighborCompressor
from argparse import Namespace
a = Namespace(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=0, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256)
ref = NearestNeighborCompressor(x.numel(), x.shape, a).compress(x)
check("per-tensor stub: codes", torch.equal(codes, ref[1]))
check("per-tensor stub: levels", torch.equal(l, ref[0][2].to(torch.int32)))
check("per-tensor stub: lb, ub", float(lb) == float(ref[0][0]) and float(ub) == float(ref[0][1]))

# ---------------------------------------------------------------------------------------------- part 2: a model's tensors per launch
class GQHSQBatch(ctypes.Structure):                       # gq_hsq_batch, include/gq_hsq.h
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("d", ctypes.c_int32), ("K", ctypes.c_int32),
                ("code_bytes", ctypes.c_int32), ("level_bytes", ctypes.c_int32), ("n_bit", ctypes.c_int32),
                ("nseg", ctypes.c_int32), ("profile_slot", ctypes.c_int32), ("ntiles", ctypes.c_int64),
                ("seg_table", ctypes.c_void_p), ("tile_seg", ctypes.c_void_p), ("codebook", ctypes.c_void_p),
                ("u_flat", ctypes.c_void_p), ("seg_minmax", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                ("dense_table", ctypes.c_void_p), ("ndense", ctypes.c_int32), ("reserved", ctypes.c_int32)]

class GQStepTail(ctypes.Structure):                       # gq_step_tail
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("rows_R", ctypes.c_int32), ("rows", ctypes.c_void_p),
                ("row_stride_bytes", ctypes.c_int64), ("n", ctypes.c_int64), ("out", ctypes.c_void_p),
                ("rng_state", ctypes.c_void_p), ("reset_dst", ctypes.c_void_p), ("reset_src", ctypes.c_void_p),
                ("rng_pairs", ctypes.c_int32), ("reset_words", ctypes.c_int32), ("ticket", ctypes.c_void_p)]

dev = torch.device("cuda")
up = lambda n: (n + 15) // 16 * 16
shapes, small = [(64, 48), (1030, 16), (20000,)], [(10,), (64,)]      # > 1000 elements: through the codebook; the rest travel as f32
U = 2                                                                  # users (ranks) whose wires are averaged
grads = [[torch.randn(s, device=dev) * 1e-2 for s in shapes + small] for _ in range(U)]
# one user's wire: per compressed tensor  codes u8[M] | levels u8[M] | lb, ub  (16-byte aligned sections), then the small tensors' f32
Ms = [torch.Size(s).numel() // 16 for s in shapes]
table, tile_seg, off, tile, out_off = [], [], 0, 0, 0
for s, M in enumerate(Ms):
    codes_off, levels_off = off, off + up(M)
    lbub_off = levels_off + up(M)
    off = lbub_off + 16
    table.append([0, M, tile, codes_off, levels_off, lbub_off, out_off, 0])       # { grad pointer (per step), M, first tile, byte offsets, float offset in out, error buffer }
    nt = (M + 63) // 64
    tile_seg += [s] * nt
    tile += nt
    out_off += M * 16
dense_off, dense = off, []
for s in small:
    n = torch.Size(s).numel()
    dense.append([0, off, n])                                                       # { source pointer (per step), byte offset in one user's wire, elements }
    off += 4 * n
user_bytes = up(off)
ntiles, nseg = tile, len(Ms)
tile_seg_d = torch.tensor(tile_seg, dtype=torch.int32, device=dev)
u_flat = torch.empty(ntiles * 64, dtype=torch.float32, device=dev)
seg_minmax = torch.tensor([[0xFFFFFFFF, 0]] * nseg, dtype=torch.int64).to(torch.int32).to(dev)   # identities of the mapped (min, max)
minmax_empty = seg_minmax.clone()
workspace = torch.zeros(_gq.gq_hsq_workspace_bytes(ctypes.c_int64(ntiles * 64)) // 4 + 1, dtype=torch.float32, device=dev)
gathered = torch.zeros((U, user_bytes), dtype=torch.uint8, device=dev)
out = torch.empty(out_off, dtype=torch.float32, device=dev)
dense_mean = torch.empty(sum(d[2] for d in dense), dtype=torch.float32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
nan = ctypes.c_float(float("nan"))                         # ef_scale = NaN: no error feedback
for u in range(U):                                          # record(user): one encode + one level launch for every tensor
    for s in range(nseg):
        table[s][0] = grads[u][s].data_ptr()
    for k in range(len(small)):
        dense[k][0] = grads[u][nseg + k].data_ptr()
    seg_table = torch.tensor(table, dtype=torch.int64, device=dev)
    dense_table = torch.tensor(dense, dtype=torch.int64, device=dev)
    seg_minmax.copy_(minmax_empty)
    b = GQHSQBatch(ctypes.sizeof(GQHSQBatch), 16, 256, 1, 1, 6, nseg, -1, ntiles, seg_table.data_ptr(), tile_seg_d.data_ptr(),
                   self.codewords.data_ptr(), u_flat.data_ptr(), seg_minmax.data_ptr(), workspace.data_ptr(), dense_table.data_ptr(), len(small), 0)
    assert _gq.gq_hsq_batched_path(ctypes.byref(b)) == 1, _gq.gq_last_error()      # 1 prefilter, 2 paged prefilter, 3 exact scoring
    rc = _gq.gq_hsq_encode_batched(ctypes.byref(b), P(gathered[u]), nan, st)
    assert rc == 0, _gq.gq_last_error()
    rc = _gq.gq_hsq_levels_batched(ctypes.byref(b), P(gathered[u]), 0, ctypes.c_uint64(0), None, 0, st)      # + the small tensors copied into the wire
    assert rc == 0, _gq.gq_last_error()
    torch.cuda.synchronize()                               # (the tables above are rebuilt per user in this example)
# apply(): ONE decode-mean over the U wires; the small tensors' mean rides in the same launch (gq_step_tail)
rows = gathered[:, dense_off:dense_off + 4 * dense_mean.numel()]
tail = GQStepTail(ctypes.sizeof(GQStepTail), U, rows.data_ptr(), gathered.stride(0), dense_mean.numel(), dense_mean.data_ptr(), None, None, None, 0, 0, None)
rc = _gq.gq_hsq_decode_sum_batched_tail(ctypes.byref(b), P(gathered), ctypes.c_int64(user_bytes), U, P(out), 0, ctypes.byref(tail), st)
assert rc == 0, _gq.gq_last_error()
torch.cuda.synchronize()
# the reference's arithmetic for the same step: decompress(compress(g)) per user and tensor, stack().mean(0)
o = 0
for s, shp in enumerate(shapes):
    comp = NearestNeighborCompressor(torch.Size(shp).numel(), torch.Size(shp), a)
    dec = [comp.decompress(comp.compress(grads[u][s])) for u in range(U)]
    want = torch.stack([d.cpu() for d in dec]).mean(0).to(dev)            # (the CPU's mean: what the reference computes with --no-cuda)
    got = out[o:o + want.numel()].view(shp)
    check("multi-tensor launches: aggregate of tensor %s" % (shp,), torch.equal(got.view(torch.int32), want.view(torch.int32)))
    o += want.numel()
o = 0
for k, shp in enumerate(small):
    want = torch.stack([grads[u][nseg + k].cpu() for u in range(U)]).mean(0).to(dev)
    got = dense_mean[o:o + want.numel()].view(shp)
    check("multi-tensor launches: mean of the uncompressed tensor %s" % (shp,), torch.equal(got.view(torch.int32), want.view(torch.int32)))
    o += want.numel()
sys.exit(0 if ok else 1)
