"""INTEGRATION.md section 3, runnable: the ctypes stub a maintainer would put into the reference's
NearestNeighborCompressor.compress, checked against this package's own class."""
import ctypes, torch, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
from gq_amd.codebook import load_codebook
_gq = ctypes.CDLL(os.path.join(ROOT, "gradient-quantization_amd", "libgq_hsq.so"))
_gq.gq_last_error.restype = ctypes.c_char_p
P = lambda t: ctypes.c_void_p(t.data_ptr())
class C: pass
self = C(); self.dim = 16; self.K = 256; self.code_dtype = torch.uint8; self.compressed_norm = True; self.n_bit = 6
self.codewords = torch.from_numpy(load_codebook(16, 256)).cuda()
def compress(self, vec):                                   # drop-in body
    vec = vec.contiguous().view(-1)
    M = vec.numel() // self.dim
    codes = torch.empty(M, dtype=self.code_dtype, device=vec.device)
    u = torch.empty(M, dtype=torch.float32, device=vec.device)
    _gq.gq_hsq_workspace_bytes.restype = ctypes.c_size_t                   # encode workspace, zeroed once
    part = torch.zeros(_gq.gq_hsq_workspace_bytes(ctypes.c_int64(M)) // 4 + 1, dtype=torch.float32, device=vec.device)
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
print("codes equal:", torch.equal(codes, ref[1]), "levels equal:", torch.equal(l, ref[0][2].to(torch.int32)), "lb/ub:", float(lb) == float(ref[0][0]), float(ub) == float(ref[0][1]))
