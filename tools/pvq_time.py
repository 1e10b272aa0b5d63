"""ProbabilisticVectorCompressor / ResidualCompressor compress times on a 25 M-element gradient."""
import os, sys, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd.compressors import ProbabilisticVectorCompressor, ResidualCompressor, NearestNeighborCompressor
x = torch.randn(25_000_000, device="cuda")
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for k_bit in (8, 0):
    a = Namespace(c_dim=16, k_bit=k_bit, n_bit=6, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256)
    for cls in (ProbabilisticVectorCompressor, ResidualCompressor, NearestNeighborCompressor):
        try:
            c = cls(x.numel(), x.shape, a)
            print("%-32s k_bit=%d: compress %.3f ms, roundtrip %.3f ms" % (cls.__name__, k_bit, t(lambda: c.compress(x)), t(lambda: c.decompress(c.compress(x)))))
        except Exception as e:
            print(cls.__name__, k_bit, "->", type(e).__name__, str(e)[:100])
