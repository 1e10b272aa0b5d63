"""Worker for the two-phase seed test (tests/test_host_logic.py): two gloo ranks that seed torch DIFFERENTLY run --two-phase with a
compressor whose stochastic rounding takes its seed from gq_amd.compressors._next_seed().  TEST-ONLY."""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "gradient-quantization_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

SHAPES = [(40, 64), (12,), (1500,)]


class SeededRounding(object):
    """A user-supplied compressor (it reaches the wire through GenericCodec): rounds to a grid of 1e-3 stochastically, the
    draws from a generator seeded by the package's _next_seed() -- the call every in-package fallback path makes."""

    def __init__(self, size, shape, args):
        self.shape = shape

    def compress(self, vec):
        from gq_amd.compressors import _next_seed
        g = torch.Generator().manual_seed(_next_seed() & (2 ** 62 - 1))
        x = vec.reshape(-1) * 1e3
        lo = torch.floor(x)
        return lo + (torch.rand(x.numel(), generator=g) < (x - lo)).float()

    def decompress(self, sig):
        return (sig * 1e-3).view(self.shape)


if __name__ == "__main__":
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + 17 * rank)      # per-rank seeds, as per-rank data augmentation makes them
    from gq_amd.quantizers import Quantizer
    args = Namespace(c_dim=16, k_bit=8, n_bit=6, no_cuda=True, random=1, ef=sys.argv[4] == "1", two_phase=True, scale="exp",
                     num_users=1, mode="ps", cr=256)
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in SHAPES]
    q = Quantizer(SeededRounding, params, args)
    res = {}
    for st in range(3):
        gen = torch.Generator().manual_seed(1000 * st + rank)
        for p in params:
            p.grad = torch.randn(p.shape, generator=gen) * 1e-2
        q.record(0, epoch=1)
        q.apply()
        for i, p in enumerate(params):
            res["s%d_p%d" % (st, i)] = p.grad.data.numpy().copy()
            if args.ef:
                res["s%d_e%d" % (st, i)] = p.server_error.numpy().copy()
    np.savez(out + "_rank%d.npz" % rank, **res)
    dist.barrier()
    dist.destroy_process_group()
