"""Worker for the multi-rank gloo tests (tests/test_host_logic.py): world 1..8, every exchange mode.  TEST-ONLY."""
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

SHAPES = [(96, 112), (96,), (12, 96), (12,), (40, 128)]


def make_args(users, **kw):
    base = dict(c_dim=16, k_bit=8, n_bit=6, no_cuda=True, random=0, ef=True, two_phase=False, scale="exp",
                num_users=users, mode="ps", cr=256)
    base.update(kw)
    return Namespace(**base)


def grads_for(global_user, step):
    g = torch.Generator().manual_seed(1000 * step + global_user)
    return [torch.randn(s, generator=g) * 1e-2 for s in SHAPES]


def run(quantizer, params, local_users, first_global_user, steps=2):
    out = {}
    for st in range(steps):
        for u in range(local_users):
            for p, gr in zip(params, grads_for(first_global_user + u, st)):
                p.grad = gr.clone()
            quantizer.record(u, epoch=1)
        quantizer.apply()
        for i, p in enumerate(params):
            out["s%d_p%d" % (st, i)] = p.grad.data.numpy().copy()
    return out


def build(users, mode="ps", slots=None):
    from oracle_codec import oracle_codec_factory
    from gq_amd.compressors import NearestNeighborCompressor
    from gq_amd.quantizers import Quantizer
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in SHAPES]
    q = Quantizer(NearestNeighborCompressor, params, make_args(slots or users, mode=mode), codec_factory=oracle_codec_factory)
    return q, params


def run_single_process(total_users, mode="ps"):
    q, params = build(total_users, mode)
    return run(q, params, total_users, 0)


if __name__ == "__main__":
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    mode = sys.argv[4] if len(sys.argv) > 4 else "ps"
    local = int(sys.argv[5]) if len(sys.argv) > 5 else 2          # users recorded per rank and step
    slots = int(sys.argv[6]) if len(sys.argv) > 6 else local      # args.num_users (wire slots per rank)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q, params = build(local, mode, slots)
    res = run(q, params, local, rank * local)
    if mode == "ps":
        res["exchange_mode"] = np.array(q.exchange_mode)
        res["cuts"] = np.array(len(q.cuts))
    res["wire_bytes"] = np.array(q.wire_bytes_per_user())
    os.environ["GQ_WIRE_LEVELS"] = "bytes"
    res["byte_wire_bytes"] = np.array(build(local, mode, slots)[0].wire_bytes_per_user())
    np.savez(out + "_rank%d.npz" % rank, **res)
    dist.barrier()
    dist.destroy_process_group()
