"""Worker for the world_size-2 GPU test (tests/test_gpu_api.py): both ranks drive the HIP kernels on cuda:0 and
exchange the wire over gloo (two ranks cannot share one GPU under RCCL).  TEST-ONLY."""
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

SHAPES = [(96, 112), (96,), (64, 64, 3, 3), (12,), (40, 128), (1024,)]


def make_args(users, **kw):
    base = dict(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=0, ef=False, two_phase=False, scale="exp",
                num_users=users, mode="ps", cr=256)
    base.update(kw)
    return Namespace(**base)


def grads_for(global_user, step):
    g = torch.Generator().manual_seed(1000 * step + global_user)
    return [torch.randn(s, generator=g) * 1e-2 for s in SHAPES]


def run(quantizer, params, local_users, first_global_user, steps=None):
    steps = int(os.environ.get("GQ_TEST_STEPS", "2")) if steps is None else steps
    out = {}
    for st in range(steps):
        for u in range(local_users):
            for p, gr in zip(params, grads_for(first_global_user + u, st)):
                p.grad = gr.cuda()
            quantizer.record(u, epoch=1)
        quantizer.apply()
        for i, p in enumerate(params):
            out["s%d_p%d" % (st, i)] = p.grad.data.cpu().numpy().copy()
    return out


def build(users, mode, quant, **kw):
    from gq_amd.compressors import NearestNeighborCompressor, QSGDCompressor
    from gq_amd.quantizers import Quantizer
    params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in SHAPES]
    if quant == "qsgd":
        kw.update(c_dim=128, n_bit=2)
    if quant == "terngrad":     # one bucket per tensor: wide and narrow tensors end up in single-tensor groups
        kw.update(c_dim=0, n_bit=1, gq_no_batch=True)
    comp = NearestNeighborCompressor if quant == "hsq" else QSGDCompressor
    return Quantizer(comp, params, make_args(users, mode=mode, **kw)), params


def run_single_process(total_users, mode, quant, **kw):
    q, params = build(total_users, mode, quant, **kw)
    return run(q, params, total_users, 0)


if __name__ == "__main__":
    rank, world, out, mode, quant, ef = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6] == "1"
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = int(sys.argv[7]) if len(sys.argv) > 7 else 2
    extra = {}      # argv[8]: more Namespace fields, "two_phase=1,random=1"
    for kv in (sys.argv[8].split(",") if len(sys.argv) > 8 and sys.argv[8] else []):
        k, v = kv.split("=")
        extra[k] = int(v)
    torch.manual_seed(1234 + 31 * rank)     # per-rank seeds: the replicated second phase (ps_quantizer.py:52-61) must not depend on them
    q, params = build(local, mode, quant, ef=ef, **extra)
    res = run(q, params, local, rank * local)
    np.savez(out + "_rank%d.npz" % rank, **res)
    with open(out + "_rank%d_graphs.txt" % rank, "w") as f:      # (gq_graph: how many records / applies were captured)
        f.write("%d %d" % (sum(1 for e in q._rec_graphs.values() if e[1] is not None),
                           sum(1 for e in q._apply_graphs.values() if e[1] is not None)))
    dist.barrier()
    dist.destroy_process_group()
