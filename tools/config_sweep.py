"""Record+apply time of the ResNet-50 parameter list over the reference's flag space (one MI355X):
looks for configurations that fall off the multi-tensor kernels."""
import itertools, json, os, sys, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
os.environ.setdefault("GQ_CODEBOOK_DIR", os.path.join(ROOT, "tests", "golden", "codebooks"))
import torch
from gq_amd.compressors import NearestNeighborCompressor, QSGDCompressor
from gq_amd.quantizers import Quantizer
sys.path.insert(0, ROOT)
from bench import gradient_feeder      # fresh gradients under the same objects every step, set by the C++ helper

shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]


def run(comp, users=1, steps=60, **kw):
    base = dict(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp",
                num_users=users, mode="ps", cr=256)
    base.update(kw)
    params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        q = Quantizer(comp, params, Namespace(**base))
    # apply() rebinds the gradient OBJECTS to the decoded tensors: re-using one list would time the codec on its own output
    # (codeword multiples: nothing is ever unsettled) from the second step on
    lists = [[torch.randn(p.shape, device="cuda") * 1e-3 for p in params] for _ in range(3)]
    feed = gradient_feeder(torch, params, lists)
    tick = [0]

    def step():
        for u in range(users):
            feed(tick[0])
            tick[0] += 1
            q.record(u, epoch=1)
        q.apply()
    for _ in range(45):      # (graph replay is the default: its captures happen in the first dozen steps PER address set, three sets)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, sum(len(g[1]) for g in q._groups)


rows = []
for c_dim, k_bit in ((16, 8), (32, 8), (8, 8), (8, 5)):
    for n_bit, random in ((6, 0), (6, 1), (8, 1), (4, 1), (32, 1), (12, 0)):
        for extra in ({}, {"ef": True, "two_phase": True}, {"mode": "ring"}):
            kw = dict(c_dim=c_dim, k_bit=k_bit, n_bit=n_bit, random=random, **extra)
            try:
                ms, nb = run(NearestNeighborCompressor, **kw)
            except Exception as e:   # noqa
                ms, nb = float("nan"), repr(e)[:80]
            rows.append(("hsq", kw, ms, nb))
for c_dim in (0, 128, 512, 2048):
    for n_bit, random in ((1, 1), (2, 1), (2, 0), (4, 1), (8, 1)):
        for extra in ({}, {"ef": True}, {"mode": "ring"}):
            kw = dict(c_dim=c_dim, n_bit=n_bit, random=random, **extra)
            try:
                ms, nb = run(QSGDCompressor, **kw)
            except Exception as e:   # noqa
                ms, nb = float("nan"), repr(e)[:80]
            rows.append(("qsgd", kw, ms, nb))
for name, kw, ms, nb in sorted(rows, key=lambda r: -r[2] if r[2] == r[2] else -1e9):
    print("%-5s %8.3f ms  batched tensors %-4s %s" % (name, ms, nb, kw))
