"""The ResNet-50 list step with the tensor list in chunks (PSQuantizer._overlap_fractions): a chunk's level + decode launch on a
second branch of the step's graph under the next chunk's encode.  One process, one box: the step time for several chunkings,
with the branches on one stream (GQ_OVERLAP_STREAMS=0: what the chunking alone costs) and on two; the decoded gradients and the
wire of every chunking are compared bit for bit with the unchunked step under deterministic rounding.
    python tools/overlap_ab.py [hsq|qsgd|hsq_ef ...]"""
import contextlib, json, os, sys, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
sys.path.insert(0, ROOT)
import torch
from gq_amd.compressors import NearestNeighborCompressor, QSGDCompressor
from gq_amd.quantizers import Quantizer
from bench import gradient_feeder

shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
dev = torch.device("cuda:0")
lists = [[torch.randn(s, device=dev) * 1e-3 for s in shapes] for _ in range(3)]
CONFIGS = {"hsq": (NearestNeighborCompressor, dict(c_dim=16, k_bit=8, n_bit=6), False),
           "hsq_d32": (NearestNeighborCompressor, dict(c_dim=32, k_bit=8, n_bit=6), False),
           "qsgd": (QSGDCompressor, dict(c_dim=128, k_bit=8, n_bit=2), False),
           "hsq_ef": (NearestNeighborCompressor, dict(c_dim=16, k_bit=8, n_bit=6), True)}
SPECS = os.environ.get("OVERLAP_SPECS", "0;0.5,0.5;0.58,0.42;0.65,0.35;0.45,0.35,0.2;0.4,0.3,0.2,0.1").split(";")


def build(Comp, kw, ef, spec, random):
    args = Namespace(no_cuda=False, random=random, ef=ef, two_phase=False, scale=0.0 if ef else "exp", num_users=1, mode="ps", cr=256,
                     gq_overlap=spec, **kw)
    params = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
    with contextlib.redirect_stdout(sys.stderr):
        q = Quantizer(Comp, params, args)
    mine = [[g.clone() for g in l] for l in lists] if ef else lists
    feed = gradient_feeder(torch, params, mine)

    def step(i):
        feed(i)
        q.record(0, epoch=1)
        q.apply()
    return q, params, step


def timed(step, warm=400, steps=400, rounds=3):
    for i in range(warm):
        step(i)
    out = []
    for r in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e6)
    return min(out), sorted(out)[len(out) // 2]


for name in (sys.argv[1:] or ["hsq", "qsgd"]):
    Comp, kw, ef = CONFIGS[name]
    # parity of the chunked step with the unchunked one (deterministic rounding: the draws of the device generator are keyed per group)
    ref = None
    for spec in SPECS:
        q, params, step = build(Comp, kw, ef, spec, 0)
        for i in range(12):
            step(i)
        torch.cuda.synchronize()
        got = [p.grad.data.clone() for p in params] + [q._wire.clone()]
        whole = sum(1 for e in q._step_graphs.values() if e[1] is not None)
        if ref is None:
            ref = got
        same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(ref, got))
        print("%-8s %-20s groups=%d whole_step_graphs=%d bit-identical to the unchunked step: %s" % (name, spec, len(q._groups), whole, same), flush=True)
        del q, params, step
    for spec in SPECS:
        for streams in (("1",) if spec == "0" else ("0", "1")):
            os.environ["GQ_OVERLAP_STREAMS"] = streams
            q, params, step = build(Comp, kw, ef, spec, 1)
            lo, med = timed(step)
            sizes = [sum(q.codecs[i].numel for i in g[1]) for g in q._groups]
            print("%-8s chunks %-20s streams=%s  step %.1f us (median %.1f)  groups %s" % (
                name, spec, "two" if streams == "1" else "one", lo, med, [round(x / 1e6, 2) for x in sizes]), flush=True)
            del q, params, step
