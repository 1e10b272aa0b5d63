"""The ResNet-50 list step piece by piece for ONE library build (A/B through tools/ab_script.py):
QSGD compress launch with and without the dense tensors riding in it, HSQ level launch likewise, and the whole
record + apply step eager and replayed from graphs, for both codecs.
    python tools/ab_script.py tools/list_step_ab.py product tools/exp/libgq_X.so"""
import contextlib, json, os, sys, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd.compressors import NearestNeighborCompressor, QSGDCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
dev = torch.device("cuda:0")


def ev_time(fn, reps=60):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for name, Comp, kw in (("hsq ", NearestNeighborCompressor, dict(c_dim=16, k_bit=8, n_bit=6)), ("qsgd", QSGDCompressor, dict(c_dim=128, k_bit=8, n_bit=2))):
    res = {}
    for graph in (False, True):
        args = Namespace(no_cuda=False, random=1, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256, gq_graph=graph, **kw)
        params = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
        with contextlib.redirect_stdout(sys.stderr):
            q = Quantizer(Comp, params, args)
        lists = [[torch.randn(s, device=dev) * 1e-3 for s in shapes] for _ in range(3)]
        fresh = [[g.view(g.shape) for g in lists[i % 3]] for i in range(260)]
        def step(i):
            for p, g in zip(params, fresh[i]):
                p.grad = g
            q.record(0, epoch=1); q.apply()
        for i in range(60): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(60, 260): step(i)
        torch.cuda.synchronize()
        res["graph" if graph else "eager"] = (time.perf_counter() - t0) / 200 * 1e6
        if not graph:
            grp = q._groups[0][2]
            for p, g in zip(params, lists[0]):
                p.grad = g.view(g.shape)
            gl = [params[i].grad.data for i in grp.idxs]
            dn = [params[i].grad.data for i in q.dense_idx]
            res["compress"] = ev_time(lambda: grp.encode(gl, q._wire[0], 0, 0))
            res["compress+dense"] = ev_time(lambda: grp.encode(gl, q._wire[0], 0, 0, dense=dn))
    print("%s: compress launches %.1f us, with the dense tensors riding %.1f us; step eager %.1f us, graph %.1f us"
          % (name, res["compress"], res["compress+dense"], res["eager"], res["graph"]))
