"""What each launch of a ResNet-50 list step costs INSIDE a replayed HIP graph (HSQ d16, one rank): graphs of subsets of the
step's launches, replayed back to back.  Tells what removing the header copy / the dense tensors' mean launch would buy.
    python tools/graph_pieces.py"""
import contextlib, json, os, sys
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
import gc
gc.disable()      # (a CUDAGraph finalized by the collector INSIDE a stream capture takes the process down: gq_amd.quantizers._capturing)
from gq_amd import native
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
dev = torch.device("cuda:0")
args = Namespace(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256, gq_graph=False)
params = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
with contextlib.redirect_stdout(sys.stderr):
    q = Quantizer(NearestNeighborCompressor, params, args)
grads = [torch.randn(s, device=dev) * 1e-3 for s in shapes]
for p, g in zip(params, grads):
    p.grad = g.view(g.shape)
q.record(0, epoch=1); q.apply()
for p, g in zip(params, grads):
    p.grad = g.view(g.shape)
grp = q._groups[0][2]
gl = [params[i].grad.data for i in grp.idxs]
dn = [params[i].grad.data for i in q.dense_idx]
wire = q._wire[0]
grp.encode(gl, wire, 0, 0, dense=dn)
hdr = grp._host[grp._last_slot].clone().pin_memory()
seed = grp._counter_seed(0)
mode = native.RANDOM_DEVICE_COUNTER if seed is not None else native.RANDOM_OFF
rows = q._wire[:1][:, q.dense_off:q.dense_off + q.dense_bytes].view(torch.float32)
dmean = torch.empty(rows.shape[1], dtype=torch.float32, device=dev)
out, _ = grp._out_buffer(dev, advance=False)


hdr_dev = hdr.to(dev)
def copy(): grp._dev.copy_(hdr, non_blocking=True)
def copy_dd(): grp._dev.copy_(hdr_dev, non_blocking=True)
def copy_k(): torch.bitwise_or(hdr_dev, 0, out=grp._dev)
def reset(): grp._acc_clean = False; grp.ensure_clean()
def enc(): grp._batch.encode(wire, None, -1)
def lev(): grp._batch.levels(wire, mode, seed or 0, None)
def dec(): grp._batch.decode(q._wire[:1], 1, out)
def mean(): native.mean_rows(rows, dmean, rng_state=q._rng_state)


def graph_of(fns):
    g = torch.cuda.CUDAGraph()
    for f in fns: f()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for f in fns: f()
    return g


def timed(graphs, reps=300):
    for _ in range(30):
        for g in graphs: g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for g in graphs: g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


cases = [("record [copy, encode, levels] + apply [decode, mean]  (header from pinned host memory)", [[copy, enc, lev], [dec, mean]]),
         ("record [device-to-device copy, encode, levels] + apply [decode, mean]", [[copy_dd, enc, lev], [dec, mean]]),
         ("[device-to-device copy]", [[copy_dd]]),
         ("record [copy by an elementwise kernel, encode, levels] + apply [decode, mean]", [[copy_k, enc, lev], [dec, mean]]),
         ("record [encode, levels, accumulator reset kernel] + apply [decode, mean]  (the library: a graph reads its own tables)", [[enc, lev, reset], [dec, mean]]),
         ("one graph [encode, levels, reset, decode, mean]  (the library, one rank and one user)", [[enc, lev, reset, dec, mean]]),
         ("one graph [copy, encode, levels, decode, mean]", [[copy, enc, lev, dec, mean]]),
         ("no header copy: [encode, levels] + [decode, mean]", [[enc, lev], [dec, mean]]),
         ("no mean launch: [copy, encode, levels] + [decode]", [[copy, enc, lev], [dec]]),
         ("neither: [encode, levels] + [decode]", [[enc, lev], [dec]]),
         ("neither, one graph [encode, levels, decode]", [[enc, lev, dec]]),
         ("[encode]", [[enc]]), ("[levels]", [[lev]]), ("[decode]", [[dec]]), ("[mean]", [[mean]]), ("[copy]", [[copy]])]
for name, gs in cases:
    graphs = [graph_of(f) for f in gs]
    print("%-70s %6.1f us per step" % (name, timed(graphs)))
