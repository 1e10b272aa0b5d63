"""The one-rank list step as ONE replayed HIP graph (encode; levels + decode + tail) against the SAME two launches issued directly on
the stream, back to back: does the graph's boundary between two replays (3.9 us, profiles/r06_overlap_ab.txt B) go away?
    python tools/direct_vs_graph.py"""
import contextlib, ctypes, json, os, sys, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd")); sys.path.insert(0, ROOT)
import torch
from gq_amd import native
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
dev = torch.device("cuda:0")
args = Namespace(no_cuda=False, random=1, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256, c_dim=16, k_bit=8, n_bit=6)
params = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
with contextlib.redirect_stdout(sys.stderr):
    q = Quantizer(NearestNeighborCompressor, params, args)
grads = [torch.randn(s, device=dev) * 1e-3 for s in shapes]
keep = [g.clone() for g in grads]
for p, g in zip(params, grads):
    p.grad = g.view(g.shape)
objs = [p.grad for p in params]
for i in range(30):
    for o, g, k in zip(objs, grads, keep):
        o.data = g
    q.record(0, epoch=1); q.apply()
torch.cuda.synchronize()
fents = [e for e in q._step_graphs.values() if e[1] is not None]
print("whole-step graphs:", len(fents), q.graph_counts())
graph = fents[0][1]


def ev(fn, n=3000, warm=500):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(n): fn()
    e.record(); t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3, t_issue / n * 1e6

print("graph replay:        %.2f us per step on the device (host issues one in %.1f us)" % ev(graph.replay))
# the same two launches, directly: the group's descriptor on the exact graph's own device header, the tail the capture built
grp = q._groups[0][2]
hdr = [e for k, e in q._rec_graphs.items() if e[1] is not None and k[0] != "any"][0][2][0]
grp._batch.set_table(hdr[:grp._table_words])
grp._batch.set_dense(hdr[grp._dense_at:].view(grp.ndense, 3), grp.ndense)
L = native.lib()
wire = q._wire[0]
out = grp._outs[0]
dense_rows = q._wire[:1][:, q.dense_off:q.dense_off + q.dense_bytes].view(torch.float32)
tail = native.StepTail(rows=dense_rows, out=q._dense_mean[0], rng_state=q._rng_state, reset=(grp._dev[grp._table_words:grp._dense_at], grp._acc_init), ticket=q._ticket_for(dev, 0))
seed = grp._counter_seed(0)
st = native._stream()
wp, op = ctypes.c_void_p(wire.data_ptr()), ctypes.c_void_p(out.data_ptr())
nan = ctypes.c_float(float("nan"))
mode, seed_c, zero, null = ctypes.c_int(native.RANDOM_DEVICE_COUNTER), ctypes.c_uint64(seed), ctypes.c_int(0), ctypes.c_void_p(0)
enc, ld, bref, tref = L.gq_hsq_encode_batched, L.gq_hsq_levels_decode_batched, grp._batch.ref, tail.ref
grp.ensure_clean()

def direct():
    rc1 = enc(bref, wp, nan, st)
    rc2 = ld(bref, wp, mode, seed_c, null, zero, op, zero, tref, st)
    assert rc1 == 0 and rc2 == 0
print("two direct launches: %.2f us per step on the device (host issues one in %.1f us)" % ev(direct))
print("graph replay again:  %.2f us per step on the device (host issues one in %.1f us)" % ev(graph.replay))
