"""The multi-tensor d16 encode against the flat one on the same number of elements (ResNet-50 list, 23.5 M): kernel launches
alone, HIP events around back-to-back launches.  Where does the segment-table form lose?
    python tools/batched_vs_flat.py"""
import contextlib, json, os, sys
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
dev = torch.device("cuda:0")
D_, KB_ = int(os.environ.get("GQ_AB_D", "16")), int(os.environ.get("GQ_AB_KBIT", "8"))     # other shapes: GQ_AB_D=8 GQ_AB_KBIT=5 ...
os.environ.setdefault("GQ_CODEBOOK_DIR", os.path.join(ROOT, "tests", "golden", "codebooks"))


def ev(fn, reps=200, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def one_list(shapes, label):
    args = Namespace(c_dim=D_, k_bit=KB_, n_bit=6, no_cuda=False, random=0, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256, gq_graph=False)
    params = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
    with contextlib.redirect_stdout(sys.stderr):
        q = Quantizer(NearestNeighborCompressor, params, args)
    for p in params:
        p.grad = torch.randn(p.shape, device=dev) * 1e-3
    q.record(0, epoch=1); q.apply()
    grp = q._groups[0][2]
    for p in params:
        p.grad = torch.randn(p.shape, device=dev) * 1e-3
    gl = [params[i].grad.data for i in grp.idxs]
    grp.encode(gl, q._wire[0], 0, 0)
    t = ev(lambda: grp._batch.encode(q._wire[0], None, -1))
    print("%-44s %3d tensors %6.2f M elements: multi-tensor encode %.1f us = %.2f ns / K element" % (label, len(gl), sum(g.numel() for g in gl) / 1e6, t, t * 1e3 / (sum(g.numel() for g in gl) / 1e3)))


big = [s for s in shapes if int(torch.Size(s).numel()) > 1000]
one_list([(11796480,), (11796480,)], "two equal tensors")
one_list([(310272,)] * 76, "76 equal tensors")
one_list(sorted(big, key=lambda s: -torch.Size(s).numel()), "the ResNet-50 list, largest first")
one_list([s for s in big if torch.Size(s).numel() >= 65536], "the ResNet-50 list without tensors < 64 K")
args = Namespace(c_dim=D_, k_bit=KB_, n_bit=6, no_cuda=False, random=0, ef=False, two_phase=False, scale="exp", num_users=1, mode="ps", cr=256, gq_graph=False)
params = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
with contextlib.redirect_stdout(sys.stderr):
    q = Quantizer(NearestNeighborCompressor, params, args)
grads = [torch.randn(s, device=dev) * 1e-3 for s in shapes]
for p, g in zip(params, grads):
    p.grad = g
q.record(0, epoch=1); q.apply()
grp = q._groups[0][2]
# (apply() rebinds the gradient OBJECTS to the decoded tensors: `grads` holds codeword multiples by now, on which nothing is
# ever unsettled -- fresh draws for what is timed)
grads = [torch.randn(s, device=dev) * 1e-3 for s in shapes]
for p, g in zip(params, grads):
    p.grad = g
gl = [params[i].grad.data for i in grp.idxs]
n = sum(g.numel() for g in gl)
grp.encode(gl, q._wire[0], 0, 0)
t_enc = ev(lambda: grp._batch.encode(q._wire[0], None, -1))
t_lv = ev(lambda: grp._batch.levels(q._wire[0], native.RANDOM_OFF, 0, None))
flat = torch.cat([g.reshape(-1) for g in gl])
M = flat.numel() // D_
codes, u, ws = torch.empty(M, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.float32, device=dev), native.new_workspace(dev, M)
cb = grp.codebook
t_flat = ev(lambda: native.hsq_encode(flat, cb, codes, u, ws))
big = [g for g in gl if g.numel() >= 262144]
print("%d tensors, %.2f M elements: multi-tensor encode %.1f us, levels %.1f us; the same elements as ONE flat tensor %.1f us"
      % (len(gl), n / 1e6, t_enc, t_lv, t_flat))
