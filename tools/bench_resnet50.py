"""BASELINE config 3 shape: ResNet-50/CIFAR parameter list (161 tensors, 23.5 M parameters),
HSQ c_dim=16 k_bit=8 n_bit=6, one record() + apply() per step on one MI355X.
Times the quantizer with the batched (segment table) kernels and with per-tensor launches."""
import json, os, sys, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd.compressors import NearestNeighborCompressor, QSGDCompressor
from gq_amd.quantizers import Quantizer
sys.path.insert(0, ROOT)
from bench import gradient_feeder      # fresh gradients under the same objects every step, set by the C++ helper

shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
n = sum(int(torch.Size(s).numel()) for s in shapes)


def run(tag, comp, users=1, steps=100, **kw):
    base = dict(c_dim=16, k_bit=8, n_bit=6, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp",
                num_users=users, mode="ps", cr=256)
    base.update(kw)
    params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
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
    for _ in range(40):      # (the graph captures of the default configuration happen in the first dozen steps; the clock settles later)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%-34s users=%d  %.3f ms/step  %.1f M elements/s (record+apply, %d tensors, wire %.2f MB/user)"
          % (tag, users, dt * 1e3, users * n / dt / 1e6, len(shapes), q.wire_bytes_per_user() / 1e6))


run("HSQ batched (segment table)", NearestNeighborCompressor)
run("HSQ batched, the reference's CPU draws", NearestNeighborCompressor, gq_rng="reference")
run("HSQ per-tensor launches", NearestNeighborCompressor, gq_no_batch=True)
run("HSQ batched, 8 simulated users", NearestNeighborCompressor, users=8, steps=5)
run("QSGD d128 n2 batched, packed wire", QSGDCompressor, c_dim=128, n_bit=2)
run("QSGD d128 n2 per-tensor launches", QSGDCompressor, c_dim=128, n_bit=2, gq_no_batch=True)
run("HSQ batched, error feedback", NearestNeighborCompressor, ef=True)
run("HSQ batched, EF + two-phase", NearestNeighborCompressor, ef=True, two_phase=True)
run("QSGD d128 n2 batched, error feedback", QSGDCompressor, c_dim=128, n_bit=2, ef=True)
run("HSQ c-dim 32 batched", NearestNeighborCompressor, c_dim=32)
run("HSQ c-dim 32 per-tensor launches", NearestNeighborCompressor, c_dim=32, gq_no_batch=True)
run("HSQ c-dim 8 batched", NearestNeighborCompressor, c_dim=8)
os.environ.setdefault("GQ_CODEBOOK_DIR", os.path.join(ROOT, "tests", "golden", "codebooks"))
run("HSQ main.py defaults (d32 n8 random) batched", NearestNeighborCompressor, c_dim=32, n_bit=8)
run("HSQ main.py defaults per-tensor launches", NearestNeighborCompressor, c_dim=32, n_bit=8, gq_no_batch=True)
run("HSQ d8 K32 batched (one row block)", NearestNeighborCompressor, c_dim=8, k_bit=5)
run("HSQ d16 K64 batched (two row blocks)", NearestNeighborCompressor, c_dim=16, k_bit=6)
run("HSQ d16 K32 batched (one row block)", NearestNeighborCompressor, c_dim=16, k_bit=5)
run("HSQ d8 K32 per-tensor launches", NearestNeighborCompressor, c_dim=8, k_bit=5, gq_no_batch=True)
run("HSQ d8 K32 batched, EF + two-phase", NearestNeighborCompressor, c_dim=8, k_bit=5, ef=True, two_phase=True)
run("TernGrad (qsgd c-dim 0 n-bit 1) batched", QSGDCompressor, c_dim=0, n_bit=1)
run("TernGrad per-tensor launches", QSGDCompressor, c_dim=0, n_bit=1, gq_no_batch=True)
run("TernGrad batched, error feedback", QSGDCompressor, c_dim=0, n_bit=1, ef=True)
