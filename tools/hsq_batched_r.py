"""HSQ (c_dim 16, k_bit 8, n_bit 6; $GQ_R_CDIM / $GQ_R_NBIT: others, e.g. main.py's defaults 32 / 8) on the ResNet-50 tensor list:
kernel times of the batched compress and of the decode-mean as a function of the number of payloads R (BASELINE configs 3-4)."""
import json, os, sys
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd.compressors import NearestNeighborCompressor
from gq_amd.quantizers import Quantizer
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_cifar_shapes.json")))["parameter_shapes"]
n = sum(int(torch.Size(s).numel()) for s in shapes)
def ev_time(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for R in ([int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]):
    args = Namespace(c_dim=int(os.environ.get("GQ_R_CDIM", "16")), k_bit=8, n_bit=int(os.environ.get("GQ_R_NBIT", "6")), no_cuda=False, random=1, ef=False, two_phase=False, scale="exp",
                     num_users=R, mode="ps", cr=256)
    params = [torch.nn.Parameter(torch.zeros(*s, device="cuda")) for s in shapes]
    q = Quantizer(NearestNeighborCompressor, params, args)
    grads = [torch.randn(p.shape, device="cuda") * 1e-3 for p in params]
    for u in range(R):
        for p, g in zip(params, grads):
            p.grad = g
        q.record(u, epoch=1)
    grp = q._groups[0][2]
    wire = q._wire[:R]
    t_dec = ev_time(lambda: grp.decode_mean(wire, R))
    gl = [params[i].grad.data for i in q._groups[0][1]]
    t_enc = ev_time(lambda: grp.encode(gl, wire[0], 0, 0))
    q.apply()
    print("R=%d: decode-mean %.1f us for %.1f M elements (%.2f TB/s of output); compress call incl. header upload and its host-side wait %.1f us (%.2f TB/s of input; kernel times: rocprofv3)"
          % (R, t_dec, n / 1e6, 4 * n / t_dec / 1e6, t_enc, 4 * n / t_enc / 1e6))
