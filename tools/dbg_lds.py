import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import native
dev = torch.device("cuda:0")
d, K, M = (int(x) for x in sys.argv[1:4])
torch.manual_seed(0)
g = torch.randn(M * d, device=dev) * 0.1
cb = torch.randn(K, d, device=dev); cb = cb / cb.norm(dim=1, keepdim=True)
res = {}
for impl in (2, 5):
    codes = torch.empty(M, dtype=torch.uint8 if K <= 256 else torch.int32, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = native.new_workspace(dev, M)
    native.hsq_encode(g, cb, codes, u, ws, impl=impl)
    torch.cuda.synchronize()
    res[impl] = (codes, u)
    print("impl", impl, "ok", flush=True)
print("equal:", torch.equal(res[2][0], res[5][0]), torch.equal(res[2][1].view(torch.int32), res[5][1].view(torch.int32)))
