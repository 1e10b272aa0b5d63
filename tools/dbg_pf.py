import sys, os, numpy as np, torch
sys.path.insert(0, 'gradient-quantization_amd'); sys.path.insert(0, '.')
from gq_amd import native
g = np.load('tests/golden/hsq_heavytail_det.npz')
cbn = np.load('tests/golden/codebook_d16_k256_normalized.npy')
dev = torch.device('cuda:0')
x = torch.from_numpy(g['x'].reshape(-1)).to(dev); cb = torch.from_numpy(cbn).to(dev)
M = x.numel() // 16
out = {}
for impl in (1, 4):
    codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = native.new_workspace(dev, M); native.mark_worklist(ws, M)
    native.hsq_encode(x, cb, codes, u, ws, impl=impl); torch.cuda.synchronize()
    out[impl] = (codes.cpu().numpy(), u.cpu().numpy(), native.fixup_count(ws, M))
bad = np.nonzero(out[1][0] != out[4][0])[0]
print('fixups', out[4][2], 'n bad', len(bad))
for m in bad[:6]:
    v = g['x'].reshape(-1, 16)[m].astype(np.float64)
    p = cbn.astype(np.float64) @ v
    o = np.argsort(-np.abs(p))[:4]
    print(m, 'exact', out[1][0][m], out[1][1][m], 'pf', out[4][0][m], out[4][1][m], 'top4', o, np.abs(p[o]), 'vmax', np.abs(v).max())
