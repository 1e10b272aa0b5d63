"""Randomised check of gq_hsq_levels_decode (level quantiser + decode of one payload in one launch) against gq_hsq_levels
followed by gq_hsq_decode_sum, and of the decode-mean kernels for random R against the sum of R single decodes' inputs:
random M (ragged, tiny, multi-item), n_bit, random mode, byte / packed levels, gradient scales incl. zeros / NaN / inf.
    python tools/fuzz_fused.py [seconds] [seed]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import numpy as np
import torch
from gq_amd import native as nat
from gq_amd.codebook import load_codebook

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
t0, rounds, bad, means = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    rounds += 1
    M = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 257, 4099, 70_001, 524_288, 1_200_003])) + int(rng.integers(0, 4))
    packed6 = bool(rng.random() < 0.5)
    random = int(rng.choice([0, 0, 1, 2]))
    n_bit = int(rng.integers(1, 6 if (packed6 and random) else (7 if packed6 else (8 if random else 9))))
    K = int(rng.choice([256, 256, 256, 64, 17]))
    cbk = cb[:K].contiguous()
    x = torch.randn(M * 16, device=dev) * float(10.0 ** rng.integers(-6, 3))
    sp = rng.random()
    if sp < 0.05:
        x.zero_()
    elif sp < 0.10:
        x[int(rng.integers(0, M * 16))] = float("nan")
    elif sp < 0.15:
        x[int(rng.integers(0, M * 16))] = float("inf")
    elif sp < 0.20:
        x[: 16 * (M // 2)] = 0
    codes = torch.empty(M, dtype=torch.uint8, device=dev)
    u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = nat.new_workspace(dev, M)
    nat.hsq_encode(x, cbk, codes, u, ws)
    r = torch.rand(M, device=dev) if random == 1 else None
    seed = int(rng.integers(0, 2 ** 62))
    nlev = nat.packed6_bytes(M) + 4 if packed6 else M
    res = []
    for fused in (False, True):
        lb_ub = torch.zeros(2, dtype=torch.float32, device=dev)
        levels = torch.zeros(nlev, dtype=torch.uint8, device=dev)
        out = torch.full((M * 16,), 7.0, dtype=torch.float32, device=dev)
        if fused:
            assert nat.hsq_levels_decode(u, n_bit, random, r, seed, ws, lb_ub, levels, codes, cbk, out, packed6)
        else:
            nat.hsq_levels(u, n_bit, random, r, seed, ws, lb_ub, levels, packed6)
            if packed6:
                rc = nat.lib().gq_hsq_decode_sum_strided(
                    ctypes.c_void_p(codes.data_ptr()), 1, ctypes.c_int64(M), ctypes.c_void_p(levels.data_ptr()), nat.LEVELS_PACKED6,
                    ctypes.c_int64(levels.numel()), ctypes.c_void_p(lb_ub.data_ptr()), ctypes.c_int64(8),
                    ctypes.c_void_p(cbk.data_ptr()), 1, ctypes.c_int64(M), 16, K, n_bit, ctypes.c_void_p(out.data_ptr()), None)
                assert rc == 0
            else:
                nat.hsq_decode_sum(codes, levels, lb_ub, cbk, n_bit, out, R=1)
        torch.cuda.synchronize()
        res.append((lb_ub, levels, out))
    ok = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(res[0], res[1]))
    if not ok:
        bad += 1
        print("MISMATCH fused", dict(M=M, packed6=packed6, random=random, n_bit=n_bit, K=K, sp=sp))
    # decode-mean over R copies-with-different-levels: R payloads built by re-quantising u with different seeds
    if not packed6 and rounds % 3 == 0 and M <= 70_010:
        means += 1
        R = int(rng.choice([2, 3, 4, 7, 8, 9, 12, 16, 17, 20]))
        cs, ls, bs = [], [], []
        for k in range(R):
            lbk = torch.zeros(2, dtype=torch.float32, device=dev)
            lvk = torch.zeros(M, dtype=torch.uint8, device=dev)
            nat.hsq_levels(u, min(n_bit, 7), 2, None, seed + k, ws, lbk, lvk)
            cs.append(codes); ls.append(lvk); bs.append(lbk)
        outR = torch.empty(M * 16, dtype=torch.float32, device=dev)
        nat.hsq_decode_sum(torch.stack(cs).view(-1), torch.stack(ls).view(-1), torch.stack(bs).view(-1), cbk, min(n_bit, 7), outR, R=R)
        acc = torch.zeros(M * 16, dtype=torch.float32, device=dev)
        one = torch.empty(M * 16, dtype=torch.float32, device=dev)
        for k in range(R):      # the reference's sum: (+0 + d_0 + d_1 + ...) in payload order, then / R
            nat.hsq_decode_sum(cs[k], ls[k], bs[k], cbk, min(n_bit, 7), one, R=1)
            acc = acc + one
        want = (acc.cpu() / R).to(dev)      # the CPU's true division (torch's GPU division by a scalar multiplies by 1 / R)
        torch.cuda.synchronize()
        same = torch.equal(outR.view(torch.int32), want.view(torch.int32)) or bool(((outR == want) | (outR.isnan() & want.isnan())).all())
        if not same:
            bad += 1
            print("MISMATCH mean", dict(M=M, R=R, n_bit=n_bit, K=K, sp=sp))
print("fuzz_fused: %d rounds (%d with a decode-mean check), %d mismatches" % (rounds, means, bad))
