"""Randomised GPU-vs-oracle checks of the whole pipeline (encode -> levels -> decode-mean) over shapes, level
widths and payload counts that the fixed tests do not enumerate.  Bit-exact throughout."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("seed", range(6))
def test_pipeline_random_shapes_match_oracle(oracle, seed):
    from gq_amd import native as nat
    rng = np.random.RandomState(500 + seed)
    dev = torch.device("cuda:0")
    for _ in range(6):
        d = int(rng.choice([4, 8, 8, 12, 16, 16, 20, 24, 32, 32, 40, 64]))
        K = int(rng.choice([16, 32, 64, 100, 256, 256, 256, 512, 1024]))
        M = int(rng.choice([1, 3, 64, 65, 500, 2049, 9000]))
        R = int(rng.choice([1, 1, 2, 3, 5]))
        n_bit = int(rng.choice([1, 2, 4, 6, 8]))
        random = int(rng.choice([0, 0, 1]))
        cb = rng.standard_normal((K, d)).astype(np.float32)
        cb /= np.maximum(np.linalg.norm(cb, axis=1, keepdims=True), 1e-20)
        cbt = torch.from_numpy(cb).to(dev)
        code_dt = torch.int32 if (K > 256 or rng.rand() < 0.3) else torch.uint8    # int32 codes are legal for any K
        top = (1 << n_bit) - (0 if random else 1)
        level_dt = torch.uint8 if top <= 255 else torch.int16
        codes = torch.empty((R, M), dtype=code_dt, device=dev)
        levels = torch.empty((R, M), dtype=level_dt, device=dev)
        lbub = torch.empty((R, 2), dtype=torch.float32, device=dev)
        decs = []
        for r in range(R):
            scale = 10.0 ** rng.uniform(-5, 2)
            x = (rng.standard_normal(M * d) * scale).astype(np.float32)
            if rng.rand() < 0.2:
                x[:] = x[:d].repeat(M).reshape(d, M).T.reshape(-1)       # one subvector repeated: lb == ub
            rr = rng.rand(M).astype(np.float32) if random else None
            ref = oracle.hsq_compress(x, cb, n_bit, random, rr)
            g = torch.from_numpy(x).to(dev)
            u = torch.empty(M, dtype=torch.float32, device=dev)
            ws = nat.new_workspace(dev, M)
            nat.hsq_encode(g, cbt, codes[r], u, ws)
            nat.hsq_levels(u, n_bit, random, torch.from_numpy(rr).to(dev) if random else None, 0, ws, lbub[r], levels[r])
            torch.cuda.synchronize()
            tag = (d, K, M, R, n_bit, random, r)
            if M > 1:   # M == 1: the reference's torch.mm takes the sgemv path (documented deviation)
                assert np.array_equal(codes[r].cpu().numpy().astype(np.int64), ref["codes"].astype(np.int64)), tag
                assert np.array_equal(_bits(u.cpu().numpy()), _bits(ref["u"])), tag
                assert np.array_equal(levels[r].cpu().numpy().astype(np.int64), ref["levels"].astype(np.int64)), tag
                lu = lbub[r].cpu().numpy()
                assert _bits(lu[0]) == _bits(ref["lb"]) and _bits(lu[1]) == _bits(ref["ub"]), tag
            # decode what the GPU produced, with the oracle's decoder
            decs.append(oracle.hsq_decompress(codes[r].cpu().numpy().astype(np.int32), levels[r].cpu().numpy().astype(np.int32),
                                              float(lbub[r, 0]), float(lbub[r, 1]), cb, n_bit))
        want = oracle.mean_users(np.stack(decs, 0))
        out = torch.empty(M * d, dtype=torch.float32, device=dev)
        nat.hsq_decode_sum(codes, levels, lbub, cbt, n_bit, out, R=R)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(want)), (d, K, M, R, n_bit, random)
