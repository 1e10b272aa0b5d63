"""TEST-ONLY codecs: the wire layout of the product codecs (gq_amd.quantizers.HSQCodec /
QSGDCodec) with the COMPUTE done by the CPU oracle.  They let the quantizer's host logic
(slots, wire offsets, error feedback, two-phase, the all-gather and the (rank, user) mean
order) run under `gloo` on a machine without a GPU.  Never imported by the product."""
import numpy as np
import torch

import oracle
from gq_amd.compressors import IdenticalCompressor, NearestNeighborCompressor, QSGDCompressor
from gq_amd.quantizers import DenseCodec, GenericCodec, HSQCodec, QSGDCodec


class OracleHSQCodec(HSQCodec):
    def encode_into(self, grad, wire_user, off, salt):
        c = self.c
        assert c.compressed_norm and not c.norm_compressor.random, "oracle codec: deterministic levels only"
        cb = c.codewords.cpu().numpy()
        res = oracle.hsq_compress(grad.detach().cpu().numpy().reshape(-1), cb, c.n_bit, 0)
        codes, levels, lb_ub = self._views(wire_user, off)
        codes.copy_(torch.from_numpy(res["codes"].astype(np.uint8 if self.code_dtype == torch.uint8 else np.int32)))
        levels.copy_(torch.from_numpy(res["levels"]).to(self.level_dtype))
        lb_ub.copy_(torch.tensor([res["lb"], res["ub"]], dtype=torch.float32))

    def _decode(self, gathered, off, R, out):
        c = self.c
        cb = c.codewords.cpu().numpy()
        decs = []
        for r in range(R):
            codes, levels, lb_ub = self._views(gathered[r], off)
            decs.append(oracle.hsq_decompress(codes.numpy().astype(np.int32), levels.numpy().astype(np.int32),
                                              np.float32(lb_ub[0].item()), np.float32(lb_ub[1].item()), cb, c.n_bit))
        out.copy_(torch.from_numpy(oracle.mean_users(np.stack(decs, 0))))


class OracleQSGDCodec(QSGDCodec):
    def encode_into(self, grad, wire_user, off, salt):
        c = self.c
        assert not c.random
        norm, signs, levels = oracle.qsgd_compress(grad.detach().cpu().numpy().reshape(-1), self.d, c.bit, 0)
        n, s, l = self._views(wire_user, off)
        n.copy_(torch.from_numpy(norm))
        s.copy_(torch.from_numpy(signs))
        lv = levels.copy()
        lv[lv < 0] = 0  # INT_MIN (zero bucket) -> 0 in the uint8 wire, as the HIP kernel does
        l.copy_(torch.from_numpy(lv).to(self.level_dtype))

    def _decode_rows(self, gathered, off, R, out):
        decs = []
        for r in range(R):
            n, s, l = self._views(gathered[r], off)
            decs.append(oracle.qsgd_decompress(n.numpy(), s.numpy(), l.numpy().astype(np.int32), self.d, self.c.bit))
        out.copy_(torch.from_numpy(oracle.mean_users(np.stack(decs, 0))))


def oracle_codec_factory(compressor, numel, shape):
    if isinstance(compressor, IdenticalCompressor):
        return DenseCodec(compressor, numel, shape)
    if isinstance(compressor, NearestNeighborCompressor):
        return OracleHSQCodec(compressor, numel, shape)
    if isinstance(compressor, QSGDCompressor):
        return OracleQSGDCodec(compressor, numel, shape)
    return GenericCodec(compressor, numel, shape)
