"""TEST-ONLY codecs: the wire layout of the product codecs (gq_amd.quantizers.HSQCodec /
QSGDCodec) with the COMPUTE done by the CPU oracle.  They let the quantizer's host logic
(slots, wire offsets, error feedback, two-phase, the all-gather and the (rank, user) mean
order) run under `gloo` on a machine without a GPU.  Never imported by the product."""
import numpy as np
import torch

import oracle
from gq_amd.compressors import IdenticalCompressor, NearestNeighborCompressor, QSGDCompressor
from gq_amd.quantizers import DenseCodec, GenericCodec, HSQCodec, QSGDCodec


def pack6(levels):
    """GQ_LEVELS_PACKED6 (include/gq_hsq.h): four 6-bit levels per three bytes, little-endian 24-bit groups."""
    l = np.zeros((levels.size + 3) // 4 * 4, np.uint32)
    l[:levels.size] = np.where(levels < 0, 0, levels).astype(np.uint32) & 63      # a NaN quotient's INT_MIN travels as 0, like the byte form
    w = l[0::4] | (l[1::4] << 6) | (l[2::4] << 12) | (l[3::4] << 18)
    return np.stack([w & 255, (w >> 8) & 255, (w >> 16) & 255], 1).astype(np.uint8).reshape(-1)


def unpack6(raw, M):
    b = raw.astype(np.uint32).reshape(-1, 3)
    w = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
    return np.stack([(w >> (6 * k)) & 63 for k in range(4)], 1).reshape(-1)[:M].astype(np.int32)


class OracleHSQCodec(HSQCodec):
    def encode_into(self, grad, wire_user, off, salt, r=None):
        c = self.c
        cb = c.codewords.cpu().numpy()
        if not c.compressed_norm:       # n_bit == 32: the projections travel as float32 (nearest_neighbor_compressor.py:75-78)
            codes_, u_ = oracle.hsq_encode(grad.detach().cpu().numpy().reshape(-1), cb)
            codes, norms, _ = self._views(wire_user, off)
            codes.copy_(torch.from_numpy(codes_.astype(np.uint8 if self.code_dtype == torch.uint8 else np.int32)))
            norms.copy_(torch.from_numpy(u_))
            return
        random = bool(c.norm_compressor.random)
        if random:      # reference-parity draws only: handed in by the quantizer, or drawn here as the reference does
            assert c.norm_compressor._rng == "reference", "oracle codec: deterministic levels or the reference's draws"
            r = (torch.rand(self.M) if r is None else r).cpu().numpy()
        cb = c.codewords.cpu().numpy()
        res = oracle.hsq_compress(grad.detach().cpu().numpy().reshape(-1), cb, c.n_bit, 1 if random else 0, r)
        codes, levels, lb_ub = self._views(wire_user, off)
        codes.copy_(torch.from_numpy(res["codes"].astype(np.uint8 if self.code_dtype == torch.uint8 else np.int32)))
        if self.packed6:
            levels.copy_(torch.from_numpy(pack6(res["levels"])))
        else:
            levels.copy_(torch.from_numpy(res["levels"]).to(self.level_dtype))
        lb_ub.copy_(torch.tensor([res["lb"], res["ub"]], dtype=torch.float32))

    def _decode(self, gathered, off, R, out):
        c = self.c
        cb = c.codewords.cpu().numpy()
        decs = []
        for r in range(R):
            codes, levels, lb_ub = self._views(gathered[r], off)
            if not c.compressed_norm:
                decs.append(oracle.hsq_decode(codes.numpy().astype(np.int32), levels.numpy(), cb))
                continue
            lv = unpack6(levels.numpy(), self.M) if self.packed6 else levels.numpy().astype(np.int32)
            decs.append(oracle.hsq_decompress(codes.numpy().astype(np.int32), lv,
                                              np.float32(lb_ub[0].item()), np.float32(lb_ub[1].item()), cb, c.n_bit))
        out.copy_(torch.from_numpy(decs[0] if R == 1 else oracle.mean_users(np.stack(decs, 0))))   # one payload: the plain decompress


class OracleQSGDCodec(QSGDCodec):
    """Packed QSGD wire (norm | 4- or 8-bit codes = sign<<(bits-1) | level) filled by the oracle."""

    def encode_into(self, grad, wire_user, off, salt):
        c = self.c
        if self.bits == 0:      # the plain signature form (norm | signs | levels): the reference's draws
            assert not c.random or c._rng == "reference", "oracle codec: deterministic levels or the reference's draws"
            r = torch.rand(self.Mb, self.d).numpy() if c.random else None      # qsgd_compressor.py:56
            norm, signs, levels = oracle.qsgd_compress(grad.detach().cpu().numpy().reshape(-1), self.d, c.bit,
                                                       1 if c.random else 0, r)
            vn, vs, vl = self._views(wire_user, off)
            vn.copy_(torch.from_numpy(norm))
            vs.copy_(torch.from_numpy(signs))
            if self.level_dtype == torch.uint8:
                levels = np.where(levels < 0, 0, levels)      # INT_MIN (zero bucket) -> 0 in the byte form, decodes to 0
            vl.copy_(torch.from_numpy(levels).to(self.level_dtype))
            return
        assert not c.random and self.bits in (4, 8)
        norm, signs, levels = oracle.qsgd_compress(grad.detach().cpu().numpy().reshape(-1), self.d, c.bit, 0)
        lv = levels.copy()
        lv[lv < 0] = 0  # INT_MIN (zero bucket) -> level 0 on the wire, as the HIP kernel does
        codes = (lv.astype(np.uint8) | (signs.astype(np.uint8) << (self.bits - 1))).astype(np.uint8)
        if self.bits == 4:
            codes = (codes[0::2] | (codes[1::2] << 4)).astype(np.uint8)
        wire_user[off + self.norm_off:off + self.norm_off + self.Mb * 4].view(torch.float32).copy_(torch.from_numpy(norm))
        wire_user[off + self.codes_off:off + self.codes_off + codes.size].copy_(torch.from_numpy(codes))

    def _decode_rows(self, gathered, off, R, out, plain=False):
        decs = []
        if self.bits == 0:
            for r in range(R):
                vn, vs, vl = self._views(gathered[r], off)
                decs.append(oracle.qsgd_decompress(vn.numpy(), vs.numpy(), vl.numpy().astype(np.int32), self.d, self.c.bit))
            out.copy_(torch.from_numpy(decs[0] if R == 1 else oracle.mean_users(np.stack(decs, 0))))   # one payload: the plain decompress
            return
        nb = self.numel * self.bits // 8
        for r in range(R):
            norm = gathered[r, off + self.norm_off:off + self.norm_off + self.Mb * 4].view(torch.float32).numpy()
            raw = gathered[r, off + self.codes_off:off + self.codes_off + nb].numpy()
            if self.bits == 4:
                codes = np.empty(self.numel, np.uint8)
                codes[0::2] = raw & 15
                codes[1::2] = raw >> 4
            else:
                codes = raw
            signs = codes >> (self.bits - 1)
            levels = (codes & ((1 << (self.bits - 1)) - 1)).astype(np.int32)
            decs.append(oracle.qsgd_decompress(norm, signs, levels, self.d, self.c.bit))
        out.copy_(torch.from_numpy(decs[0] if R == 1 else oracle.mean_users(np.stack(decs, 0))))   # one payload: the plain decompress


def oracle_codec_factory(compressor, numel, shape, packed6=False):
    if isinstance(compressor, IdenticalCompressor):
        return DenseCodec(compressor, numel, shape)
    if isinstance(compressor, NearestNeighborCompressor):
        return OracleHSQCodec(compressor, numel, shape, packed6)
    if isinstance(compressor, QSGDCompressor):
        return OracleQSGDCodec(compressor, numel, shape)
    return GenericCodec(compressor, numel, shape)
