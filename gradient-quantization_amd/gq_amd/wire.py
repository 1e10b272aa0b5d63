"""Wire format of one rank's compressed gradient (what RCCL moves over xGMI).

The reference never ships compressed bytes (ps_quantizer.py:41-43 decodes in place);
this layout is what makes the (codes, levels) payload real.  For one HSQ tensor of M
subvectors with k_bit <= 8 and levels that fit a byte:

    [ codes u8[M] | pad to 16 | levels u8[M] | pad to 16 | lb f32 | ub f32 | pad to 16 ]

All sections are 16-byte aligned so the kernels can write them in place (the encode
and level kernels write straight into views of this buffer; nothing is repacked).
`packed6=True`: the levels travel as four 6-bit values per three bytes (GQ_LEVELS_PACKED6, include/gq_hsq.h:
top level <= 63), 3 * ceil(M / 4) bytes instead of M -- 12.5 % off the payload of the BASELINE configuration.

`SplitHSQWire` is the same payload arranged for a two-part exchange (gq_amd/exchange.py,
"split"): bytes [0, cut) hold everything the first MA subvectors need,

    [ lb f32 | ub f32 | pad | levels[0:MA] | codes[0:MA] ]  cut  [ codes[MA:M] | pad | levels[MA:M] | pad ]

so that they can be decoded while bytes [cut, nbytes) are still in flight; the codes stay
ONE contiguous u8[M] section (the encode kernel writes them in one piece).
"""
import torch


def _up(x, a=16):
    return (x + a - 1) // a * a


class HSQWire:
    """Offsets of the sections of one rank's payload."""

    def __init__(self, M, packed6=False):
        self.M = M
        self.packed6 = packed6
        self.level_bytes = 3 * ((M + 3) // 4) if packed6 else M
        self.codes_off = 0
        self.levels_off = _up(M)
        self.lbub_off = self.levels_off + _up(self.level_bytes)
        self.nbytes = self.lbub_off + 16

    def alloc(self, device, ranks=None):
        shape = (self.nbytes,) if ranks is None else (ranks, self.nbytes)
        return torch.empty(shape, dtype=torch.uint8, device=device)

    def views(self, buf):
        """(codes u8[M], levels u8[M] -- or the packed section u8[3 * ceil(M / 4)] --, lb_ub f32[2]) views into a 1-D payload buffer."""
        assert buf.dim() == 1 and buf.numel() == self.nbytes
        codes = buf[self.codes_off:self.codes_off + self.M]
        levels = buf[self.levels_off:self.levels_off + self.level_bytes]
        lb_ub = buf[self.lbub_off:self.lbub_off + 8].view(torch.float32)
        return codes, levels, lb_ub


class SplitHSQWire:
    """The two-part arrangement (see the module docstring).  MA is a multiple of 64."""

    def __init__(self, M):
        self.M = M
        self.MA = (M // 2) // 64 * 64
        self.MB = M - self.MA
        self.lbub_off = 0
        self.levels_a_off = 16
        self.codes_off = 16 + self.MA
        self.cut = self.codes_off + self.MA
        self.levels_b_off = _up(self.codes_off + M)
        self.nbytes = _up(self.levels_b_off + self.MB)

    def views(self, buf):
        """(codes u8[M], levels u8[MA], levels u8[MB], lb_ub f32[2]) views into a 1-D payload buffer."""
        assert buf.dim() == 1 and buf.numel() == self.nbytes
        codes = buf[self.codes_off:self.codes_off + self.M]
        la = buf[self.levels_a_off:self.levels_a_off + self.MA]
        lb = buf[self.levels_b_off:self.levels_b_off + self.MB]
        lb_ub = buf[self.lbub_off:self.lbub_off + 8].view(torch.float32)
        return codes, la, lb, lb_ub
