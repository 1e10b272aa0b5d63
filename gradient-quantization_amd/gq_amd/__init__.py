"""gq_amd -- MI355X-native gradient vector quantisation (HSQ / QSGD hot path).

The compute lives in libgq_hsq.so (hand-written HIP for gfx950, C ABI in
include/gq_hsq.h); this package is the host-side mirror of the reference's
Compressor / Quantizer protocol.
"""
from . import native  # noqa: F401
