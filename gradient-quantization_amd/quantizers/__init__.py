"""Drop-in for the reference's `quantizers` package (see compressors/__init__.py)."""
from gq_amd.quantizers import Quantizer, PSQuantizer, RingQuantizer  # noqa: F401

__all__ = ["Quantizer", "PSQuantizer", "RingQuantizer"]
