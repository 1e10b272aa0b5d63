"""Drop-in for the reference's `compressors` package: put this directory
(gradient-quantization_amd/) ahead of the reference tree on PYTHONPATH and
`from compressors import *` in main.py resolves to the MI355X implementations."""
from gq_amd.compressors import (IdenticalCompressor, QSGDCompressor, NearestNeighborCompressor,  # noqa: F401
                                ProbabilisticScalarCompressor, ProbabilisticVectorCompressor,
                                ResidualCompressor, SignSGDCompressor, TopKSparsificationCompressor,
                                MaureySparsification)

__all__ = ["IdenticalCompressor", "QSGDCompressor", "NearestNeighborCompressor", "ProbabilisticScalarCompressor",
           "ProbabilisticVectorCompressor", "ResidualCompressor", "SignSGDCompressor", "TopKSparsificationCompressor",
           "MaureySparsification"]
