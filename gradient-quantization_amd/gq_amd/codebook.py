"""Codebook files: the reference's `.fvecs` wire format and row normalisation.

Host-side, one-off per compressor (the reference does this in NumPy too):
  * fvecs:  a little-endian int32 stream, one row = [d | d x float32]
            (reference utils/vecs_io.py:5-12)
  * rows are L2-normalised in float32, zero rows stay zero (utils/vec_np.py:4-10)
  * file name  codebooks/learned_codebook/angular_dim_{d}_Ks_{K}.fvecs, looked up
    relative to the cwd first, exactly like the reference
    (compressors/nearest_neighbor_compressor.py:50-51), then in $GQ_CODEBOOK_DIR,
    then in the copies shipped with this package (K = 256 for d = 8, 16 and 32: gq_amd/data/codebooks).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "codebooks")


def read_fvecs(path):
    raw = np.fromfile(path, dtype="<i4")
    if raw.size == 0:
        raise ValueError("empty fvecs file: %s" % path)
    d = int(raw[0])
    if d <= 0 or raw.size % (d + 1) != 0:
        raise ValueError("malformed fvecs file %s (leading dimension %d, %d words)" % (path, d, raw.size))
    rows = raw.reshape(-1, d + 1)
    if not np.all(rows[:, 0] == d):
        raise ValueError("malformed fvecs file %s: inconsistent row dimensions" % path)
    return np.ascontiguousarray(rows[:, 1:]).view(np.float32)


def normalize_rows(vecs):
    """Row-wise v / ||v||_2 in float32; rows with zero norm are returned as zeros."""
    vecs = np.asarray(vecs, dtype=np.float32)
    norms = np.sqrt(np.add.reduce(vecs * vecs, axis=1))
    col = norms[:, None]
    out = np.zeros_like(vecs)
    np.divide(vecs, col, out=out, where=col != 0)
    return norms, out


def codebook_path(dim, K):
    rel = os.path.join("learned_codebook", "angular_dim_%d_Ks_%d.fvecs" % (dim, K))
    tried = []
    for base in (os.path.join(".", "codebooks"), os.environ.get("GQ_CODEBOOK_DIR"), _DATA):
        if not base:
            continue
        p = os.path.join(base, rel)
        tried.append(p)
        if os.path.exists(p):
            return p
    raise FileNotFoundError("no codebook for dim=%d K=%d; looked in %s" % (dim, K, ", ".join(tried)))


def load_codebook(dim, K):
    """-> float32 [K, dim], row-normalised, as the reference builds `self.codewords`."""
    cb = read_fvecs(codebook_path(dim, K))
    if cb.shape != (K, dim):
        # the reference tree has a few double-written files (append-mode writer); the
        # reference would fail later on those, we fail here with a clear message
        raise ValueError("codebook %s has shape %s, expected (%d, %d)" % (codebook_path(dim, K), cb.shape, K, dim))
    return normalize_rows(cb)[1]


def repaired_dim(size, c_dim):
    """The reference's sub-dimension choice (nearest_neighbor_compressor.py:23-29,
    qsgd_compressor.py:15-22): whole tensor if c_dim == 0 or size < c_dim, else c_dim
    grown by x1.5 up to ten times until it divides `size`."""
    if c_dim == 0 or size < c_dim:
        return size
    dim = c_dim
    for _ in range(10):
        if size % dim != 0:
            dim = dim // 2 * 3
    return dim
