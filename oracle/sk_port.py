"""The reference's distance step on its own third-party kernel.  TEST INFRASTRUCTURE ONLY.

Only tests/ and bench.py's cpu_baseline leg may import this module (same rule as ref_port.py / bfk_oracle.c).

The arithmetic of the reference's hot path is not in the reference's repository: get_neighbours_batch
(src/breakfast/breakfast.py:223-276) hands a scipy CSR count matrix to scikit-learn's
`pairwise_distances_chunked(metric="manhattan", n_jobs=1)` (uv.lock pins scikit-learn 1.9.0; this image has
1.7.2 — `_sparse_manhattan`, sklearn/metrics/_pairwise_fast.pyx, is unchanged between them) and thresholds each
row of the dense chunk with `flatnonzero(d <= max_dist)` (:226-228).  This file restates exactly that call
pattern on a CSR given as (indptr, indices) — the band of rows whose length is within max_dist of one query length
at a time, batch rows x band rows (:249-253, called per distinct length by cluster_features :314-318) — so that

  * tests/test_oracle.py can pin the C restatement (bfk_oracle.c) against the very kernel the reference calls, and
  * bench.py can time that kernel on the GPU box's host cores beside the GPU number.

If scikit-learn / scipy are missing, `available()` is False and both users skip it.
"""

from __future__ import annotations

import time

import numpy as np


def available() -> bool:
    try:
        import scipy.sparse  # noqa: F401
        import sklearn.metrics  # noqa: F401
    except Exception:
        return False
    return True


def versions() -> str:
    import scipy
    import sklearn

    return f"scikit-learn {sklearn.__version__}, scipy {scipy.__version__}"


def count_matrix(indptr, indices, n_vocab=None):
    """CSR of token counts, repeats summed (what sparse_feature_matrix builds, :212-215)."""
    from scipy.sparse import csr_matrix

    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    width = int(n_vocab) if n_vocab else (int(indices.max()) + 1 if indices.size else 1)
    m = csr_matrix((np.ones(indices.size, dtype=np.int64), indices, indptr), shape=(len(indptr) - 1, width))
    m.sum_duplicates()
    return m


def neighbours(indptr, indices, max_dist, select_ind=None, n_vocab=None):
    """-> (lists, seconds in the sklearn calls): lists[i] = sorted columns within max_dist of query row i, self
    included; the query rows are select_ind (or all rows).  One pairwise_distances_chunked call per distinct query
    length, on the length band of that length, like the reference."""
    from sklearn.metrics import pairwise_distances_chunked

    x = count_matrix(indptr, indices, n_vocab)
    lengths = np.diff(np.asarray(indptr, dtype=np.int64))
    queries = np.arange(len(lengths)) if select_ind is None else np.asarray(select_ind, dtype=np.int64)
    found = [set() for _ in queries]
    spent = 0.0
    for q in dict.fromkeys(lengths[queries].tolist()):
        in_band = np.abs(lengths - q) <= max_dist
        band_cols = np.flatnonzero(in_band)
        band_queries = np.flatnonzero(in_band[queries])
        t0 = time.perf_counter()
        chunks = pairwise_distances_chunked(
            X=x[queries[band_queries], :], Y=x[band_cols, :], metric="manhattan", n_jobs=1,
            reduce_func=lambda dist, start: [np.flatnonzero(row <= max_dist) for row in dist])
        hits = [h for chunk in chunks for h in chunk]
        spent += time.perf_counter() - t0
        for qi, h in zip(band_queries, hits):
            found[qi].update(band_cols[h].tolist())
    return [np.array(sorted(s), dtype=np.int64) for s in found], spent


def merges(indptr, max_dist, select_ind=None) -> int:
    """row pairs the calls above merge (every query row meets its band once per query length in its band)"""
    lengths = np.diff(np.asarray(indptr, dtype=np.int64))
    queries = np.arange(len(lengths)) if select_ind is None else np.asarray(select_ind, dtype=np.int64)
    total = 0
    for q in dict.fromkeys(lengths[queries].tolist()):
        in_band = np.abs(lengths - q) <= max_dist
        total += int(in_band[queries].sum()) * int(in_band.sum())
    return total
