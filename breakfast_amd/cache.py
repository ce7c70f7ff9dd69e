"""Incremental neighbour cache — counterpart of the reference's src/breakfast/cache.py.

Same file format as the reference (gzip + pickle protocol 2 of ``{"max_dist", "version", "neigh", "meta"}``
with ``neigh`` = list of index arrays and ``meta`` = DataFrame[id, feature]), so caches written by either
tool can be read by the other.  What is cached are the neighbour lists of ``get_neighbours_batch``
(breakfast.py:223-276); a later run re-indexes them onto the new input (rows are matched by their feature
string), computes lists only for rows that are new (query = new rows x all columns, the ``select_ind`` shape
of breakfast.py:241-245 / :300-304) and recovers components from the union of old and new lists.

  load / validate          cache.py:9-15, :35-48   (max_dist mismatch -> full recompute, version mismatch warns)
  save                     cache.py:18-32
  map_features / find_new  cache.py:84-112
  update_neighbours        cache.py:51-71

The distance work runs on the GPU (bfk_neighbours_csr), and so does the lists -> components step
(bfk_labels_from_lists); there is no scipy/sklearn/networkx path here.
"""

from __future__ import annotations

import gzip
import pickle

import numpy as np

from . import __version__, _lib


class CacheMismatch(Exception):
    """the cache cannot be used (other max_dist): fall back to a full computation, like the reference"""


def load(input_file, max_dist):
    with gzip.open(input_file, "rb") as f:
        print("Import from pickle file")
        cache = pickle.load(f)
    validate(cache, max_dist, __version__)
    return cache


def validate(cache, max_dist, version):
    cached = cache["max_dist"]
    if max_dist != cached:
        print("WARNING: Cached results were created using a differnt max-dist paramter")
        print(f"Current max-dist parameter: {max_dist}")
        print(f"Cached max-dist parameter: {cached}")
        raise CacheMismatch()
    if cache["version"] != version:
        print(f"WARNING: Cached results were created using breakfast version {cache['version']}")


def save(output_file, neigh, meta, max_dist):
    try:
        print("Export results as pickle")
        d = {"max_dist": max_dist, "version": __version__, "neigh": neigh, "meta": meta[["id", "feature"]]}
        output_file.parent.mkdir(parents=True, exist_ok=True)
        with gzip.open(output_file, "wb") as f:
            pickle.dump(d, f, 2)
    except TypeError:
        print("Export of pickle was not succesfull")


def map_features(cached_feats, new_feats):
    """-> (c2n: cached row -> new row or -1, new_rows: rows of the new input absent from the cache, in the
    order the reference visits them (its outer join sorts by feature string, cache.py:94-112))."""
    new_feats = list(new_feats)
    where = {f: i for i, f in enumerate(new_feats)}
    cached_feats = list(cached_feats)
    c2n = np.array([where.get(f, -1) for f in cached_feats], dtype=np.int64)
    cached_set = set(cached_feats)
    new_rows = [i for f, i in sorted(where.items()) if f not in cached_set]
    return c2n, np.array(new_rows, dtype=np.int64)


def update_neighbours(neigh, c2n):
    """re-index cached lists onto the new input, dropping deleted rows and empty lists (cache.py:51-71)"""
    out = []
    for nlist in neigh:
        m = c2n[np.asarray(nlist, dtype=np.int64)]
        m = m[m >= 0]
        if len(m):
            out.append(m)
    return out


def _band_lists(indptr, indices, n_features, max_dist, select_ind):
    """The lists cluster_features collects (breakfast.py:314-318): for every distinct length q of the query
    rows, in first-appearance order, one list per query row of the band = its neighbours inside the band.
    One GPU call for all query rows; the per-band restriction is a mask on the result."""
    nf = np.asarray(n_features).ravel()
    rows = np.arange(len(nf), dtype=np.int64) if select_ind is None else np.asarray(select_ind, dtype=np.int64)
    if len(rows) == 0:
        return []
    ptr, idx = _lib.neighbours_csr(indptr, indices, max_dist, rows)
    out = []
    for q in dict.fromkeys(nf[rows].tolist()):
        in_band = np.isclose(nf, q, atol=max_dist)
        for s in np.flatnonzero(in_band[rows]):
            l = idx[ptr[s]: ptr[s + 1]].astype(np.int64)
            out.append(l[in_band[l]])
    return out


def cluster_with_cache(meta, indptr, indices, max_dist, input_cache, output_cache):
    """-> canonical labels (smallest row index per component) for the rows of `meta`."""
    n = len(meta)
    neigh, select_ind = [], None
    try:
        if input_cache is None:
            raise CacheMismatch()
        cache = load(input_cache, max_dist)
        c2n, select_ind = map_features(cache["meta"]["feature"], meta["feature"])
        neigh += update_neighbours(cache["neigh"], c2n)
    except CacheMismatch:
        print("Imported cached results are not available. "
              "Distance matrix of complete dataset will be calculated.")
        neigh, select_ind = [], None
    neigh = neigh + _band_lists(indptr, indices, meta["n_features"], max_dist, select_ind)
    if output_cache:
        save(output_cache, neigh, meta, max_dist)
    return _lib.labels_from_lists(n, neigh)
