"""Side-car neighbour cache: the reference's incremental cache (src/breakfast/cache.py) in a container that scales.

The reference pickles ``{"max_dist", "version", "neigh", "meta"}`` with ``neigh`` = a Python list of one numpy array per
neighbour list (3M objects at 1M rows) and ``meta`` = a DataFrame of id tuples and feature strings (cache.py:18-32);
``breakfast_amd/cache.py`` reads and writes exactly that for interchange.  This module keeps the same SEMANTICS in three flat
arrays: rows of a new input are matched to cached rows by their feature string (cache.py:94-112) — here by two 64-bit hashes of
it —, cached lists are re-indexed onto the new input and lose the rows that are gone (cache.py:51-71; a list that is left
empty is dropped), lists are computed only for rows that are new (query = new rows x all columns, breakfast.py:241-245,
:300-304), and the components come from the union of old and new lists (each list united as a path, breakfast.py:93-113).

File (little endian): MAGIC, int32 max_dist, int64 n_rows, int64 n_lists, int64 total; uint64[n_rows][2] feature hashes;
int64[n_lists + 1] list offsets; int32[total] list members.  Chosen by the file's magic on input and by the suffix
``.bfkc`` on output; a pickle cache can be read and continued as a side-car, not the other way round (the pickle needs the
feature strings and id tuples, which a side-car does not hold).
"""

from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

from . import _lib
from . import cache as ca

MAGIC = b"BFKCACHE\x01\n"
# format 2 = the same layout with a promise: every list lies inside the neighbourhood (at max_dist) of one row OF THE FILE, and
# together the lists hold every edge — what a run without an input cache writes, and a run that continues such a cache without
# losing a row.  A list carried over from a row that is gone still chains that row's neighbours (cache.py:51-71): such a file is
# format 1.  The device stages take an exact cache whose rows are all still in the input as proof that the cache run equals
# the no-cache run (fastpath._run_sidecar_on_device).
MAGIC_EXACT = b"BFKCACHE\x02\n"
SUFFIX = ".bfkc"
_HEAD = struct.Struct("<iqqq")


def is_sidecar(path) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(len(MAGIC)) in (MAGIC, MAGIC_EXACT)
    except OSError:
        return False


def is_exact(path) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(len(MAGIC)) == MAGIC_EXACT
    except OSError:
        return False


def wants_sidecar(input_cache, output_cache) -> bool:
    return (bool(output_cache) and str(output_cache).endswith(SUFFIX)) or (input_cache is not None and is_sidecar(input_cache))


def save(path, max_dist: int, hashes, list_indptr, list_indices, exact: bool = False):
    print("Export results as side-car cache")
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    hashes = np.ascontiguousarray(hashes, dtype="<u8").reshape(-1, 2)
    off = np.ascontiguousarray(list_indptr, dtype="<i8")
    flat = np.ascontiguousarray(list_indices, dtype="<i4")
    with open(path, "wb") as f:
        f.write(MAGIC_EXACT if exact else MAGIC)
        f.write(_HEAD.pack(int(max_dist), len(hashes), len(off) - 1, len(flat)))
        hashes.tofile(f)
        off.tofile(f)
        flat.tofile(f)


def load(path, max_dist: int):
    """-> (hashes uint64[n, 2], list offsets int64[L + 1], list members int32[total]); CacheMismatch on another max_dist;
    ValueError on a file that is short, whose counts are negative, whose offsets are not a non-decreasing run from 0 to
    `total` or whose members are not rows of the cached input (a damaged file must never turn into wrong clusters)"""
    with open(path, "rb") as f:
        if f.read(len(MAGIC)) not in (MAGIC, MAGIC_EXACT):
            raise ValueError(f"{path} is not a side-car cache")
        print("Import from side-car cache")
        head = f.read(_HEAD.size)
        if len(head) != _HEAD.size:
            raise ValueError(f"{path}: truncated side-car cache (header)")
        d, n, n_lists, total = _HEAD.unpack(head)
        if n < 0 or n_lists < 0 or total < 0:
            raise ValueError(f"{path}: corrupted side-car cache (negative counts)")
        ca.validate({"max_dist": d, "version": ca.__version__}, max_dist, ca.__version__)
        hashes = np.fromfile(f, dtype="<u8", count=2 * n)
        off = np.fromfile(f, dtype="<i8", count=n_lists + 1)
        flat = np.fromfile(f, dtype="<i4", count=total)
    if len(hashes) != 2 * n or len(off) != n_lists + 1 or len(flat) != total:
        raise ValueError(f"{path}: truncated side-car cache")
    if off[0] != 0 or off[-1] != total or (n_lists and bool(np.any(np.diff(off) < 0))):
        raise ValueError(f"{path}: corrupted side-car cache (list offsets)")
    if total and (int(flat.min()) < 0 or int(flat.max()) >= n):
        raise ValueError(f"{path}: corrupted side-car cache (list member out of range)")
    return hashes.reshape(n, 2), off, flat


def _load_any(path, max_dist):
    """a side-car, or the reference's pickle turned into the same three arrays"""
    if is_sidecar(path):
        return load(path, max_dist)
    cache = ca.load(path, max_dist)
    hashes = _lib.hash_rows(list(cache["meta"]["feature"]))
    lists = [np.asarray(x, dtype=np.int32) for x in cache["neigh"]]
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    if lists:
        np.cumsum([len(x) for x in lists], out=off[1:])
    flat = np.concatenate(lists).astype(np.int32) if off[-1] else np.zeros(0, np.int32)
    return hashes, off, flat


def match_rows(cached, new):
    """cached / new: uint64[., 2] hashes (new rows are unique) -> int64[len(cached)]: the new row with the same feature
    string, or -1 (cache.map_features, cache.py:94-112)"""
    return _lib.match_hashes(cached, new)


def update_lists(off, flat, c2n):
    """re-index cached lists onto the new input, dropping rows that are gone and lists that are left empty
    (cache.update_neighbours, cache.py:51-71), on the flat arrays"""
    m = c2n[flat] if len(flat) else np.zeros(0, np.int64)
    keep = m >= 0
    cs = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
    new_len = cs[off[1:]] - cs[off[:-1]]
    new_len = new_len[new_len > 0]
    new_off = np.zeros(len(new_len) + 1, dtype=np.int64)
    np.cumsum(new_len, out=new_off[1:])
    return new_off, m[keep].astype(np.int32)


def cluster_with_sidecar(hashes, indptr, indices, max_dist, input_cache, output_cache):
    """-> canonical labels (smallest row index per component) of the rows whose feature hashes are `hashes` and whose CSR is
    (indptr, indices): cluster_features' cache branch (breakfast.py:294-326) on flat arrays"""
    n = len(indptr) - 1
    off, flat = np.zeros(1, np.int64), np.zeros(0, np.int32)
    select = None
    exact = True
    try:
        if input_cache is None:
            raise ca.CacheMismatch()
        c_hash, c_off, c_flat = _load_any(input_cache, max_dist)
        c2n = match_rows(c_hash, hashes)
        exact = is_exact(input_cache) and bool(np.all(c2n >= 0))
        off, flat = update_lists(c_off, c_flat, c2n)
        known = np.zeros(n, dtype=bool)
        known[c2n[c2n >= 0]] = True
        select = np.flatnonzero(~known).astype(np.int64)
    except ca.CacheMismatch:
        print("Imported cached results are not available. "
              "Distance matrix of complete dataset will be calculated.")
    if select is None or len(select):
        ptr, idx = _lib.neighbours_csr(indptr, indices, max_dist, select)
        off = np.concatenate([off, off[-1] + ptr[1:]])
        flat = np.concatenate([flat, idx.astype(np.int32)])
    if output_cache:
        save(output_cache, max_dist, hashes, off, flat, exact)
    return _lib.labels_from_csr(n, off, flat)
