"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see bfk_oracle.c).  Function names and argument meaning follow the reference
(src/breakfast/breakfast.py) so parity tests read like the reference's own; the compute is the
C restatement in bfk_oracle.c, the glue below restates the few pandas lines around it:

  sparse_feature_matrix     breakfast.py:193-215
  get_neighbours_batch      breakfast.py:223-276
  cluster_features          breakfast.py:279-340   (cluster-id assignment :329-339)
  cluster_identical_features breakfast.py:343-364
  cluster                   breakfast.py:82-89
  write_output_bytes        breakfast.py:32-69
"""

from __future__ import annotations

import csv
import ctypes as C
import io
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)


def build() -> Path:
    so = _HERE / "libbfk_oracle.so"
    src = _HERE / "bfk_oracle.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE)], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(str(build()))
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_build_csr.argtypes = [C.c_char_p, c_i64p, C.c_int64, C.c_char_p, C.c_int64, c_i32p,
                                    C.POINTER(c_i32p), c_i64p, c_i32p]
        L.orc_get_neighbours_batch.argtypes = [c_i32p, c_i32p, C.c_int64, c_i64p, C.c_int64, C.c_int32, c_i64p,
                                               C.c_int64, C.c_int, C.POINTER(c_i64p), C.POINTER(c_i64p), c_i64p]
        L.orc_cluster_features.argtypes = [c_i32p, c_i32p, C.c_int64, C.c_int32, c_i64p, C.c_int64, C.c_int, c_i32p,
                                           c_i32p, C.POINTER(c_i64p), C.POINTER(c_i64p), c_i64p, c_i64p]
        L.orc_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p32(a):
    return a.ctypes.data_as(c_i32p)


def _p64(a):
    return a.ctypes.data_as(c_i64p)


def _n_threads(n_threads):
    if n_threads is None:
        n_threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    return max(1, int(n_threads))


def sparse_feature_matrix(features, feature_sep):
    """-> (indptr int32[N+1], indices int32[nnz], n_vocab).  Floats (NaN) count as empty rows."""
    rows = [b"" if isinstance(f, float) else str(f).encode() for f in features]
    off = np.zeros(len(rows) + 1, dtype=np.int64)
    if rows:
        off[1:] = np.cumsum([len(r) for r in rows])
    buf = b"".join(rows)
    sep = feature_sep.encode()
    indptr = np.zeros(len(rows) + 1, dtype=np.int32)
    out = c_i32p()
    nnz = C.c_int64()
    nv = C.c_int32()
    rc = lib().orc_build_csr(buf, _p64(off), len(rows), sep, len(sep), _p32(indptr), C.byref(out), C.byref(nnz),
                             C.byref(nv))
    if rc != 0:
        raise ValueError("empty separator" if rc == -2 else f"oracle error {rc}")
    indices = np.ctypeslib.as_array(out, shape=(max(nnz.value, 1),))[: nnz.value].copy()
    lib().orc_free(out)
    return indptr, indices.astype(np.int32), int(nv.value)


def _take_lists(off_p, flat_p, n_lists):
    off = np.ctypeslib.as_array(off_p, shape=(n_lists + 1,)).copy()
    flat = np.ctypeslib.as_array(flat_p, shape=(max(int(off[-1]), 1),))[: int(off[-1])].copy()
    lib().orc_free(off_p)
    lib().orc_free(flat_p)
    return off, flat


def get_neighbours_batch(indptr, indices, n_features_all, n_features_query, max_dist, select_ind=None,
                         n_threads=None):
    """-> list of int64 arrays, one per query row in the band, exactly as the reference orders them."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    nf = np.ascontiguousarray(n_features_all, dtype=np.int64)
    sel = None if select_ind is None else np.ascontiguousarray(select_ind, dtype=np.int64)
    off_p, flat_p, nl = c_i64p(), c_i64p(), C.c_int64()
    rc = lib().orc_get_neighbours_batch(_p32(indptr), _p32(indices), len(indptr) - 1, _p64(nf), int(n_features_query),
                                        int(max_dist), None if sel is None else _p64(sel),
                                        0 if sel is None else len(sel), _n_threads(n_threads), C.byref(off_p),
                                        C.byref(flat_p), C.byref(nl))
    if rc != 0:
        raise MemoryError(f"oracle error {rc}")
    off, flat = _take_lists(off_p, flat_p, nl.value)
    return [flat[off[i]: off[i + 1]] for i in range(nl.value)]


def cluster_csr(indptr, indices, max_dist, select_ind=None, n_threads=None, want_neigh=False):
    """Band loop + graph + components on a CSR.  -> dict(labels, comp_order, n_merges[, neigh_off, neigh_flat])"""
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n = len(indptr) - 1
    labels = np.zeros(max(n, 1), dtype=np.int32)
    order = np.zeros(max(n, 1), dtype=np.int32)
    sel = None if select_ind is None else np.ascontiguousarray(select_ind, dtype=np.int64)
    off_p, flat_p, nl, nm = c_i64p(), c_i64p(), C.c_int64(), C.c_int64()
    rc = lib().orc_cluster_features(_p32(indptr), _p32(indices), n, int(max_dist),
                                    None if sel is None else _p64(sel), 0 if sel is None else len(sel),
                                    _n_threads(n_threads), _p32(labels), _p32(order),
                                    C.byref(off_p) if want_neigh else None, C.byref(flat_p) if want_neigh else None,
                                    C.byref(nl) if want_neigh else None, C.byref(nm))
    if rc != 0:
        raise MemoryError(f"oracle error {rc}")
    res = {"labels": labels[:n], "comp_order": order[:n], "n_merges": int(nm.value)}
    if want_neigh:
        res["neigh_off"], res["neigh_flat"] = _take_lists(off_p, flat_p, nl.value)
    return res


def _assign_ids(comp_order, group_size, min_cluster_size):
    """breakfast.py:329-339: ids 1.. in the order networkx yields components, size = sum of id-tuple lengths."""
    n = len(comp_order)
    cid = np.zeros(n, dtype=np.int32)  # 0 = pd.NA
    if n == 0:
        return cid
    ncomp = int(comp_order.max()) + 1 if (comp_order >= 0).any() else 0
    if ncomp:
        m = comp_order >= 0
        size = np.bincount(comp_order[m], weights=np.asarray(group_size)[m], minlength=ncomp)
        keep = size >= min_cluster_size
        new_id = np.cumsum(keep) * keep
        cid[m] = new_id[comp_order[m]]
    return cid


def cluster_features(features, group_size, feature_sep, max_dist, min_cluster_size, n_threads=None,
                     want_neigh=False):
    """features: unique strings (collapse_duplicates order); group_size[i] = len(id tuple i).
    -> dict(n_features, cluster_id (0 = NA), labels, indptr, indices, ...)"""
    indptr, indices, nv = sparse_feature_matrix(features, feature_sep)
    if len(indices) == 0:
        # scipy cannot infer the shape of an all-empty matrix: the reference raises here (breakfast.py:214)
        raise ValueError("unable to infer matrix dimensions")
    res = cluster_csr(indptr, indices, max_dist, n_threads=n_threads, want_neigh=want_neigh)
    res["indptr"], res["indices"], res["n_vocab"] = indptr, indices, nv
    res["n_features"] = np.diff(indptr).astype(np.int64)
    res["cluster_id"] = _assign_ids(res["comp_order"], group_size, min_cluster_size)
    return res


def cluster_identical_features(group_size, min_cluster_size):
    """breakfast.py:343-364: every unique string is its own cluster if its tuple is big enough."""
    keep = np.asarray(group_size) >= min_cluster_size
    return (np.cumsum(keep) * keep).astype(np.int32)


def cluster(features, group_size, sep2, max_dist, min_cluster_size, n_threads=None):
    """breakfast.py:82-89 -> cluster_id per unique row (0 = NA)"""
    if max_dist == 0:
        return cluster_identical_features(group_size, min_cluster_size)
    return cluster_features(features, group_size, sep2, max_dist, min_cluster_size, n_threads)["cluster_id"]


def collapse(features):
    """collapse_duplicates (breakfast.py:72-79) on plain lists: -> (unique features in first-appearance
    order, list of member-index lists)."""
    pos: dict[str, int] = {}
    members: list[list[int]] = []
    for i, f in enumerate(features):
        j = pos.setdefault(f, len(pos))
        if j == len(members):
            members.append([])
        members[j].append(i)
    return list(pos), members


def write_output_bytes(ids, members, cluster_id):
    """write_output (breakfast.py:32-69) for input ids `ids` (input order), unique-row member lists and the
    per-unique-row cluster ids (0 = NA): renumber by first appearance in input order, NA -> empty field."""
    per_seq = np.zeros(len(ids), dtype=np.int64)
    for u, mem in enumerate(members):
        per_seq[mem] = cluster_id[u]
    remap: dict[int, int] = {}
    buf = io.StringIO()
    w = csv.writer(buf, delimiter="\t", lineterminator="\n", quoting=csv.QUOTE_MINIMAL)
    w.writerow(["id", "cluster_id"])
    for i, c in zip(ids, per_seq):
        if c == 0:
            w.writerow([i, ""])
        else:
            w.writerow([i, remap.setdefault(int(c), len(remap) + 1)])
    return buf.getvalue().encode()


def pipeline_bytes(ids, features, sep2, max_dist, min_cluster_size, n_threads=None):
    """filtered feature strings -> clusters.tsv bytes (collapse + cluster + write_output)."""
    ufeats, members = collapse(features)
    gs = np.array([len(m) for m in members], dtype=np.int32)
    cid = cluster(ufeats, gs, sep2, max_dist, min_cluster_size, n_threads)
    return write_output_bytes(ids, members, cid)
