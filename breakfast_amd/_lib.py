"""ctypes binding of libbfk.so (include/bfk.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in the product path."""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["BFK_LIB"]) if os.environ.get("BFK_LIB") else _HERE / "libbfk.so"  # BFK_LIB: A/B builds
_lib = None

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)


class BfkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbfk error {code}: {msg}")
        self.code = code
        self.msg = msg


class Stats(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64), ("nnz", C.c_int64), ("pairs_resolved", C.c_int64), ("pairs_in_band", C.c_int64),
        ("pairs_filtered", C.c_int64), ("n_candidates", C.c_int64), ("n_edges", C.c_int64), ("n_retry_slices", C.c_int64),
        ("max_row_len", C.c_int32), ("sig_words", C.c_int32), ("n_work_items", C.c_int32), ("profiled", C.c_int32),
        ("ms_prep", C.c_float), ("ms_prefilter", C.c_float), ("ms_verify", C.c_float), ("ms_flatten", C.c_float),
        ("ms_total", C.c_float), ("path", C.c_int32), ("n_gpus_used", C.c_int32), ("n_connected", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TextStats(C.Structure):
    _fields_ = [("text_bytes", C.c_int64), ("n_rows", C.c_int64), ("nnz", C.c_int64), ("table_slots", C.c_int64),
                ("n_vocab", C.c_int32), ("table_growths", C.c_int32), ("host_fallback", C.c_int32), ("reserved_", C.c_int32),
                ("ms_h2d", C.c_float), ("ms_scan", C.c_float), ("ms_hash", C.c_float), ("ms_ids", C.c_float),
                ("ms_total", C.c_float), ("ms_head", C.c_float), ("reserved2_", C.c_float), ("n_invalid", C.c_int64),
                ("n_empty", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class FilterOpts(C.Structure):
    _fields_ = [("var_type", C.c_int32), ("skip_ins", C.c_int32), ("skip_del", C.c_int32),
                ("trim_start", C.c_int64), ("trim_end", C.c_int64), ("reference_length", C.c_int64)]


class PrepInfo(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_unique", C.c_int64), ("nnz", C.c_int64), ("n_invalid", C.c_int64),
                ("n_vocab", C.c_int32), ("filtered", C.c_int32)]


VAR_TYPES = {"covsonar_dna": 0, "covsonar_aa": 1, "nextclade_dna": 2, "nextclade_aa": 3, "raw": 4}
EUNSUPPORTED = -7
ABI_VERSION = 3  # BFK_ABI_VERSION of include/bfk.h this binding (struct layouts, EXPORTS) was written against

EXPORTS = {
    "bfk_abi_version": (C.c_int, []),
    "bfk_device_count": (C.c_int, []),
    "bfk_last_error": (C.c_char_p, []),
    "bfk_free": (None, [C.c_void_p]),
    "bfk_build_csr": (C.c_int, [C.c_char_p, c_i64p, C.c_int64, C.c_char_p, C.c_int64, c_i32p, C.POINTER(c_i32p),
                                c_i64p, c_i32p]),
    "bfk_build_csr_device": (C.c_int, [C.c_char_p, c_i64p, C.c_int64, C.c_char_p, C.c_int64, c_i32p, C.POINTER(c_i32p),
                                       c_i64p, c_i32p]),
    "bfk_cluster_text": (C.c_int, [C.c_char_p, c_i64p, C.c_int64, C.c_char_p, C.c_int64, C.c_int32, c_i32p, C.POINTER(Stats),
                                   c_i64p, c_i32p, c_i32p]),
    "bfk_ctx_build_csr": (C.c_int, [C.c_void_p, C.c_char_p, c_i64p, C.c_int64, C.c_char_p, C.c_int64, c_i64p, c_i32p]),
    "bfk_ctx_download_csr": (C.c_int, [C.c_void_p, c_i32p, c_i32p]),
    "bfk_text_device_bytes": (C.c_int64, [C.c_int64]),
    "bfk_ctx_build_csr_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, c_i64p,
                                           c_i32p]),
    "bfk_ctx_cluster_text_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64,
                                              C.c_int32, C.c_void_p]),
    "bfk_host_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "bfk_host_free": (C.c_int, [C.c_void_p]),
    "bfk_ctx_text_stats": (C.c_int, [C.c_void_p, C.POINTER(TextStats)]),
    "bfk_cluster_csr": (C.c_int, [c_i32p, c_i32p, C.c_int64, C.c_int32, C.c_int32, c_i32p, C.POINTER(Stats)]),
    "bfk_neighbours_csr": (C.c_int, [c_i32p, c_i32p, C.c_int64, C.c_int32, c_i64p, C.c_int64, C.POINTER(c_i64p),
                                     C.POINTER(c_i32p)]),
    "bfk_labels_from_lists": (C.c_int, [C.c_int64, c_i64p, c_i32p, C.c_int64, c_i32p]),
    "bfk_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "bfk_ctx_destroy": (C.c_int, [C.c_void_p]),
    "bfk_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bfk_ctx_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "bfk_ctx_set_candidate_path": (C.c_int, [C.c_void_p, C.c_int32]),
    "bfk_ctx_upload_csr": (C.c_int, [C.c_void_p, c_i32p, c_i32p, C.c_int64]),
    "bfk_ctx_bind_csr_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "bfk_ctx_cluster": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "bfk_ctx_merge_labels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "bfk_ctx_sync": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "bfk_ctx_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "bfk_ctx_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "bfk_ctx_device_alloc": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "bfk_ctx_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bfk_ctx_set_edge_capture": (C.c_int, [C.c_void_p, C.c_int32]),
    "bfk_ctx_set_exact_edges": (C.c_int, [C.c_void_p, C.c_int32]),
    "bfk_ctx_set_token_ids": (C.c_int, [C.c_void_p, C.c_int32]),
    "bfk_ctx_edges": (C.c_int, [C.c_void_p, C.POINTER(c_i32p), c_i64p]),
    "bfk_table_open": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "bfk_table_from_buffers": (C.c_int, [C.c_char_p, c_i64p, C.c_char_p, c_i64p, C.c_int64, C.POINTER(C.c_void_p)]),
    "bfk_table_rows": (C.c_int64, [C.c_void_p]),
    "bfk_table_close": (None, [C.c_void_p]),
    "bfk_table_prepare": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.POINTER(PrepInfo)]),
    "bfk_table_group": (c_i32p, [C.c_void_p]),
    "bfk_table_weight": (c_i32p, [C.c_void_p]),
    "bfk_table_indptr": (c_i32p, [C.c_void_p]),
    "bfk_table_indices": (c_i32p, [C.c_void_p]),
    "bfk_table_invalid": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), c_i64p]),
    "bfk_table_invalid_count": (C.c_int64, [C.c_void_p]),
    "bfk_table_feature": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), c_i64p]),
    "bfk_table_id": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), c_i64p]),
    "bfk_table_write": (C.c_int, [C.c_void_p, C.c_char_p, c_i32p, c_i64p]),
    "bfk_table_prepare_device": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.POINTER(PrepInfo)]),
    "bfk_table_cluster_write_device": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32,
                                                 C.c_char_p, C.POINTER(PrepInfo), c_i64p]),
    "bfk_table_pipeline_device": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32,
                                            C.c_char_p, C.POINTER(PrepInfo), c_i64p]),
    "bfk_table_cluster_write_device_gpus": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32, C.c_int32,
                                                      C.c_char_p, C.POINTER(PrepInfo), c_i64p]),
    "bfk_table_pipeline_device_gpus": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_char_p, C.POINTER(PrepInfo), c_i64p]),
    "bfk_table_cluster_write_device_cache": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32,
                                                       C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(PrepInfo), c_i64p]),
    "bfk_table_pipeline_device_cache": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32,
                                                  C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(PrepInfo), c_i64p]),
    "bfk_table_features": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(c_i64p)]),
    "bfk_table_ids": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(c_i64p)]),
    "bfk_table_cluster_write": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, c_i64p]),
    "bfk_hash_rows": (C.c_int, [C.c_char_p, c_i64p, C.c_int64, C.POINTER(C.c_uint64)]),
    "bfk_table_feature_hashes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "bfk_match_hashes": (C.c_int, [C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_uint64), C.c_int64, c_i64p]),
    "bfk_warmup": (C.c_int, [C.c_int, C.c_int64, C.c_int64]),
    "bfk_preload_start": (C.c_int, [C.c_char_p, C.c_int, C.c_int64, C.c_int64]),
    "bfk_preload_wait": (C.c_int, []),
    "bfk_preload_join": (None, []),
}


def load():
    """Load libbfk.so once.  Raises if it was not built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP library must be built first "
                "(make -C breakfast_amd/csrc, or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.bfk_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {lib.bfk_abi_version()}, this binding expects {ABI_VERSION}: "
                               "rebuild it (make -C breakfast_amd/csrc)")
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise BfkError(rc, load().bfk_last_error().decode(errors="replace"))


def _p32(a):
    return a.ctypes.data_as(c_i32p)


def _p64(a):
    return a.ctypes.data_as(c_i64p)


def build_csr(features, sep: str):
    """bfk_build_csr on a sequence of str (floats/NaN count as empty rows, breakfast.py:200-201).
    -> (indptr int32[N+1], indices int32[nnz], n_vocab)"""
    lib = load()
    rows = [b"" if isinstance(f, float) else f.encode() for f in features]
    n = len(rows)
    off = np.zeros(n + 1, dtype=np.int64)
    if n:
        np.cumsum(np.fromiter((len(r) for r in rows), dtype=np.int64, count=n), out=off[1:])
    buf = b"".join(rows)
    sepb = sep.encode()
    indptr = np.zeros(n + 1, dtype=np.int32)
    out = c_i32p()
    nnz = C.c_int64()
    nv = C.c_int32()
    rc = lib.bfk_build_csr(buf, _p64(off), n, sepb, len(sepb), _p32(indptr), C.byref(out), C.byref(nnz), C.byref(nv))
    if rc == -1 and len(sepb) == 0:
        raise ValueError("empty separator")
    _check(rc)
    indices = np.ctypeslib.as_array(out, shape=(max(nnz.value, 1),))[: nnz.value].copy()
    lib.bfk_free(out)
    return indptr, indices, int(nv.value)


def pack_rows(features):
    """sequence of str (floats/NaN = empty rows, breakfast.py:200-201) -> (one byte buffer, int64 offsets[N+1])"""
    rows = [b"" if isinstance(f, float) else f.encode() for f in features]
    off = np.zeros(len(rows) + 1, dtype=np.int64)
    if rows:
        np.cumsum(np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows)), out=off[1:])
    return b"".join(rows), off


def build_csr_bytes(buf: bytes, row_off, sep: str, device: bool = False):
    """bfk_build_csr on the C-ABI's own input: one byte buffer + int64 offsets[N+1] (no Python strings touched).
    device=True: bfk_build_csr_device — the same contract computed by the HIP tokeniser."""
    lib = load()
    off = np.ascontiguousarray(row_off, dtype=np.int64)
    n = len(off) - 1
    sepb = sep.encode()
    indptr = np.zeros(n + 1, dtype=np.int32)
    out = c_i32p()
    nnz = C.c_int64()
    nv = C.c_int32()
    fn = lib.bfk_build_csr_device if device else lib.bfk_build_csr
    rc = fn(buf, _p64(off), n, sepb, len(sepb), _p32(indptr), C.byref(out), C.byref(nnz), C.byref(nv))
    if rc == -1 and len(sepb) == 0:
        raise ValueError("empty separator")
    if rc == EUNSUPPORTED:
        raise Unsupported(lib.bfk_last_error().decode(errors="replace"))
    _check(rc)
    indices = np.ctypeslib.as_array(out, shape=(max(nnz.value, 1),))[: nnz.value].copy()
    lib.bfk_free(out)
    return indptr, indices, int(nv.value)


def build_csr_device(features, sep: str):
    """bfk_build_csr_device on a sequence of str: the CSR of sparse_feature_matrix, computed on the GPU."""
    buf, off = pack_rows(features)
    return build_csr_bytes(buf, off, sep, device=True)


class PinnedBuffer:
    """bfk_host_alloc / bfk_host_free: page-locked host memory a caller builds its profile text in (copies from it run at the
    PCIe rate from the first byte).  `.view` is a writable uint8 numpy view; pass the object itself as `buf` to cluster_text."""

    def __init__(self, nbytes: int):
        self.lib = load()
        p = C.c_void_p()
        _check(self.lib.bfk_host_alloc(int(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p, int(nbytes)
        self.view = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(int(nbytes), 1),))[: int(nbytes)]

    def free(self):
        if self.ptr:
            self.view = None
            self.lib.bfk_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def cluster_text(buf, row_off, sep: str, max_dist: int, want_stats: bool = True, indptr_out=None, labels_out=None, want_vocab: bool = True):
    """bfk_cluster_text: profile text -> (labels, stats dict, nnz, n_vocab); the CSR is built and stays on the device.
    buf: bytes, or a PinnedBuffer (bfk_host_alloc).  want_stats=False passes stats_out = NULL (the counters that need
    host-side sums are then not gathered); indptr_out: optional int32[N+1] array that receives the CSR's row pointer;
    labels_out: optional int32[N] array to write the labels into (no allocation per call); want_vocab=False passes
    n_vocab_out = NULL (n_vocab comes back as -1): with BFK_TOK_ANY_IDS=1 a max_dist-1 step then keeps table slots as ids."""
    lib = load()
    off = np.ascontiguousarray(row_off, dtype=np.int64)
    n = len(off) - 1
    sepb = sep.encode()
    labels = np.empty(max(n, 1), dtype=np.int32) if labels_out is None else labels_out
    st = Stats()
    nnz, nv = C.c_int64(), C.c_int32()
    if isinstance(buf, PinnedBuffer):
        buf = C.cast(buf.ptr, C.c_char_p)
    rc = lib.bfk_cluster_text(buf, _p64(off), n, sepb, len(sepb), int(max_dist), _p32(labels),
                              C.byref(st) if want_stats else None, C.byref(nnz), C.byref(nv) if want_vocab else None,
                              None if indptr_out is None else _p32(indptr_out))
    if rc == -1 and len(sepb) == 0:
        raise ValueError("empty separator")
    _check(rc)
    return labels[:n], (st.as_dict() if want_stats else None), int(nnz.value), int(nv.value) if want_vocab else -1


def cluster_csr(indptr, indices, max_dist: int, n_gpus: int = 1):
    """bfk_cluster_csr -> (labels int32[N] = min row index of the component, stats dict)"""
    lib = load()
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n = len(indptr) - 1
    labels = np.empty(max(n, 1), dtype=np.int32)
    st = Stats()
    _check(lib.bfk_cluster_csr(_p32(indptr), _p32(indices), n, int(max_dist), int(n_gpus), _p32(labels), C.byref(st)))
    return labels[:n], st.as_dict()


def neighbours_csr(indptr, indices, max_dist: int, select_ind=None):
    """bfk_neighbours_csr -> (nbr_indptr int64[Q+1], nbr_indices int32[...]) ascending lists incl. self"""
    lib = load()
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n = len(indptr) - 1
    sel = None if select_ind is None else np.ascontiguousarray(select_ind, dtype=np.int64)
    op, oi = c_i64p(), c_i32p()
    _check(lib.bfk_neighbours_csr(_p32(indptr), _p32(indices), n, int(max_dist), None if sel is None else _p64(sel),
                                  0 if sel is None else len(sel), C.byref(op), C.byref(oi)))
    nq = n if sel is None else len(sel)
    nbr_indptr = np.ctypeslib.as_array(op, shape=(nq + 1,)).copy()
    tot = int(nbr_indptr[-1])
    nbr_indices = np.ctypeslib.as_array(oi, shape=(max(tot, 1),))[:tot].copy()
    lib.bfk_free(op)
    lib.bfk_free(oi)
    return nbr_indptr, nbr_indices


def hash_rows(features):
    """bfk_hash_rows on a sequence of str -> uint64[N, 2]: the two hashes of every feature string (side-car cache)"""
    buf, off = pack_rows(features)
    out = np.zeros((max(len(off) - 1, 1), 2), dtype=np.uint64)
    _check(load().bfk_hash_rows(buf, _p64(off), len(off) - 1, out.ctypes.data_as(C.POINTER(C.c_uint64))))
    return out[: len(off) - 1]


def match_hashes(cached, new):
    """bfk_match_hashes: uint64[., 2] hash pairs -> int64[len(cached)]: the row of `new` with the same pair, or -1"""
    a = np.ascontiguousarray(cached, dtype=np.uint64).reshape(-1, 2)
    b = np.ascontiguousarray(new, dtype=np.uint64).reshape(-1, 2)
    out = np.full(max(len(a), 1), -1, dtype=np.int64)
    u64p = C.POINTER(C.c_uint64)
    _check(load().bfk_match_hashes(a.ctypes.data_as(u64p), len(a), b.ctypes.data_as(u64p), len(b), _p64(out)))
    return out[: len(a)]


def labels_from_csr(n_rows: int, list_indptr, list_indices):
    """bfk_labels_from_lists on lists given as CSR (int64 offsets, int32 members): no Python object per list"""
    lib = load()
    off = np.ascontiguousarray(list_indptr, dtype=np.int64)
    flat = np.ascontiguousarray(list_indices if len(list_indices) else np.zeros(1, np.int32), dtype=np.int32)
    labels = np.empty(max(n_rows, 1), dtype=np.int32)
    _check(lib.bfk_labels_from_lists(int(n_rows), _p64(off), _p32(flat), len(off) - 1, _p32(labels)))
    return labels[:n_rows]


def labels_from_lists(n_rows: int, lists):
    """bfk_labels_from_lists: components of a list of index arrays (each list united as a path)."""
    lib = load()
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    if len(lists):
        np.cumsum(np.fromiter((len(x) for x in lists), dtype=np.int64, count=len(lists)), out=off[1:])
    flat = (np.concatenate([np.asarray(x, dtype=np.int32) for x in lists]) if off[-1] else np.zeros(1, np.int32))
    flat = np.ascontiguousarray(flat, dtype=np.int32)
    labels = np.empty(max(n_rows, 1), dtype=np.int32)
    _check(lib.bfk_labels_from_lists(int(n_rows), _p64(off), _p32(flat), len(lists), _p32(labels)))
    return labels[:n_rows]


def text_device_bytes(text_bytes: int) -> int:
    """bytes a device text buffer handed to Context.build_csr_device / cluster_text_device must have"""
    return int(load().bfk_text_device_bytes(int(text_bytes)))


class Unsupported(Exception):
    """the native reader declined the input (BFK_EUNSUPPORTED): use the pandas path"""


class Table:
    """bfk_table_*: native reader / filter + collapse + CSR / writer around the GPU path (include/bfk.h)."""

    def __init__(self, handle):
        self.lib = load()
        self.h = handle
        self.info = None

    @classmethod
    def open(cls, path, sep: str, id_col: str, feature_col: str):
        lib = load()
        h = C.c_void_p()
        sepb = sep.encode()
        rc = lib.bfk_table_open(str(path).encode(), sepb, len(sepb), id_col.encode(), feature_col.encode(), C.byref(h))
        if rc == EUNSUPPORTED:
            raise Unsupported(lib.bfk_last_error().decode(errors="replace"))
        _check(rc)
        return cls(h)

    @classmethod
    def from_lists(cls, ids, features):
        """table from Python strings (tests, API callers); what UTF-8 cannot encode (lone surrogates) raises Unsupported"""
        lib = load()

        def pack(strs):
            try:
                rows = [s.encode("utf-8") for s in strs]
            except UnicodeEncodeError as e:
                raise Unsupported(str(e))
            off = np.zeros(len(rows) + 1, dtype=np.int64)
            if rows:
                np.cumsum(np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows)), out=off[1:])
            return b"".join(rows), off

        ib, io = pack(ids)
        fb, fo = pack(features)
        h = C.c_void_p()
        rc = lib.bfk_table_from_buffers(ib, _p64(io), fb, _p64(fo), len(io) - 1, C.byref(h))
        if rc == EUNSUPPORTED:
            raise Unsupported(lib.bfk_last_error().decode(errors="replace"))
        _check(rc)
        return cls(h)

    def close(self):
        if self.h:
            self.lib.bfk_table_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self.lib.bfk_table_rows(self.h))

    def prepare(self, sep2: str, var_type: str, skip_ins, skip_del, trim_start, trim_end, reference_length):
        opts = FilterOpts(VAR_TYPES[var_type], int(bool(skip_ins)), int(bool(skip_del)), int(trim_start), int(trim_end),
                          int(reference_length))
        info = PrepInfo()
        sepb = sep2.encode()
        rc = self.lib.bfk_table_prepare(self.h, sepb, len(sepb), C.byref(opts), C.byref(info))
        if rc == EUNSUPPORTED:
            raise Unsupported(self.lib.bfk_last_error().decode(errors="replace"))
        if rc == -1 and len(sepb) == 0:
            raise ValueError("empty separator")
        _check(rc)
        self.info = info
        return info

    def prepare_device(self, sep2: str, var_type: str, skip_ins, skip_del, trim_start, trim_end, reference_length):
        """bfk_table_prepare_device: the same contract computed on the GPU (filter in the tokeniser, collapse by row hash)"""
        opts = FilterOpts(VAR_TYPES[var_type], int(bool(skip_ins)), int(bool(skip_del)), int(trim_start), int(trim_end),
                          int(reference_length))
        info = PrepInfo()
        sepb = sep2.encode()
        rc = self.lib.bfk_table_prepare_device(self.h, sepb, len(sepb), C.byref(opts), C.byref(info))
        if rc == EUNSUPPORTED:
            raise Unsupported(self.lib.bfk_last_error().decode(errors="replace"))
        if rc == -1 and len(sepb) == 0:
            raise ValueError("empty separator")
        _check(rc)
        self.info = info
        return info

    def cluster_write_device(self, sep2: str, var_type: str, skip_ins, skip_del, trim_start, trim_end, reference_length, max_dist: int,
                             min_cluster_size: int, path, n_gpus: int = 1, cache_path=None, in_cache=None):
        """bfk_table_cluster_write_device[_gpus | _cache] -> (PrepInfo, clusters written)"""
        opts = FilterOpts(VAR_TYPES[var_type], int(bool(skip_ins)), int(bool(skip_del)), int(trim_start), int(trim_end),
                          int(reference_length))
        info, n = PrepInfo(), C.c_int64()
        sepb = sep2.encode()
        if cache_path is not None or in_cache is not None:
            rc = self.lib.bfk_table_cluster_write_device_cache(self.h, sepb, len(sepb), C.byref(opts), int(max_dist), int(min_cluster_size),
                                                               str(path).encode(), None if in_cache is None else str(in_cache).encode(),
                                                               None if cache_path is None else str(cache_path).encode(), C.byref(info), C.byref(n))
        elif n_gpus > 1:
            rc = self.lib.bfk_table_cluster_write_device_gpus(self.h, sepb, len(sepb), C.byref(opts), int(max_dist), int(min_cluster_size),
                                                              int(n_gpus), str(path).encode(), C.byref(info), C.byref(n))
        else:
            rc = self.lib.bfk_table_cluster_write_device(self.h, sepb, len(sepb), C.byref(opts), int(max_dist), int(min_cluster_size),
                                                         str(path).encode(), C.byref(info), C.byref(n))
        if rc == EUNSUPPORTED:
            raise Unsupported(self.lib.bfk_last_error().decode(errors="replace"))
        _check(rc)
        self.info = info
        return info, int(n.value)

    def _view(self, fn, n):
        p = fn(self.h)
        return np.ctypeslib.as_array(p, shape=(max(int(n), 1),))[: int(n)]

    @property
    def group(self):
        return self._view(self.lib.bfk_table_group, self.info.n_rows)

    @property
    def weight(self):
        return self._view(self.lib.bfk_table_weight, self.info.n_unique)

    @property
    def indptr(self):
        return self._view(self.lib.bfk_table_indptr, self.info.n_unique + 1)

    @property
    def indices(self):
        return self._view(self.lib.bfk_table_indices, self.info.nnz)

    def _str(self, fn, i, owned):
        p, n = C.c_void_p(), C.c_int64()
        _check(fn(self.h, int(i), C.byref(p), C.byref(n)))
        s = C.string_at(p, n.value).decode("utf-8")
        if owned:
            self.lib.bfk_free(p)
        return s

    def invalid_count(self) -> int:
        return int(self.lib.bfk_table_invalid_count(self.h))

    def invalid(self, i):
        return self._str(self.lib.bfk_table_invalid, i, False)

    def feature(self, u):
        return self._str(self.lib.bfk_table_feature, u, True)

    def id(self, r):
        return self._str(self.lib.bfk_table_id, r, False)

    def _strings(self, fn, n):
        """bulk accessor -> list of n str (one decode, n slices)"""
        p, o = C.c_void_p(), c_i64p()
        _check(fn(self.h, C.byref(p), C.byref(o)))
        off = np.ctypeslib.as_array(o, shape=(int(n) + 1,)).tolist()
        raw = C.string_at(p, off[-1])
        self.lib.bfk_free(p)
        self.lib.bfk_free(o)
        if raw.isascii():  # one decode, n slices (byte offsets are character offsets)
            blob = raw.decode("ascii")
            return [blob[off[i]: off[i + 1]] for i in range(int(n))]
        return [raw[off[i]: off[i + 1]].decode("utf-8") for i in range(int(n))]

    def features(self):
        """filtered feature strings of the unique rows (collapse_duplicates order)"""
        return self._strings(self.lib.bfk_table_features, self.info.n_unique)

    def feature_hashes(self):
        """uint64[n_unique, 2]: the two hashes of every unique row's filtered feature string (side-car cache)"""
        out = np.zeros((max(int(self.info.n_unique), 1), 2), dtype=np.uint64)
        _check(self.lib.bfk_table_feature_hashes(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out[: int(self.info.n_unique)]

    def ids(self):
        """ids of the input rows"""
        return self._strings(self.lib.bfk_table_ids, len(self))

    def write(self, path, cluster_of_unique) -> int:
        c = np.ascontiguousarray(cluster_of_unique, dtype=np.int32)
        if len(c) != self.info.n_unique:
            raise ValueError("one cluster number per unique row expected")
        if len(c) == 0:
            c = np.zeros(1, np.int32)
        n = C.c_int64()
        _check(self.lib.bfk_table_write(self.h, str(path).encode(), _p32(c), C.byref(n)))
        return int(n.value)


class Context:
    """Resident-context API (bfk_ctx_*): CSR and labels live in HBM, launches go to a HIP stream."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        _check(self.lib.bfk_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.n_rows = 0
        self._owned = []
        self.config_epoch = 0  # bumped by every setter that changes WHICH kernels a step runs (distributed.py keys on it)

    def close(self):
        if self.h:
            for p in self._owned:
                self.lib.bfk_ctx_device_free(self.h, p)
            self._owned = []
            self.lib.bfk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        _check(self.lib.bfk_ctx_set_stream(self.h, C.c_void_p(stream_handle or 0)))

    def set_profiling(self, on=True):
        _check(self.lib.bfk_ctx_set_profiling(self.h, 1 if on else 0))

    def set_exact_edges(self, on: bool = True):
        """on: every candidate pair is checked, `n_edges` is the number of edges of the graph; off (default): labels-only
        steps at max_dist >= 3 drop candidates whose rows are already in one component (`n_connected`)."""
        _check(self.lib.bfk_ctx_set_exact_edges(self.h, 1 if on else 0))
        self.config_epoch += 1

    def set_token_ids(self, any_ids: bool = True):
        """any_ids: labels-only text steps at max_dist 1 keep the vocabulary table's slot numbers as column ids (an injective
        renaming of the reference's first-appearance ids: same labels, three kernels fewer); off (default): the reference's CSR"""
        _check(self.lib.bfk_ctx_set_token_ids(self.h, 1 if any_ids else 0))
        self.config_epoch += 1

    def set_candidate_path(self, mode: str = "auto"):
        _check(self.lib.bfk_ctx_set_candidate_path(self.h, {"auto": 0, "allpairs": 1, "join": 2, "prefix": 3}[mode]))
        self.config_epoch += 1

    def set_edge_capture(self, on: bool = True):
        _check(self.lib.bfk_ctx_set_edge_capture(self.h, 1 if on else 0))
        self.config_epoch += 1

    def upload_csr(self, indptr, indices):
        indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.n_rows = len(indptr) - 1
        _check(self.lib.bfk_ctx_upload_csr(self.h, _p32(indptr), _p32(indices), self.n_rows))

    def bind_csr_device(self, d_indptr: int, d_indices: int, n_rows: int):
        self.n_rows = int(n_rows)
        _check(self.lib.bfk_ctx_bind_csr_device(self.h, C.c_void_p(d_indptr), C.c_void_p(d_indices), self.n_rows))

    def build_csr(self, buf: bytes, row_off, sep: str):
        """bfk_ctx_build_csr: tokenise on the device, leave the CSR resident and bound -> (nnz, n_vocab)"""
        off = np.ascontiguousarray(row_off, dtype=np.int64)
        sepb = sep.encode()
        nnz, nv = C.c_int64(), C.c_int32()
        rc = self.lib.bfk_ctx_build_csr(self.h, buf, _p64(off), len(off) - 1, sepb, len(sepb), C.byref(nnz), C.byref(nv))
        if rc == EUNSUPPORTED:
            raise Unsupported(self.lib.bfk_last_error().decode(errors="replace"))
        _check(rc)
        self.n_rows = len(off) - 1
        self.nnz = int(nnz.value)
        return int(nnz.value), int(nv.value)

    def build_csr_device(self, d_text: int, text_bytes: int, d_row_off: int, n_rows: int, sep: str):
        """bfk_ctx_build_csr_device: the tokeniser on text resident in HBM (device pointers as ints) -> (nnz, n_vocab)"""
        sepb = sep.encode()
        nnz, nv = C.c_int64(), C.c_int32()
        rc = self.lib.bfk_ctx_build_csr_device(self.h, C.c_void_p(d_text), int(text_bytes), C.c_void_p(d_row_off), int(n_rows), sepb,
                                               len(sepb), C.byref(nnz), C.byref(nv))
        if rc == EUNSUPPORTED:
            raise Unsupported(self.lib.bfk_last_error().decode(errors="replace"))
        _check(rc)
        self.n_rows = int(n_rows)
        self.nnz = int(nnz.value)
        return int(nnz.value), int(nv.value)

    def cluster_text_device(self, d_text: int, text_bytes: int, d_row_off: int, n_rows: int, sep: str, max_dist: int, d_labels: int):
        """bfk_ctx_cluster_text_device: profile strings in HBM -> labels in HBM (asynchronous behind the one wait inside)"""
        sepb = sep.encode()
        rc = self.lib.bfk_ctx_cluster_text_device(self.h, C.c_void_p(d_text), int(text_bytes), C.c_void_p(d_row_off), int(n_rows),
                                                  sepb, len(sepb), int(max_dist), C.c_void_p(d_labels))
        if rc == EUNSUPPORTED:
            raise Unsupported(self.lib.bfk_last_error().decode(errors="replace"))
        _check(rc)
        self.n_rows = int(n_rows)

    def download_csr(self, n_rows=None, nnz=None):
        """the bound CSR as numpy arrays; its sizes come from a bfk_ctx_sync unless the caller knows them"""
        if n_rows is None or nnz is None:
            st = Stats()  # (rows and entries of the CSR that is bound: also completes a text step's open bind)
            _check(self.lib.bfk_ctx_sync(self.h, C.byref(st)))
            n_rows, nnz = int(st.n_rows), int(st.nnz)
        self.n_rows, self.nnz = int(n_rows), int(nnz)
        indptr = np.empty(self.n_rows + 1, dtype=np.int32)
        indices = np.empty(max(self.nnz, 1), dtype=np.int32)
        _check(self.lib.bfk_ctx_download_csr(self.h, _p32(indptr), _p32(indices)))
        return indptr, indices[: self.nnz]

    def text_stats(self) -> dict:
        st = TextStats()
        _check(self.lib.bfk_ctx_text_stats(self.h, C.byref(st)))
        return st.as_dict()

    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _check(self.lib.bfk_ctx_device_alloc(self.h, int(nbytes), C.byref(p)))
        self._owned.append(p)
        return p.value

    def cluster(self, max_dist: int, d_labels: int, shard: int = 0, n_shards: int = 1):
        _check(self.lib.bfk_ctx_cluster(self.h, int(max_dist), int(shard), int(n_shards), C.c_void_p(d_labels)))

    def merge_labels(self, d_gathered: int, n_parts: int, d_labels: int, d_changed: int = 0):
        _check(self.lib.bfk_ctx_merge_labels(self.h, C.c_void_p(d_gathered), int(n_parts), C.c_void_p(d_labels),
                                             C.c_void_p(d_changed or 0)))

    def sync(self, want_stats: bool = True) -> dict:
        """want_stats=False: wait for the stream only (the statistics cost host work: band pair counts from the row lengths)"""
        if not want_stats:
            _check(self.lib.bfk_ctx_sync(self.h, None))
            return {}
        st = Stats()
        _check(self.lib.bfk_ctx_sync(self.h, C.byref(st)))
        return st.as_dict()

    def upload_i32(self, h: np.ndarray, d_ptr: int):
        h = np.ascontiguousarray(h, dtype=np.int32)
        _check(self.lib.bfk_ctx_upload(self.h, h.ctypes.data_as(C.c_void_p), C.c_void_p(d_ptr), h.size * 4))

    def download_i32(self, d_ptr: int, n: int) -> np.ndarray:
        out = np.empty(max(n, 1), dtype=np.int32)
        _check(self.lib.bfk_ctx_download(self.h, C.c_void_p(d_ptr), out.ctypes.data_as(C.c_void_p), int(n) * 4))
        return out[:n]
