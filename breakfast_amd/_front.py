"""ctypes binding of libbfk_front.so — the HIP-free part of the C-ABI (include/bfk.h: bfk_table_*, bfk_preload_*,
bfk_table_cluster_write).  Imports nothing but ctypes: the CLI's native path (fastpath.py) starts the HIP library load on
a native thread through this module before anything else is imported, and never needs numpy."""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# (BFK_FRONT_LIB: another build of the library, e.g. the sanitizer build of `make -C breakfast_amd/csrc asan`)
FRONT_PATH = Path(os.environ["BFK_FRONT_LIB"]) if os.environ.get("BFK_FRONT_LIB") else _HERE / "libbfk_front.so"
LIB_PATH = Path(os.environ["BFK_LIB"]) if os.environ.get("BFK_LIB") else _HERE / "libbfk.so"
EUNSUPPORTED = -7
ENOMEM = -2
EHIP = -4
ABI_VERSION = 3  # BFK_ABI_VERSION of include/bfk.h (same number as _lib.ABI_VERSION)
VAR_TYPES = {"covsonar_dna": 0, "covsonar_aa": 1, "nextclade_dna": 2, "nextclade_aa": 3, "raw": 4}
_lib = None
_preloading = False


class FilterOpts(C.Structure):
    _fields_ = [("var_type", C.c_int32), ("skip_ins", C.c_int32), ("skip_del", C.c_int32),
                ("trim_start", C.c_int64), ("trim_end", C.c_int64), ("reference_length", C.c_int64)]


class PrepInfo(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_unique", C.c_int64), ("nnz", C.c_int64), ("n_invalid", C.c_int64),
                ("n_vocab", C.c_int32), ("filtered", C.c_int32)]


class Unsupported(Exception):
    """the native reader declined the input (BFK_EUNSUPPORTED): use the pandas path"""


class FrontError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbfk error {code}: {msg}")
        self.code = code
        self.msg = msg


def load():
    global _lib
    if _lib is None:
        if not FRONT_PATH.exists():
            raise RuntimeError(f"{FRONT_PATH} is missing: build it first (make -C breakfast_amd/csrc, or "
                               "__graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(str(FRONT_PATH))
        lib.bfk_last_error.restype = C.c_char_p
        if lib.bfk_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{FRONT_PATH} has ABI version {lib.bfk_abi_version()}, this binding expects {ABI_VERSION}: "
                               "rebuild it (make -C breakfast_amd/csrc)")
        lib.bfk_preload_join.restype = None
        lib.bfk_preload_start.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int64]
        lib.bfk_table_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.bfk_table_close.argtypes = [C.c_void_p]
        lib.bfk_table_close.restype = None
        lib.bfk_table_prepare.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.POINTER(PrepInfo)]
        lib.bfk_table_invalid.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.bfk_table_invalid_count.argtypes = [C.c_void_p]
        lib.bfk_table_invalid_count.restype = C.c_int64
        lib.bfk_table_cluster_write.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.c_int64)]
        lib.bfk_table_pipeline_device.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32,
                                                  C.c_char_p, C.POINTER(PrepInfo), C.POINTER(C.c_int64)]
        lib.bfk_table_pipeline_device_gpus.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32, C.c_int32,
                                                       C.c_char_p, C.POINTER(PrepInfo), C.POINTER(C.c_int64)]
        lib.bfk_table_pipeline_device_cache.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(FilterOpts), C.c_int32, C.c_int32,
                                                        C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(PrepInfo), C.POINTER(C.c_int64)]
        _lib = lib
    return _lib


def _err(lib):
    return lib.bfk_last_error().decode(errors="replace")


def preload_join():
    """wait for the preload thread, if one was started (a no-op otherwise)"""
    if _lib is not None:
        _lib.bfk_preload_join()


def preload(input_file=None, device: int = 0):
    """Start loading libbfk.so (HIP runtime, device context, code object, workspace sized from the input file's size) on a
    native thread.  Idempotent; errors surface at cluster time."""
    global _preloading
    lib = load()
    if not _preloading:
        # every exit path that never reaches bfk_table_cluster_write (max-dist 0, a declined input, an exception, --help)
        # joins the thread here — before interpreter teardown and the HIP runtime's own exit handlers
        import atexit

        atexit.register(lib.bfk_preload_join)
        _preloading = True
    rows = nnz = 0
    if input_file is not None:
        try:
            size = os.stat(input_file).st_size
            rows, nnz = size // 250 + 1024, size // 6 + 4096
        except OSError:
            pass
    lib.bfk_preload_start(str(LIB_PATH).encode(), int(device), int(rows), int(nnz))


class Table:
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
            raise Unsupported(_err(lib))
        if rc:
            raise FrontError(rc, _err(lib))
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

    def prepare(self, sep2: str, var_type: str, skip_ins, skip_del, trim_start, trim_end, reference_length):
        opts = FilterOpts(VAR_TYPES[var_type], int(bool(skip_ins)), int(bool(skip_del)), int(trim_start), int(trim_end),
                          int(reference_length))
        info = PrepInfo()
        sepb = sep2.encode()
        rc = self.lib.bfk_table_prepare(self.h, sepb, len(sepb), C.byref(opts), C.byref(info))
        if rc == EUNSUPPORTED:
            raise Unsupported(_err(self.lib))
        if rc:
            raise FrontError(rc, _err(self.lib))
        self.info = info
        return info

    def pipeline_device(self, sep2: str, var_type: str, skip_ins, skip_del, trim_start, trim_end, reference_length, max_dist: int,
                        min_cluster_size: int, path, n_gpus: int = 1, cache_path=None, in_cache=None):
        """filter + collapse + CSR + clustering on the device and the writer, one native call (bfk_table_pipeline_device[_gpus]:
        with n_gpus > 1 the unique rows are clustered on several devices where that pays; with cache_path every edge is recorded
        and a side-car cache of the run is written there, with in_cache an exact side-car of this max_dist is checked against the input
        (every cached row still there: the run equals the cache run) — bfk_table_pipeline_device_cache, one device);
        -> (PrepInfo, clusters written); Unsupported: the device stages decline the input, nothing was done"""
        opts = FilterOpts(VAR_TYPES[var_type], int(bool(skip_ins)), int(bool(skip_del)), int(trim_start), int(trim_end),
                          int(reference_length))
        info, n = PrepInfo(), C.c_int64()
        sepb = sep2.encode()
        if cache_path is not None or in_cache is not None:
            rc = self.lib.bfk_table_pipeline_device_cache(self.h, sepb, len(sepb), C.byref(opts), int(max_dist), int(min_cluster_size),
                                                          str(path).encode(), None if in_cache is None else str(in_cache).encode(),
                                                          None if cache_path is None else str(cache_path).encode(), C.byref(info), C.byref(n))
        elif n_gpus > 1:
            rc = self.lib.bfk_table_pipeline_device_gpus(self.h, sepb, len(sepb), C.byref(opts), int(max_dist), int(min_cluster_size),
                                                         int(n_gpus), str(path).encode(), C.byref(info), C.byref(n))
        else:
            rc = self.lib.bfk_table_pipeline_device(self.h, sepb, len(sepb), C.byref(opts), int(max_dist), int(min_cluster_size),
                                                    str(path).encode(), C.byref(info), C.byref(n))
        if rc == EUNSUPPORTED:
            raise Unsupported(_err(self.lib))
        if rc:
            raise FrontError(rc, _err(self.lib))
        self.info = info
        return info, int(n.value)

    def invalid_count(self) -> int:
        return int(self.lib.bfk_table_invalid_count(self.h))

    def invalid(self, i):
        p, n = C.c_void_p(), C.c_int64()
        rc = self.lib.bfk_table_invalid(self.h, int(i), C.byref(p), C.byref(n))
        if rc:
            raise FrontError(rc, _err(self.lib))
        return C.string_at(p, n.value).decode("utf-8")

    def cluster_write(self, max_dist: int, min_cluster_size: int, path, n_gpus: int = 1) -> int:
        n = C.c_int64()
        rc = self.lib.bfk_table_cluster_write(self.h, int(max_dist), int(min_cluster_size), int(n_gpus), str(path).encode(),
                                              C.byref(n))
        if rc:
            raise FrontError(rc, _err(self.lib))
        return int(n.value)
