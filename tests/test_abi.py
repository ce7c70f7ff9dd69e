"""CPU: the C-ABI library loads, exports every symbol include/bfk.h declares, fails loudly without a GPU,
and its host-side tokeniser/vocabulary (bfk_build_csr) reproduces the reference CSR."""

import ctypes as C
import re

import numpy as np
import pytest
from conftest import ROOT, load_stage, stage_names

from breakfast_amd import _lib


def declared_symbols():
    txt = (ROOT / "include" / "bfk.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bfk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(str(_lib.LIB_PATH))
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libbfk.so does not export {s}"
    assert set(syms) == set(_lib.EXPORTS), "python binding table and bfk.h disagree"
    assert _lib.load().bfk_abi_version() == _lib.ABI_VERSION == 3


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    assert lib.bfk_device_count() == 0
    with pytest.raises(_lib.BfkError) as ei:
        _lib.cluster_csr(np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32), 1)
    assert ei.value.code == -3  # BFK_ENODEV: no CPU fallback
    with pytest.raises(_lib.BfkError):
        _lib.Context(0)


@pytest.mark.parametrize("name", stage_names())
def test_build_csr_matches_reference(name):
    g = load_stage(name)
    indptr, indices, nv = _lib.build_csr(g["ufeatures"], g["sep"])
    assert np.array_equal(indptr, g["indptr"])
    assert np.array_equal(indices, g["indices"])
    assert nv == int(g["n_vocab"])


def test_build_csr_kats(kats):
    for c in kats["cluster"]:
        if "error" in c:
            continue
        uf = list(dict.fromkeys(c["features"]))
        indptr, indices, _ = _lib.build_csr(uf, c["sep"])
        assert indptr.tolist() == c["indptr"] and indices.tolist() == c["indices"], c


def test_build_csr_edge_cases():
    indptr, indices, nv = _lib.build_csr(["", "C241T"], " ")  # reference tests/test_filtering.py:94-99
    assert indptr.tolist() == [0, 0, 1] and indices.tolist() == [0] and nv == 1
    indptr, indices, nv = _lib.build_csr([], " ")
    assert indptr.tolist() == [0] and len(indices) == 0 and nv == 0
    indptr, indices, nv = _lib.build_csr(["a||b||||a", float("nan"), "||"], "||")
    assert indptr.tolist() == [0, 3, 3, 3] and indices.tolist() == [0, 1, 0] and nv == 2
    with pytest.raises(ValueError):
        _lib.build_csr(["a b"], "")
    big = " ".join(f"T{i}" for i in range(200000))  # forces hash-table growth
    indptr, indices, nv = _lib.build_csr([big, big], " ")
    assert nv == 200000 and np.array_equal(indices[:200000], np.arange(200000)) and indptr[-1] == 400000


def test_library_embeds_a_gfx950_code_object():
    """the build must carry device code for MI355X (a host-only link would load fine and fail at first launch)"""
    data = _lib.LIB_PATH.read_bytes()
    assert b"amdgcn-amd-amdhsa--gfx950" in data  # the offload bundle entry of the embedded code object
    assert b"amdgcn-amd-amdhsa--gfx906" not in data  # hipcc's default when --offload-arch got lost
    assert b"k_prefilter" in data
