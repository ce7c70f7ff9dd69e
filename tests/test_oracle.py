"""Pins the CPU oracle (oracle/bfk_oracle.c + oracle/ref_port.py) against vectors produced by the
imported reference (tests/golden/, tools/make_golden.py).  CPU only."""

import hashlib
import json

import numpy as np
import pytest
from conftest import GOLD, load_stage, stage_names

from oracle import ref_port as orc


@pytest.mark.parametrize("name", stage_names())
def test_csr_matches_reference(name):
    g = load_stage(name)
    indptr, indices, nv = orc.sparse_feature_matrix(g["ufeatures"], g["sep"])
    assert np.array_equal(indptr, g["indptr"])
    assert np.array_equal(indices, g["indices"])  # same first-appearance vocabulary numbering
    assert nv == int(g["n_vocab"])


@pytest.mark.parametrize("name", stage_names())
def test_neighbour_lists_match_reference_exactly(name):
    g = load_stage(name)
    res = orc.cluster_csr(g["indptr"], g["indices"], g["max_dist"], want_neigh=True, n_threads=2)
    assert np.array_equal(res["neigh_off"], g["neigh_off"])
    assert np.array_equal(res["neigh_flat"], g["neigh_flat"])
    assert np.array_equal(res["labels"], g["labels"])


@pytest.mark.parametrize("name", stage_names())
def test_cluster_ids_and_tsv_match_reference(name):
    g = load_stage(name)
    res = orc.cluster_features(g["ufeatures"], g["group_size"], g["sep"], g["max_dist"], g["min_cluster_size"])
    assert np.array_equal(res["n_features"], g["n_features"])
    assert np.array_equal(res["cluster_id"], g["cluster_id"])
    ids = [f"seq{i:07d}" for i in range(len(g["features"]))]
    out = orc.pipeline_bytes(ids, g["features"], g["sep"], g["max_dist"], g["min_cluster_size"])
    assert out == g["clusters_tsv"]


@pytest.mark.parametrize("name", ["syn200_d1", "multiset300_d2", "longrows_d1"])
def test_get_neighbours_batch_per_band(name):
    """band-by-band: the concatenation over distinct lengths is the stored list; also the edge set."""
    g = load_stage(name)
    nf = g["n_features"]
    lists = []
    for q in dict.fromkeys(nf.tolist()):
        lists += orc.get_neighbours_batch(g["indptr"], g["indices"], nf, q, g["max_dist"])
    flat = np.concatenate(lists) if lists else np.zeros(0, np.int64)
    assert np.array_equal(flat, g["neigh_flat"])
    # every stored edge (exact distances from sklearn) shows up as co-membership of a list whose query row is i
    edges = {tuple(e) for e in g["edges"].tolist()}
    seen = set()
    for q in dict.fromkeys(nf.tolist()):
        qrows = np.flatnonzero(np.abs(nf - q) <= g["max_dist"])
        nb = orc.get_neighbours_batch(g["indptr"], g["indices"], nf, q, g["max_dist"])
        assert len(nb) == len(qrows)
        for i, l in zip(qrows, nb):
            assert i in l  # self
            seen |= {(min(i, j), max(i, j)) for j in l.tolist() if j != i}
    assert seen == edges


@pytest.mark.parametrize("n,d,p_del,p_ins", [(1500, 1, 0.0, 0.0), (1500, 2, 0.0, 0.0), (800, 5, 0.05, 0.01)])
def test_c_restatement_against_the_sklearn_kernel_the_reference_calls(n, d, p_del, p_ins):
    """oracle/sk_port.py makes the reference's own third-party call (pairwise_distances_chunked, manhattan, on the
    length band) — the C restatement must find the same neighbours row by row, do the same number of row merges,
    and the components must be the components of that graph (seeded inputs beyond the committed fixtures)"""
    from oracle import sk_port

    if not sk_port.available():
        pytest.skip("scikit-learn / scipy not importable")
    from breakfast_amd.synth import generate_profiles

    rows = list(dict.fromkeys(generate_profiles(n, p_del=p_del, p_ins=p_ins, seed=99 + d)))
    indptr, indices, nv = orc.sparse_feature_matrix(rows, " ")
    nf = np.diff(indptr)
    want, _ = sk_port.neighbours(indptr, indices, d, n_vocab=nv)
    got = [set() for _ in rows]
    for q in dict.fromkeys(nf.tolist()):
        qrows = np.flatnonzero(np.abs(nf - q) <= d)
        for i, l in zip(qrows, orc.get_neighbours_batch(indptr, indices, nf, q, d)):
            got[i] |= set(l.tolist())
    assert all(sorted(g) == w.tolist() for g, w in zip(got, want))
    res = orc.cluster_csr(indptr, indices, d)
    assert res["n_merges"] == sk_port.merges(indptr, d)
    # components of the sklearn graph by a plain label propagation
    lab = np.arange(len(rows))
    changed = True
    while changed:
        changed = False
        for i, w in enumerate(want):
            m = lab[w].min()
            if (lab[w] != m).any():
                lab[w] = m
                changed = True
    assert np.array_equal(res["labels"], lab)


def test_select_ind_path():
    g = load_stage("syn200_d1")
    nf = g["n_features"]
    sel = np.array([5, 3, 100, 42], dtype=np.int64)
    q = int(nf[3])
    full = orc.get_neighbours_batch(g["indptr"], g["indices"], nf, q, 1)
    qrows = np.flatnonzero(np.abs(nf - q) <= 1).tolist()
    sub = orc.get_neighbours_batch(g["indptr"], g["indices"], nf, q, 1, select_ind=sel)
    want = [full[qrows.index(i)] for i in sel.tolist() if i in qrows]
    assert len(sub) == len(want) and all(np.array_equal(a, b) for a, b in zip(sub, want))
    assert orc.get_neighbours_batch(g["indptr"], g["indices"], nf, q, 1, select_ind=np.zeros(0, np.int64)) == []


def test_cluster_kats(kats):
    for c in kats["cluster"]:
        ids = [f"s{i}" for i in range(len(c["features"]))]
        if "error" in c:
            with pytest.raises(ValueError):
                orc.pipeline_bytes(ids, c["features"], c["sep"], c["max_dist"], c["min_cluster_size"])
            continue
        out = orc.pipeline_bytes(ids, c["features"], c["sep"], c["max_dist"], c["min_cluster_size"])
        assert out.decode() == c["clusters_tsv"], c
        uf, _ = orc.collapse(c["features"])
        indptr, indices, _ = orc.sparse_feature_matrix(uf, c["sep"])
        assert indptr.tolist() == c["indptr"] and indices.tolist() == c["indices"]
        assert np.diff(indptr).tolist() == c["row_sums"]


def test_sha256_2k():
    """App. A 2k workload end to end (features are inside the default trims, so filtering is the identity)."""
    from breakfast_amd.synth import generate_profiles

    sha = json.loads((GOLD / "sha256.json").read_text())["syn2000_d1"]
    rows = generate_profiles(2000)
    ids = [f"seq{i:07d}" for i in range(2000)]
    out = orc.pipeline_bytes(ids, rows, " ", 1, 2, n_threads=4)
    assert hashlib.sha256(out).hexdigest() == sha["clusters_sha256"]
