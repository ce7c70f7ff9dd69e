"""GPU parity: the HIP path, called through the C-ABI (libbfk.so), against (a) the golden vectors produced
by the imported reference, (b) the CPU oracle on seeded inputs, (c) size-independent properties at the
BASELINE.json sizes.  Integer work: the bar is bit-exact."""

import hashlib
import io
import json
import os
from contextlib import redirect_stdout

import click.testing
import numpy as np
import pandas as pd
import pytest
from conftest import GOLD, load_stage, stage_names, set_generator

from breakfast_amd import fastpath, _lib, breakfast, console
from breakfast_amd.synth import generate_profiles
from oracle import ref_port as orc

pytestmark = pytest.mark.gpu
ORACLE_THREADS = min(16, os.cpu_count() or 8)  # (the oracle's OpenMP threads: a GPU box gives a test run 16 cores)
FIX = GOLD / "ref_fixtures"


def test_device_present():
    assert _lib.load().bfk_device_count() >= 1


@pytest.mark.exact_edges
@pytest.mark.parametrize("name", stage_names())
def test_labels_match_reference_golden(name):
    g = load_stage(name)
    labels, st = _lib.cluster_csr(g["indptr"], g["indices"], g["max_dist"])
    assert np.array_equal(labels, g["labels"])
    assert st["n_edges"] == len(g["edges"])  # every reference edge, and only those, passed the exact merge
    assert st["n_retry_slices"] == 0


@pytest.mark.parametrize("name", stage_names())
def test_neighbour_lists_match_reference_golden(name):
    g = load_stage(name)
    ptr, idx = _lib.neighbours_csr(g["indptr"], g["indices"], g["max_dist"])
    n = len(g["indptr"]) - 1
    got = {(i, int(j)) for i in range(n) for j in idx[ptr[i]: ptr[i + 1]] if j > i}
    assert got == {tuple(e) for e in g["edges"].tolist()}
    for i in range(n):
        l = idx[ptr[i]: ptr[i + 1]]
        assert i in l and np.all(np.diff(l) > 0)


@pytest.mark.parametrize("name", ["syn200_d1", "multiset300_d2", "indel200_d5", "longrows_d2"])
def test_get_neighbours_batch_mirror_matches_reference_order(name):
    """band by band through the mirror API: exactly the list the reference builds"""
    g = load_stage(name)
    nf = g["n_features"]
    lists = []
    for q in dict.fromkeys(nf.tolist()):
        lists += breakfast.get_neighbours_batch((g["indptr"], g["indices"]), nf, q, g["max_dist"])
    flat = np.concatenate(lists)
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    assert np.array_equal(off, g["neigh_off"]) and np.array_equal(flat, g["neigh_flat"])


@pytest.mark.parametrize("name", stage_names())
def test_clusters_tsv_bytes_from_features(name, tmp_path):
    """feature strings -> clusters.tsv through collapse + cluster (GPU) + write_output"""
    g = load_stage(name)
    meta = pd.DataFrame({"id": [f"seq{i:07d}" for i in range(len(g["features"]))], "feature": g["features"]})
    with redirect_stdout(io.StringIO()):
        nod = breakfast.collapse_duplicates(meta)
        cl = breakfast.cluster(nod, g["sep"], g["max_dist"], g["min_cluster_size"], None, None)
        breakfast.write_output(cl, meta, tmp_path)
    assert (tmp_path / "clusters.tsv").read_bytes() == g["clusters_tsv"]
    assert np.array_equal(np.asarray(cl["n_features"]), g["n_features"])


def test_cluster_kats(kats, tmp_path):
    for n, c in enumerate(kats["cluster"]):
        meta = pd.DataFrame({"id": [f"s{i}" for i in range(len(c["features"]))], "feature": c["features"]})
        with redirect_stdout(io.StringIO()):
            nod = breakfast.collapse_duplicates(meta)
            if "error" in c:
                with pytest.raises(ValueError):
                    breakfast.cluster(nod, c["sep"], c["max_dist"], c["min_cluster_size"], None, None)
                continue
            cl = breakfast.cluster(nod, c["sep"], c["max_dist"], c["min_cluster_size"], None, None)
            out = tmp_path / str(n)
            breakfast.write_output(cl, meta, out)
        assert (out / "clusters.tsv").read_text() == c["clusters_tsv"], c


@pytest.mark.parametrize("scenario", ["dist0", "dist1", "dist1_noskipdel", "raw_defaults", "raw_explicit",
                                      "nextclade_dist0", "nextclade_dist1", "dist2_mcs3", "dist1_mcs3_noskip"])
@pytest.mark.parametrize("path", ["native", "pandas"])
def test_cli_matches_reference_bytes(scenario, path, cli_runs, tmp_path, monkeypatch):
    """the reference's own CLI scenarios (tests/test_breakfast.py) — byte-identical clusters.tsv, through the
    native front end / writer (fastpath.py) and through the pandas mirror of the reference's functions"""
    monkeypatch.chdir(FIX)
    run = cli_runs[scenario]
    taken = []
    real = fastpath.run
    monkeypatch.setattr(fastpath, "run", lambda *a, **k: taken.append(real(*a, **k)) or taken[-1])
    if path == "pandas":
        monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    res = click.testing.CliRunner().invoke(console.main, run["args"] + ["--outdir", str(tmp_path)])
    assert res.exit_code == 0, res.output
    assert taken == ([True] if path == "native" else [])
    data = (tmp_path / "clusters.tsv").read_bytes()
    assert data.decode() == run["clusters_tsv"]
    assert hashlib.sha256(data).hexdigest() == run["sha256"]


@pytest.mark.parametrize("fixture,expected", [
    (["--input-file", "testfile.tsv", "--max-dist", "1"], "expected_clusters_dist1.tsv"),
    (["--input-file", "testfile.tsv", "--max-dist", "1", "--no-skip-del"], "expected_clusters_dist1_noskipdel.tsv"),
    (["--input-file", "testfile.tsv", "--var-type", "raw"], "expected_clusters_dist1_noskipdel.tsv"),
])
def test_cli_like_reference_tests(fixture, expected, tmp_path, monkeypatch):
    monkeypatch.chdir(FIX)
    res = click.testing.CliRunner().invoke(console.main, fixture + ["--outdir", str(tmp_path)])
    assert res.exit_code == 0
    assert pd.read_table(expected, sep="\t").equals(pd.read_table(tmp_path / "clusters.tsv", sep="\t"))


# ---- seeded inputs vs the oracle ------------------------------------------------------------------
def _random_multisets(n, seed, alphabet, kmax, p_empty=0.05):
    rng = np.random.default_rng(seed)
    base = [rng.integers(0, alphabet, size=int(rng.integers(0, kmax + 1))) for _ in range(max(2, n // 8))]
    rows = []
    for _ in range(n):
        r = list(base[int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, 4))):
            op = rng.random()
            if op < 0.45 and r:
                r.pop(int(rng.integers(0, len(r))))
            elif op < 0.9:
                r.append(int(rng.integers(0, alphabet)))
            elif r:
                r.append(r[int(rng.integers(0, len(r)))])
        if rng.random() < p_empty:
            r = []
        rng.shuffle(r)
        rows.append(np.array(r, dtype=np.int32))
    indptr = np.zeros(n + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows).astype(np.int32) if indptr[-1] else np.zeros(0, np.int32)
    return indptr, indices


@pytest.mark.parametrize("n,alphabet,kmax,d,seed", [
    (1, 5, 3, 1, 0), (2, 5, 3, 1, 1), (7, 4, 3, 2, 2), (500, 40, 10, 1, 3), (500, 40, 10, 3, 4),
    (1500, 200, 40, 2, 5), (1500, 30, 6, 1, 6), (800, 1000, 300, 5, 7), (600, 3000, 700, 7, 8),
    (400, 8, 90, 4, 9), (1200, 100, 70, 12, 10), (300, 60, 90, 70, 11), (900, 12, 5, 2, 12),
    (2000, 4000, 30, 1, 13),
])
def test_random_multisets_vs_oracle(n, alphabet, kmax, d, seed):
    indptr, indices = _random_multisets(n, seed, alphabet, kmax)
    if indptr[-1] == 0:
        indices = np.zeros(0, np.int32)
    want = orc.cluster_csr(indptr, indices, d, n_threads=ORACLE_THREADS)["labels"]
    got, st = _lib.cluster_csr(indptr, indices, d)
    assert np.array_equal(got, want)
    assert st["sig_words"] == (1 if d <= 2 else 2 if d <= 5 else 4)


def test_hub_and_spokes_dense_cells():
    """a root profile with thousands of direct children (all in four (k,f,g) cells) plus grandchildren: the hub
    row has thousands of neighbours, the children's cells are dense — exercises the hit queue drain and the
    tile splitting across waves"""
    rng = np.random.default_rng(3)
    root = rng.choice(50000, size=30, replace=False)
    rows = [root]
    for _ in range(6000):
        rows.append(np.append(root, rng.integers(50000, 90000)))
    for i in range(1, 1500):
        rows.append(np.append(rows[i], rng.integers(90000, 120000)))
    order = rng.permutation(len(rows))
    rows = [rng.permutation(rows[i]).astype(np.int32) for i in order]
    indptr = np.zeros(len(rows) + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows)
    for d in (1, 2):
        want = orc.cluster_csr(indptr, indices, d, n_threads=16)["labels"]
        got, st = _lib.cluster_csr(indptr, indices, d)
        assert np.array_equal(got, want)
    ptr, idx = _lib.neighbours_csr(indptr, indices, 1, np.array([int(np.flatnonzero(order == 0)[0])], np.int64))
    assert ptr[1] - ptr[0] == 6001  # the root sees itself and its 6000 children


def test_very_long_rows_use_the_global_table():
    """rows of ~5000 tokens: pairs exceed the LDS table of k_verify_long and use the global scratch table"""
    rng = np.random.default_rng(21)
    base = rng.choice(200000, size=5000, replace=False).astype(np.int32)
    rows = [base, np.append(base, 777777).astype(np.int32), np.delete(base, 10), rng.permutation(base),
            np.append(np.delete(base, [1, 2, 3]), [888888, 888889]).astype(np.int32),
            rng.choice(200000, size=4000, replace=False).astype(np.int32), np.array([5, 6, 7], np.int32)]
    indptr = np.zeros(len(rows) + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows)
    for d in (1, 2, 5):
        want = orc.cluster_csr(indptr, indices, d, n_threads=4)["labels"]
        got, _ = _lib.cluster_csr(indptr, indices, d)
        assert np.array_equal(got, want), d


@pytest.mark.exact_edges
def test_all_identical_rows_clique():
    """every pair is within distance 0: a dense candidate set (queue pressure) must still be exact"""
    n = 3000
    rng = np.random.default_rng(5)
    row = rng.permutation(40).astype(np.int32)
    indices = np.concatenate([rng.permutation(row) for _ in range(n)]).astype(np.int32)
    indptr = (np.arange(n + 1) * 40).astype(np.int32)
    got, st = _lib.cluster_csr(indptr, indices, 1)
    assert np.all(got == 0)
    # every one of the n(n-1)/2 pairs is an edge and is counted once — whichever path served the step (the variant join gives
    # up on thousands of equal multisets and the step is redone on the band kernels, in slices if the queue is too small)
    assert st["n_edges"] == n * (n - 1) // 2
    got3, st3 = _lib.cluster_csr(indptr, indices, 3)
    assert np.all(got3 == 0) and st3["n_edges"] == n * (n - 1) // 2


def test_empty_and_degenerate_inputs():
    got, _ = _lib.cluster_csr(np.array([0], np.int32), np.zeros(0, np.int32), 1)
    assert len(got) == 0
    got, _ = _lib.cluster_csr(np.array([0, 0, 0, 0], np.int32), np.zeros(0, np.int32), 1)
    assert got.tolist() == [0, 0, 0]  # three empty rows: distance 0
    got, _ = _lib.cluster_csr(np.array([0, 0, 2, 3], np.int32), np.array([5, 6, 5], np.int32), 1)
    assert got.tolist() == [0, 0, 0]  # {} -1- {5} -1- {5,6}
    got, _ = _lib.cluster_csr(np.array([0, 0, 2, 5], np.int32), np.array([5, 6, 7, 8, 9], np.int32), 1)
    assert got.tolist() == [0, 1, 2]
    with pytest.raises(_lib.BfkError):
        _lib.cluster_csr(np.array([0, 2, 1], np.int32), np.array([1, 2], np.int32), 1)  # indptr not monotone


def test_run_to_run_determinism_of_labels():
    rows = generate_profiles(5000)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    a, _ = _lib.cluster_csr(indptr, indices, 2)
    for _ in range(3):
        b, _ = _lib.cluster_csr(indptr, indices, 2)
        assert np.array_equal(a, b)


# ---- BASELINE.json sizes ---------------------------------------------------------------------------
def _cli_sha(n, tmp_path, extra=()):
    from breakfast_amd.synth import generate_tsv

    inp = tmp_path / "in.tsv"
    generate_tsv(inp, n)
    res = click.testing.CliRunner().invoke(console.main, ["--input-file", str(inp), "--outdir", str(tmp_path),
                                                          "--max-dist", "1", *extra])
    assert res.exit_code == 0, res.output
    return hashlib.sha256(inp.read_bytes()).hexdigest(), hashlib.sha256((tmp_path / "clusters.tsv").read_bytes()).hexdigest()


@pytest.mark.parametrize("path", ["native", "pandas"])
@pytest.mark.parametrize("key", ["syn2000_d1", "syn10000_d1", "syn100000_d1"])
def test_clusters_tsv_sha256_at_baseline_sizes(key, path, tmp_path, monkeypatch):
    """configs[1] (10k) and configs[2] (100k): the whole CLI, digest recorded from the reference"""
    if path == "pandas":
        monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    want = json.loads((GOLD / "sha256.json").read_text())[key]
    h_in, h_out = _cli_sha(want["n"], tmp_path)
    assert h_in == want["input_sha256"]
    assert h_out == want["clusters_sha256"]


def test_10k_vs_oracle_full():
    rows = list(dict.fromkeys(generate_profiles(10000)))
    indptr, indices, _ = _lib.build_csr(rows, " ")
    want = orc.cluster_csr(indptr, indices, 1, n_threads=16)
    got, st = _lib.cluster_csr(indptr, indices, 1)
    assert np.array_equal(got, want["labels"])
    assert st["pairs_resolved"] == len(rows) * (len(rows) - 1) // 2


def test_properties_at_100k_d2_and_indels():
    """size-independent properties where the oracle would take too long: idempotence under row permutation
    (labels are canonical), monotonicity in d (partition at d refines partition at d+1), labels are fix points"""
    rows = list(dict.fromkeys(generate_profiles(100000, p_del=0.05, p_ins=0.01)))
    indptr, indices, _ = _lib.build_csr(rows, " ")
    l1, _ = _lib.cluster_csr(indptr, indices, 1)
    l2, st2 = _lib.cluster_csr(indptr, indices, 2)
    assert np.array_equal(l1[l1], l1) and np.array_equal(l2[l2], l2) and np.all(l1 <= np.arange(len(l1)))
    assert np.array_equal(l2[l1], l2)  # same d=1 component -> same d=2 component
    # permute rows: the partition must be the same
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(rows))
    ip2, ix2, _ = _lib.build_csr([rows[i] for i in perm], " ")
    lp, _ = _lib.cluster_csr(ip2, ix2, 2)
    back = np.empty(len(rows), np.int64)
    back[perm] = np.arange(len(rows))           # original row -> permuted row
    canon = np.full(len(rows), len(rows), np.int64)
    np.minimum.at(canon, lp[back], np.arange(len(rows)))  # min original index per permuted-run component
    assert np.array_equal(canon[lp[back]], l2)
    # sampled rows against the oracle's select_ind path (rows x all columns)
    sel = np.sort(rng.choice(len(rows), size=300, replace=False)).astype(np.int64)
    ptr, idx = _lib.neighbours_csr(indptr, indices, 2, sel)
    nf = np.diff(indptr).astype(np.int64)
    for s, i in enumerate(sel[:40].tolist()):
        want = set()
        for q in range(int(nf[i]) - 2, int(nf[i]) + 3):
            if q < 0:
                continue
            for l in orc.get_neighbours_batch(indptr, indices, nf, q, 2, select_ind=np.array([i], np.int64),
                                              n_threads=16):
                want |= set(l.tolist())
        assert set(idx[ptr[s]: ptr[s + 1]].tolist()) == want


# ---- incremental cache (reference tests/test_caching.py) --------------------------------------------
CACHE_INPUTS = ["AddedSeqs", "DisorderedOnly", "AddedAndDisorderedSeqs", "DeletedSingleSeq", "DeletedProfile",
                "ModifiedSeqs", "MultipleTests", "NoChanges"]


def _run_cli(args):
    res = click.testing.CliRunner().invoke(console.main, args)
    assert res.exit_code == 0, (res.output, res.exception)
    return res


@pytest.fixture(params=["native", "pandas"])
def own_cache(request, tmp_path, monkeypatch):
    """a cache written by this build: through the native host stages (fastpath._run_with_cache) and through the pandas mirror"""
    if request.param == "pandas":
        monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    monkeypatch.chdir(FIX)
    cache = tmp_path / "cache_dir" / "cache"  # parent directory is created on demand (test_caching.py:106-125)
    _run_cli(["--input-file", "testfile.tsv", "--outdir", str(tmp_path / "init"), "--output-cache", str(cache),
              "--max-dist", "1"])
    assert cache.exists()
    assert (tmp_path / "init" / "clusters.tsv").read_text() == json.loads(
        (GOLD / "cli_runs.json").read_text())["dist1"]["clusters_tsv"]
    monkeypatch.delenv("BFK_NO_FASTPATH", raising=False)
    return cache


@pytest.mark.parametrize("reader", ["native", "pandas"])
@pytest.mark.parametrize("which", ["own", "reference"])
@pytest.mark.parametrize("idx,name", list(enumerate(CACHE_INPUTS, 1)))
def test_cache_scenarios(idx, name, which, reader, own_cache, cli_runs, tmp_path, monkeypatch):
    """the 8 cache scenarios of the reference, with a cache written by this build (either host path) and with one written
    by the reference itself (same pickle format), read through either host path: clusters.tsv equals what the reference
    produced"""
    if reader == "pandas":
        monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    monkeypatch.chdir(FIX)
    cache = own_cache if which == "own" else GOLD / "ref_cache_testfile_d1.pkl.gz"
    inp = f"testfile_caching{idx:02d}_{name}.tsv"
    _run_cli(["--input-file", inp, "--outdir", str(tmp_path / "out"), "--input-cache", str(cache), "--max-dist", "1"])
    got = (tmp_path / "out" / "clusters.tsv").read_text()
    assert got == cli_runs[f"cache_caching{idx:02d}"]["clusters_tsv"]
    exp = pd.read_table(f"expected_clusters_caching{idx:02d}_dist1.tsv", sep="\t")
    assert exp.equals(pd.read_table(tmp_path / "out" / "clusters.tsv", sep="\t"))


def test_cache_content_matches_reference(own_cache, cli_runs):
    import gzip
    import pickle

    with gzip.open(own_cache, "rb") as f:
        c = pickle.load(f)
    ref = cli_runs["cache_init"]
    assert c["max_dist"] == ref["max_dist"]
    assert [list(map(int, x)) for x in c["neigh"]] == ref["neigh"]  # same lists in the same order
    assert [list(t) for t in c["meta"]["id"]] == ref["meta_id"] and list(c["meta"]["feature"]) == ref["meta_feature"]


def test_cache_other_max_dist_falls_back(own_cache, cli_runs, tmp_path, monkeypatch):
    monkeypatch.chdir(FIX)
    res = _run_cli(["--input-file", "testfile.tsv", "--outdir", str(tmp_path / "o"), "--input-cache", str(own_cache),
                    "--max-dist", "2", "--min-cluster-size", "3"])
    assert "differnt max-dist" in res.output
    assert (tmp_path / "o" / "clusters.tsv").read_text() == cli_runs["dist2_mcs3"]["clusters_tsv"]


def test_labels_from_lists():
    lab = _lib.labels_from_lists(7, [np.array([3, 1]), np.array([5]), np.array([6, 2, 1])])
    assert lab.tolist() == [0, 1, 1, 1, 4, 5, 1]
    assert _lib.labels_from_lists(3, []).tolist() == [0, 1, 2]
    with pytest.raises(_lib.BfkError):
        _lib.labels_from_lists(3, [np.array([0, 9])])


def test_cache_incremental_on_synthetic(tmp_path):
    """grow a 3000-profile input to 4000 through the cache: same clusters.tsv as a fresh run"""
    from breakfast_amd.synth import generate_profiles

    rows = generate_profiles(4000)

    def write(path, n):
        with open(path, "w") as f:
            f.write("accession\tdna_profile\n")
            for i in range(n):
                f.write(f"seq{i:07d}\t{rows[i]}\n")

    write(tmp_path / "a.tsv", 3000)
    write(tmp_path / "b.tsv", 4000)
    cache = tmp_path / "c.pkl.gz"
    _run_cli(["--input-file", str(tmp_path / "a.tsv"), "--outdir", str(tmp_path / "oa"), "--output-cache", str(cache)])
    _run_cli(["--input-file", str(tmp_path / "b.tsv"), "--outdir", str(tmp_path / "ob"), "--input-cache", str(cache)])
    _run_cli(["--input-file", str(tmp_path / "b.tsv"), "--outdir", str(tmp_path / "fresh")])
    assert (tmp_path / "ob" / "clusters.tsv").read_bytes() == (tmp_path / "fresh" / "clusters.tsv").read_bytes()


def test_stats_pairs_in_band_matches_the_reference_band():
    """bfk_stats.pairs_in_band = unordered pairs with |k_i - k_j| <= d, the pairs the reference merges (:250)"""
    rows = generate_profiles(3000, p_del=0.05, p_ins=0.02)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    k = np.diff(indptr).astype(np.int64)
    cnt = np.bincount(k)
    for d in (1, 3):
        want = int(sum(cnt[a] * (cnt[a] - 1) // 2 + sum(cnt[a] * cnt[a + dl] for dl in range(1, d + 1) if a + dl < len(cnt))
                       for a in range(len(cnt))))
        _, st = _lib.cluster_csr(indptr, indices, d)
        assert st["pairs_in_band"] == want
        assert st["pairs_resolved"] == len(uf) * (len(uf) - 1) // 2


@pytest.mark.exact_edges
@pytest.mark.parametrize("n_shards,d", [(2, 1), (3, 2), (8, 1)])
def test_sharded_runs_merge_to_the_single_gpu_labels(n_shards, d):
    """the N-GPU path on one GPU: every shard of the tile list clustered into its own local forest
    (bfk_ctx_cluster(shard, n_shards)), the label arrays merged with bfk_ctx_merge_labels like after an
    all_gather — must give the 1-shard labels bit for bit"""
    rows = generate_profiles(20000, p_del=0.03, p_ins=0.01)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    n = len(uf)
    want, _ = _lib.cluster_csr(indptr, indices, d)
    ctx = _lib.Context(0)
    ctx.upload_csr(indptr, indices)
    d_local = ctx.alloc(4 * n)
    d_gath = ctx.alloc(4 * n * n_shards)
    d_out = ctx.alloc(4 * n)
    edges = 0
    for s in range(n_shards):  # shard s last => the forest in the context is shard n_shards-1's
        ctx.cluster(d, d_gath + 4 * n * s, s, n_shards)
        st = ctx.sync()
        edges += st["n_edges"]
    ctx.merge_labels(d_gath, n_shards, d_out)
    ctx.sync()
    got = ctx.download_i32(d_out, n)
    ctx.close()
    assert np.array_equal(got, want)
    _, st1 = _lib.cluster_csr(indptr, indices, d)
    assert edges == st1["n_edges"]  # every edge found by exactly one shard


@pytest.mark.parametrize("merge_all", [False, True])
@pytest.mark.parametrize("n_shards,d", [(2, 5), (4, 3), (8, 5)])
def test_dense_sharded_runs_merge_to_the_single_gpu_labels(n_shards, d, merge_all, monkeypatch):
    """the merge on DENSE forests (a few components, every rank with a giant root of its own): k_merge unites (own root,
    other label) from the rank's flat part, neighbouring lanes with one pair share the union; BFK_MERGE_ALL=1 takes the own
    part like any other (pairs (row, label))"""
    if merge_all:
        monkeypatch.setenv("BFK_MERGE_ALL", "1")
    rows = generate_profiles(30000, p_del=0.05, p_ins=0.01)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    n = len(uf)
    want, _ = _lib.cluster_csr(indptr, indices, d)
    assert len(np.unique(want)) < n // 20  # dense: few components
    ctx = _lib.Context(0)
    ctx.upload_csr(indptr, indices)
    d_gath = ctx.alloc(4 * n * n_shards)
    d_out = ctx.alloc(4 * n)
    for s in list(range(1, n_shards)) + [0]:  # shard 0 last => the forest in the context is shard 0's
        ctx.cluster(d, d_gath + 4 * n * s, s, n_shards)
        ctx.sync()
    ctx.merge_labels(d_gath, n_shards, d_out)
    ctx.sync()
    got = ctx.download_i32(d_out, n)
    ctx.close()
    assert np.array_equal(got, want)


def test_merge_rejects_labels_out_of_range():
    rows = list(dict.fromkeys(generate_profiles(2000)))
    indptr, indices, _ = _lib.build_csr(rows, " ")
    n = len(rows)
    for bad_part in (0, 1):  # the rank's own part (its roots) / another rank's
        ctx = _lib.Context(0)
        ctx.upload_csr(indptr, indices)
        d_gath = ctx.alloc(4 * n * 2)
        d_out = ctx.alloc(4 * n)
        ctx.cluster(1, d_gath, 0, 2)
        ctx.sync()
        parts = np.tile(ctx.download_i32(d_gath, n), 2).astype(np.int32)
        parts[bad_part * n + 7] = n + 5
        ctx.upload_i32(parts, d_gath)
        ctx.merge_labels(d_gath, 2, d_out)
        with pytest.raises(_lib.BfkError):
            ctx.sync()
        ctx.close()


@pytest.mark.exact_edges
@pytest.mark.parametrize("seed,d", [(1, 1), (2, 2), (3, 4), (4, 6)])
def test_order_consistent_rows_with_repeats_vs_oracle(seed, d):
    """the verify kernel's certificate path: rows that keep a common token order (like real profiles), with
    repeated tokens, insertions at both ends and in the middle, substitutions, and length differences of either
    sign — the prefix / shifted-suffix matching must never accept a pair the exact count rejects"""
    rng = np.random.default_rng(seed)
    base = [np.sort(rng.integers(0, 60, size=int(rng.integers(5, 50)))) for _ in range(40)]  # repeats likely
    rows = []
    for _ in range(3000):
        r = list(base[int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, d + 2))):
            op = rng.random()
            pos = int(rng.integers(0, len(r) + 1))
            if op < 0.4 and r:
                r.pop(min(pos, len(r) - 1))
            elif op < 0.8:
                r.insert(pos, int(rng.integers(0, 60)))       # not necessarily in order: a substitution-like edit
            elif r:
                r.insert(pos, r[min(pos, len(r) - 1)])          # duplicate a neighbour
        rows.append(np.array(r, dtype=np.int32))
    indptr = np.zeros(len(rows) + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows) if indptr[-1] else np.zeros(0, np.int32)
    labels, st = _lib.cluster_csr(indptr, indices, d)
    assert np.array_equal(labels, orc.cluster_csr(indptr, indices, d, n_threads=ORACLE_THREADS)["labels"])
    # the exact edge set, by brute force on the dense count matrix
    from scipy.spatial.distance import cdist

    dense = np.zeros((len(rows), 60), np.int32)
    for i, r in enumerate(rows):
        np.add.at(dense[i], r, 1)
    dist = cdist(dense, dense, "cityblock")
    want_edges = int((np.triu(dist <= d, 1)).sum())
    assert st["n_edges"] == want_edges
    ptr, idx = _lib.neighbours_csr(indptr, indices, d)
    for i in rng.integers(0, len(rows), size=200):
        assert np.array_equal(idx[ptr[i]: ptr[i + 1]], np.flatnonzero(dist[i] <= d))


@pytest.mark.exact_edges
@pytest.mark.parametrize("env", [{"BFK_PF_ROWS": "2"}, {"BFK_PF_ROWS": "4"}, {"BFK_PF_WAVES": "2"}, {"BFK_PF_WAVES": "4"},
                                 {"BFK_VERIFY_GRID": "32"}, {"BFK_VERIFY_GRID": "8192"}, {"BFK_KEY_H": "16"},
                                 {"BFK_KEY_H": "16", "BFK_PF_ROWS": "2"}, {"BFK_UNION_BATCH": "2"},
                                 {"BFK_UNION_BATCH": "4"}, {"BFK_UNION_BATCH": "16"}, {"BFK_WAVE_TABLE_D": "3"},
                                 {"BFK_WAVE_TABLE_D": "0"}, {"BFK_SIG_WORDS": "1"}, {"BFK_SIG_WORDS": "2"},
                                 {"BFK_UF_LINK": "0"}, {"BFK_UF_LINK": "1"}, {"BFK_VERIFY_PHASES": "8"},
                                 {"BFK_VERIFY_PHASES": "3", "BFK_VERIFY_PHASE2": "2"}, {"BFK_VERIFY_PHASES": "1"},
                                 {"BFK_SIG_WORDS": "4", "BFK_PF_ROWS": "2"}])
@pytest.mark.parametrize("d", [1, 2, 3, 4])
def test_kernel_configurations_of_large_inputs_give_the_same_labels(env, d, monkeypatch):
    """128- and 256-row tiles (chosen above 2M / 8M rows), 2 and 4 waves per tile, extreme verify grids, the
    third sort key (chosen above 600k rows; d = 4 needs more candidate ranges than lanes and falls back to the
    two-key ranges over four-key cells), union batch sizes, hash-table placements, both union algorithms and the one- and
    two-phase form of the verify kernel: the configurations the default sizes of the test inputs never pick"""
    rows = generate_profiles(6000, p_del=0.05, p_ins=0.02)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    want, st0 = _lib.cluster_csr(indptr, indices, d)
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    monkeypatch.setenv("BFK_JOIN", "0")  # d = 1: these knobs configure the all-pairs kernels (want: the default path)
    got, st = _lib.cluster_csr(indptr, indices, d)
    assert np.array_equal(got, want)
    assert st["n_edges"] == st0["n_edges"] and st["n_candidates"] >= st["n_edges"]


@pytest.mark.exact_edges
@pytest.mark.parametrize("cap,d", [(64, 2), (16, 3), (4, 1)])
def test_candidate_queue_overflow_is_recovered(cap, d, monkeypatch):
    """a queue far too small for the input: dropped candidates raise the device flag, bfk_ctx_sync re-runs
    prefilter + verify over slices of the tile list until every slice fits (unions are idempotent), and the
    labels, the edge count and the neighbour lists must come out as with a queue that fits"""
    rows = generate_profiles(12000, p_del=0.05, p_ins=0.02)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    want, st0 = _lib.cluster_csr(indptr, indices, d)
    assert st0["n_retry_slices"] == 0
    ptr0, idx0 = _lib.neighbours_csr(indptr, indices, d)
    monkeypatch.setenv("BFK_CAND_CAP_SHARD", str(cap))
    monkeypatch.setenv("BFK_JOIN", "0")  # d = 1: the all-pairs kernels (the join certifies ordered profiles without the queue)
    got, st = _lib.cluster_csr(indptr, indices, d)
    assert st["n_retry_slices"] > 0
    assert np.array_equal(got, want)
    assert st["n_edges"] == st0["n_edges"]
    ptr, idx = _lib.neighbours_csr(indptr, indices, d)
    assert np.array_equal(ptr, ptr0) and np.array_equal(idx, idx0)


def _join_cases():
    rng = np.random.default_rng(77)
    cases = {}
    # small alphabet: repeated tokens inside rows, many rows that are one multiset in different orders
    cases["small_alphabet"] = _random_multisets(3000, 31, 6, 9)
    cases["medium"] = _random_multisets(2500, 32, 300, 30, p_empty=0.0)
    # rows longer than one 64-token chunk, with the inserted / repeated token before and after the chunk border
    base = rng.choice(5000, size=150, replace=False).astype(np.int32)
    rows = [base]
    for pos in (0, 10, 63, 64, 65, 127, 128, 149, 150):
        rows.append(np.insert(base, pos, 9999).astype(np.int32))            # base + new token
        rows.append(np.insert(base, pos, base[3]).astype(np.int32))          # base + repeat of an early token
        rows.append(np.insert(base, pos, base[140]).astype(np.int32))        # base + repeat of a late token
    rows += [rng.permutation(base).astype(np.int32) for _ in range(5)]      # same multiset, other orders
    rows.append(np.delete(base, 70).astype(np.int32))
    rows.append(np.append(np.append(base, 7), 8).astype(np.int32))          # distance 2
    indptr = np.zeros(len(rows) + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    cases["long_rows"] = (indptr, np.concatenate(rows).astype(np.int32))
    # 100 orders of one multiset (inside the probe limit of the join) and 400 of another (beyond it: fallback)
    for name, m in (("perms100", 100), ("perms400", 400)):
        row = rng.choice(1000, size=12, replace=False).astype(np.int32)
        rows = [rng.permutation(row) for _ in range(m)] + [np.append(row, 5000 + i).astype(np.int32) for i in range(50)]
        rows += [rng.choice(1000, size=10, replace=False).astype(np.int32) for _ in range(500)]
        indptr = np.zeros(len(rows) + 1, np.int32)
        indptr[1:] = np.cumsum([len(r) for r in rows])
        cases[name] = (indptr, np.concatenate(rows).astype(np.int32))
    uf = list(dict.fromkeys(generate_profiles(30000, p_del=0.05, p_ins=0.02)))
    cases["profiles30k"] = _lib.build_csr(uf, " ")[:2]
    # rows of 0..4 tokens: 512 consecutive tokens span far more than 64 rows, so k_join finds the row of a token by
    # binary search (slow_window) instead of from the 64 extents a wave loads
    rows = [rng.choice(4000, size=int(rng.integers(1, 5)), replace=False).astype(np.int32) for _ in range(6000)]
    for i in range(0, 6000, 150):
        rows[i] = np.zeros(0, np.int32)
    indptr = np.zeros(len(rows) + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    cases["tiny_rows"] = (indptr, np.concatenate(rows).astype(np.int32))
    return cases


@pytest.mark.exact_edges
@pytest.mark.parametrize("inline", ["default", "0"])
@pytest.mark.parametrize("name", ["small_alphabet", "medium", "long_rows", "perms100", "perms400", "profiles30k",
                                  "tiny_rows"])
def test_variant_join_equals_the_all_pairs_path(name, inline, monkeypatch):
    """max_dist 1 is served by the hash join (H(B) - h(t) lookups): labels, edge count and neighbour
    lists must equal those of the all-pairs kernels and of the oracle — repeated tokens (a pair must be found once),
    equal multisets in other orders, rows over several 64-token chunks, and the give-up path (probe chain beyond the
    limit -> the step is redone by the all-pairs kernels).  What the positional certificate cannot decide is counted
    exactly by k_join itself while no row has more than 128 tokens (no k_verify launch), and queued for k_verify beyond
    (`inline` = "0" forces the queue)"""
    indptr, indices = _join_cases()[name]
    if inline != "default":
        monkeypatch.setenv("BFK_JOIN_INLINE", inline)
    monkeypatch.setenv("BFK_JOIN", "0")
    want, st0 = _lib.cluster_csr(indptr, indices, 1)
    ptr0, idx0 = _lib.neighbours_csr(indptr, indices, 1)
    monkeypatch.setenv("BFK_JOIN", "1")
    got, st = _lib.cluster_csr(indptr, indices, 1)
    assert np.array_equal(got, want)
    assert np.array_equal(got, orc.cluster_csr(indptr, indices, 1, n_threads=ORACLE_THREADS)["labels"])
    assert st["n_edges"] == st0["n_edges"]
    if name != "small_alphabet":  # (hundreds of empty and one-token rows there: may or may not exceed the limit)
        assert (st["n_retry_slices"] > 0) == (name == "perms400")
    if st["n_retry_slices"] == 0:
        assert st["n_candidates"] == st["n_edges"]  # every table match is a real edge (up to 64-bit hash collisions)
    ptr, idx = _lib.neighbours_csr(indptr, indices, 1)
    assert np.array_equal(ptr, ptr0) and np.array_equal(idx, idx0)


@pytest.mark.exact_edges
@pytest.mark.parametrize("seed", range(int(os.environ.get("BFK_FUZZ_SEEDS", "6"))))
def test_variant_join_fuzz_vs_oracle(seed):
    """many small inputs of every shape the join special-cases: rows of 0 .. 700 tokens (single tokens, windows that
    hold dozens of rows, rows spanning several 512-token batches), repeated tokens, equal multisets in other orders,
    edits at the first / last position, few and many rows — labels and edge counts against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    by_join = 0  # inputs the join finished itself (no give-up: few rows that are one multiset)
    for it in range(25):
        n_base = int(rng.integers(1, 30))
        alphabet = int(rng.choice([3, 10, 100, 5000]))
        kmax = int(rng.choice([1, 4, 20, 70, 200, 700]))
        base = [rng.integers(0, alphabet, size=int(rng.integers(0, kmax + 1))) for _ in range(n_base)]
        rows = []
        for _ in range(int(rng.integers(1, 400))):
            r = list(base[int(rng.integers(0, n_base))])
            for _ in range(int(rng.integers(0, 3))):
                op = rng.random()
                pos = int(rng.choice([0, len(r), int(rng.integers(0, len(r) + 1))]))
                if op < 0.4 and r:
                    r.pop(min(pos, len(r) - 1))
                elif op < 0.8:
                    r.insert(pos, int(rng.integers(0, alphabet)))
                elif r:
                    r.insert(pos, r[int(rng.integers(0, len(r)))])
            if rng.random() < 0.1:
                rng.shuffle(r)
            rows.append(np.array(r, dtype=np.int32))
        indptr = np.zeros(len(rows) + 1, np.int32)
        indptr[1:] = np.cumsum([len(r) for r in rows])
        indices = np.concatenate(rows).astype(np.int32) if indptr[-1] else np.zeros(0, np.int32)
        want = orc.cluster_csr(indptr, indices, 1, n_threads=4, want_neigh=False)["labels"]
        got, st = _lib.cluster_csr(indptr, indices, 1)
        assert np.array_equal(got, want), (seed, it)
        by_join += st["n_retry_slices"] == 0
        dense = np.zeros((len(rows), alphabet), np.int32)
        for i, r in enumerate(rows):
            np.add.at(dense[i], r, 1)
        if len(rows) <= 200 and alphabet <= 100:  # exact edge count by brute force on the count matrix
            dist = np.abs(dense[:, None, :] - dense[None, :, :]).sum(axis=2)
            assert st["n_edges"] == int(np.triu(dist <= 1, 1).sum()), (seed, it)
    assert by_join >= 15, by_join


@pytest.mark.exact_edges
@pytest.mark.parametrize("seed", range(4))
def test_variant_join_inline_exact_count_vs_brute_force(seed):
    """k_join's own exact count (wave_rows_within): rows of 0 .. 128 tokens over a small alphabet, shuffled — the
    certificate fails for nearly every match, equal multisets and repeats are everywhere — labels vs the oracle, the edge
    count vs brute force on the count matrix; the same input with one row of 129 tokens goes through k_verify's queue"""
    rng = np.random.default_rng(4200 + seed)
    alphabet = 12
    for longest in (128, 129):
        rows = []
        for _ in range(300):
            k = int(rng.choice([0, 1, 2, 5, 40, 63, 64, 65, 127, 128]))
            base = np.sort(rng.integers(0, alphabet, size=k))
            for _ in range(int(rng.integers(1, 4))):
                r = list(base)
                if rng.random() < 0.5 and len(r) < 128:
                    r.append(int(rng.integers(0, alphabet)))
                rng.shuffle(r)
                rows.append(np.array(r, dtype=np.int32))
        rows.append(rng.integers(0, alphabet, size=longest).astype(np.int32))
        indptr = np.zeros(len(rows) + 1, np.int32)
        indptr[1:] = np.cumsum([len(r) for r in rows])
        indices = np.concatenate(rows).astype(np.int32)
        ctx = _lib.Context(0)
        ctx.set_candidate_path("join")
        ctx.upload_csr(indptr, indices)
        d_out = ctx.alloc(4 * len(rows))
        ctx.cluster(1, d_out)
        st = ctx.sync()
        got = ctx.download_i32(d_out, len(rows))
        ctx.close()
        want = orc.cluster_csr(indptr, indices, 1, n_threads=4, want_neigh=False)["labels"]
        assert np.array_equal(got, want), (seed, longest)
        if st["n_retry_slices"] == 0:  # (the join did it: no give-up on a long chain of equal multisets)
            assert st["path"] == 1
            dense = np.zeros((len(rows), alphabet), np.int32)
            for i, r in enumerate(rows):
                np.add.at(dense[i], r, 1)
            dist = np.abs(dense[:, None, :] - dense[None, :, :]).sum(axis=2)
            assert st["n_edges"] == int(np.triu(dist <= 1, 1).sum()), (seed, longest)


def fuzz_case(rng):
    """one small input of the general-path fuzzers: a few base rows, every row a base row with up to d + 1 edits (front,
    end, anywhere; repeats; now and then shuffled)"""
    d = int(rng.integers(2, 6))
    n_base = int(rng.integers(1, 20))
    alphabet = int(rng.choice([4, 12, 100, 3000]))
    kmax = int(rng.choice([3, 12, 40, 90, 250]))
    base = [np.sort(rng.integers(0, alphabet, size=int(rng.integers(0, kmax + 1)))) for _ in range(n_base)]
    rows = []
    for _ in range(int(rng.integers(1, 500))):
        r = list(base[int(rng.integers(0, n_base))])
        for _ in range(int(rng.integers(0, d + 2))):
            op = rng.random()
            pos = int(rng.choice([0, len(r), int(rng.integers(0, len(r) + 1))]))
            if op < 0.45 and r:
                r.pop(min(pos, len(r) - 1))
            elif op < 0.9:
                r.insert(pos, int(rng.integers(0, alphabet)))
            elif r:
                r.insert(pos, r[int(rng.integers(0, len(r)))])
        if rng.random() < 0.05:
            rng.shuffle(r)
        rows.append(np.array(r, dtype=np.int32))
    indptr = np.zeros(len(rows) + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows).astype(np.int32) if indptr[-1] else np.zeros(0, np.int32)
    return rows, indptr, indices, d, alphabet


@pytest.mark.parametrize("d,indels,path", [(1, False, "auto"), (1, True, "allpairs"), (2, True, "auto"), (5, True, "prefix")])
def test_neighbours_of_selected_rows_equal_the_full_lists(d, indels, path, monkeypatch):
    """bfk_neighbours_csr(select_ind): the kernels record only edges with a selected end (a bit per row on the device) and
    the lists are built for the selected rows alone — each equal to the row's list of the full call, in the caller's order,
    for sorted, unsorted and repeated selections, on every candidate generator (get_neighbours_batch's select_ind,
    breakfast.py:241-245)"""
    kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
    rows = list(dict.fromkeys(generate_profiles(12000, **kw)))
    indptr, indices, _ = _lib.build_csr(rows, " ")
    if path == "allpairs":
        monkeypatch.setenv("BFK_JOIN", "0")
    if path == "prefix":
        monkeypatch.setenv("BFK_PG", "1")
    full_ptr, full_idx = _lib.neighbours_csr(indptr, indices, d)
    full = [full_idx[full_ptr[i]: full_ptr[i + 1]] for i in range(len(rows))]
    assert sum(len(x) for x in full) > 2 * len(rows)
    rng = np.random.default_rng(d)
    for sel in (np.sort(rng.choice(len(rows), 1200, replace=False)), rng.choice(len(rows), 500, replace=True),
                np.array([len(rows) - 1, 0, 0, 7]), np.zeros(0, np.int64), np.arange(len(rows))[::-1].copy()):
        sel = sel.astype(np.int64)
        ptr, idx = _lib.neighbours_csr(indptr, indices, d, sel)
        assert len(ptr) == len(sel) + 1
        for s, i in enumerate(sel.tolist()):
            assert np.array_equal(idx[ptr[s]: ptr[s + 1]], full[i]), (s, i)


@pytest.mark.exact_edges
@pytest.mark.parametrize("generator", ["band", "prefix", "prefix_pos"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("BFK_FUZZ_SEEDS", "4"))))
def test_all_pairs_fuzz_vs_oracle(seed, generator, monkeypatch):
    """the general path on many small inputs at max-dist 2 .. 5 (splicing and find + hook unions, one- and two-phase
    verify, both certificates and the counting table), with the candidates from the band kernels and from the prefix
    groups (tiny alphabets, rows shorter than the prefix, repeated tokens, shuffled rows): labels against the oracle, edge
    counts against brute force"""
    set_generator(monkeypatch, generator)
    rng = np.random.default_rng(5000 + seed)
    for it in range(20):
        rows, indptr, indices, d, alphabet = fuzz_case(rng)
        want = orc.cluster_csr(indptr, indices, d, n_threads=4)["labels"]
        got, st = _lib.cluster_csr(indptr, indices, d)
        assert np.array_equal(got, want), (seed, it, d)
        if len(rows) <= 250 and alphabet <= 100:
            dense = np.zeros((len(rows), alphabet), np.int32)
            for i, r in enumerate(rows):
                np.add.at(dense[i], r, 1)
            dist = np.abs(dense[:, None, :] - dense[None, :, :]).sum(axis=2)
            assert st["n_edges"] == int(np.triu(dist <= d, 1).sum()), (seed, it, d)


@pytest.mark.exact_edges
@pytest.mark.parametrize("name,n_shards", [("perms100", 3), ("tiny_rows", 4), ("long_rows", 2)])
def test_variant_join_sharded_equals_one_shard(name, n_shards):
    """the join's multi-GPU split on one GPU: blocks of tokens (their lookups) round-robin, a pair of equal multisets to
    the shard its later row picks — every edge exactly once over the shards, merged labels = the 1-shard labels"""
    indptr, indices = _join_cases()[name]
    n = len(indptr) - 1
    want, st1 = _lib.cluster_csr(indptr, indices, 1)
    assert st1["n_retry_slices"] == 0
    ctx = _lib.Context(0)
    ctx.upload_csr(indptr, indices)
    d_gath = ctx.alloc(4 * n * n_shards)
    d_out = ctx.alloc(4 * n)
    edges = 0
    for s_ in range(n_shards):
        ctx.cluster(1, d_gath + 4 * n * s_, s_, n_shards)
        st = ctx.sync()
        assert st["n_retry_slices"] == 0
        edges += st["n_edges"]
    ctx.merge_labels(d_gath, n_shards, d_out)
    ctx.sync()
    got = ctx.download_i32(d_out, n)
    ctx.close()
    assert np.array_equal(got, want)
    assert edges == st1["n_edges"]


@pytest.mark.exact_edges
def test_variant_join_queue_overflow_falls_back(monkeypatch):
    """rows in no common order: the join cannot certify its matches itself and queues them for k_verify; a queue
    that is too small makes bfk_ctx_sync redo the step on the all-pairs path (which has the sliced recovery)"""
    indptr, indices = _join_cases()["medium"]
    monkeypatch.setenv("BFK_JOIN_INLINE", "0")  # (rows of at most 128 tokens are otherwise decided by k_join itself: no queue)
    want, st0 = _lib.cluster_csr(indptr, indices, 1)
    assert st0["n_retry_slices"] == 0
    monkeypatch.setenv("BFK_CAND_CAP_SHARD", "1")
    got, st = _lib.cluster_csr(indptr, indices, 1)
    assert st["n_retry_slices"] > 0
    assert np.array_equal(got, want) and st["n_edges"] == st0["n_edges"]


@pytest.mark.exact_edges
def test_variant_join_on_a_resident_context_alternates_its_tables():
    """steps on one context: the join clears the table set of the NEXT step; rebinding a smaller and a larger CSR
    in between must not leave stale entries"""
    cases = _join_cases()
    ctx = _lib.Context(0)
    for name in ("medium", "long_rows", "profiles30k", "tiny_rows", "medium", "perms100"):
        indptr, indices = cases[name]
        n = len(indptr) - 1
        want = orc.cluster_csr(indptr, indices, 1, n_threads=ORACLE_THREADS)["labels"]
        ctx.upload_csr(indptr, indices)
        d_out = ctx.alloc(4 * n)
        for _ in range(3):
            ctx.cluster(1, d_out)
            st = ctx.sync()
            assert st["n_retry_slices"] == 0
            assert np.array_equal(ctx.download_i32(d_out, n), want)
        ctx.cluster(2, d_out)  # the all-pairs path in between
        ctx.sync()
        ctx.cluster(1, d_out)
        st2 = ctx.sync()
        assert np.array_equal(ctx.download_i32(d_out, n), want)
        assert st2["n_edges"] == st["n_edges"] and st2["n_candidates"] == st["n_candidates"]
    ctx.close()


@pytest.mark.skipif(not os.environ.get("BFK_SLOW_TESTS"), reason="opt-in (BFK_SLOW_TESTS=1): the oracle needs minutes")
@pytest.mark.parametrize("indels", [False, True])
@pytest.mark.parametrize("d", [1, 2, 3, 5])
def test_50k_rows_full_oracle_comparison(d, indels):
    """the end-of-round check: 50k profiles, every max-dist the BASELINE configurations use, with and without indels —
    labels equal to the oracle's (9-90 s of oracle time each on 16 cores)"""
    kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
    uf = list(dict.fromkeys(generate_profiles(50000, **kw)))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    got, st = _lib.cluster_csr(indptr, indices, d)
    want = orc.cluster_csr(indptr, indices, d, n_threads=os.cpu_count() or 8)["labels"]
    assert d > 3 or st["n_retry_slices"] == 0  # (d = 5 without indels overflows the first queue: recovered in slices)
    assert np.array_equal(got, want)


def test_allreduce_min_merge_reaches_the_fix_point():
    """the north-star merge form on one GPU: elementwise MIN of the shards' label arrays (what all_reduce(MIN)
    delivers), united into the forest with bfk_ctx_merge_labels(n_parts=1) until the changed flag stays 0"""
    rows = generate_profiles(15000, p_del=0.03, p_ins=0.01)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    n, d, n_shards = len(uf), 2, 4
    want, _ = _lib.cluster_csr(indptr, indices, d)
    # every "rank" = one context with its own local forest
    ctxs, local = [], []
    for s in range(n_shards):
        c = _lib.Context(0)
        c.upload_csr(indptr, indices)
        dl = c.alloc(4 * n)
        c.cluster(d, dl, s, n_shards)
        c.sync()
        ctxs.append(c)
        local.append(c.download_i32(dl, n).copy())
    cur = local
    for rounds in range(1, 20):
        red = np.minimum.reduce(cur).astype(np.int32)  # all_reduce(MIN)
        nxt, changed = [], 0
        for c in ctxs:
            d_red, d_out, d_flag = c.alloc(4 * n), c.alloc(4 * n), c.alloc(4)
            c.upload_i32(red, d_red)
            c.merge_labels(d_red, 1, d_out, d_flag)
            c.sync()
            nxt.append(c.download_i32(d_out, n).copy())
            changed |= int(c.download_i32(d_flag, 1)[0])
        cur = nxt
        if not changed:
            break
    for c in ctxs:
        c.close()
    assert rounds >= 2  # at least one confirming round
    for l in cur:
        assert np.array_equal(l, want)


_ORACLE_LABELS = {}


def oracle_labels_indel(n, d):
    """labels of the CPU oracle for generate_profiles(n, indels kept) at max-dist d (16 host threads: ~15 s at 20k rows and
    d = 3, ~25 s at d = 5), computed once per session"""
    if (n, d) not in _ORACLE_LABELS:
        uf = list(dict.fromkeys(generate_profiles(n, p_del=0.05, p_ins=0.01)))
        indptr, indices, _ = _lib.build_csr(uf, " ")
        _ORACLE_LABELS[(n, d)] = orc.cluster_csr(indptr, indices, d, n_threads=16)["labels"]
    return _ORACLE_LABELS[(n, d)]


def test_max_dist_1_long_rows_turn_to_the_band_kernels_where_the_join_stops_paying():
    """join_pays (bfk_host.cpp): the join's look-ups go with the tokens, the band kernels' pair work with the rows — rows of ~105
    tokens take the band kernels from ~60k rows on (by CSR, and by text: the step's text says 'long rows' before it is tokenised and
    the step waits once), rows of ~40 tokens stay on the join; every choice gives the labels of the forced generators"""
    from breakfast_amd.synth import generate_family

    for family, n, want_path in (("long", 90000, 0), ("long", 30000, 1), ("default", 90000, 1)):
        uf = list(dict.fromkeys(generate_family(family, n) if family != "default" else generate_profiles(n)))
        indptr, indices, _ = _lib.build_csr(uf, " ")
        got = {}
        for path in ("auto", "join", "allpairs"):
            ctx = _lib.Context(0)
            ctx.set_candidate_path(path)
            ctx.upload_csr(indptr, indices)
            d_out = ctx.alloc(4 * len(uf))
            ctx.cluster(1, d_out)
            st = ctx.sync()
            got[path] = ctx.download_i32(d_out, len(uf)).copy()
            if path == "auto":
                assert st["path"] == want_path, (family, n, st["path"])
            ctx.close()
        assert np.array_equal(got["auto"], got["join"]) and np.array_equal(got["auto"], got["allpairs"]), (family, n)
        lab_t, st_t, nnz, _ = _lib.cluster_text(*_lib.pack_rows(uf), " ", 1)
        assert nnz == len(indices) and np.array_equal(lab_t, got["auto"]) and st_t["path"] == want_path, (family, n)


@pytest.mark.parametrize("family,d", [("long", 1), ("long", 3), ("star", 1), ("star", 2), ("star", 5), ("aa", 1), ("aa", 3)])
def test_other_workload_families_equal_the_oracle_at_20k_rows(family, d):
    """workload shapes the dispatch thresholds were not fitted on (breakfast_amd/synth.py: generate_family — rows of 100+
    tokens, a star phylogeny whose hubs have thousands of neighbours, amino-acid tokens): the default configuration and every
    forced candidate generator against the FULL oracle at 20k rows; tools/family_matrix.py times them"""
    from breakfast_amd.synth import generate_family

    uf = list(dict.fromkeys(generate_family(family, 20000)))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    want = orc.cluster_csr(indptr, indices, d, n_threads=16)["labels"]
    labels, st = _lib.cluster_csr(indptr, indices, d)
    assert np.array_equal(labels, want)
    assert st["n_retry_slices"] == 0
    for path in ("allpairs", "join" if d == 1 else "prefix"):
        ctx = _lib.Context(0)
        ctx.set_candidate_path(path)
        ctx.set_exact_edges(True)
        ctx.upload_csr(indptr, indices)
        d_out = ctx.alloc(4 * len(uf))
        ctx.cluster(d, d_out)
        st2 = ctx.sync()
        assert np.array_equal(ctx.download_i32(d_out, len(uf)), want), path
        if path == "allpairs":
            n_edges = st2["n_edges"]
        else:
            assert st2["n_edges"] == n_edges  # every generator finds the same edge set
        ctx.close()
    # the text entry on the same rows (long rows: a device-driven step redone by the host; aa: tokens beyond the inline key)
    buf, off = _lib.pack_rows(uf)
    lab_t, _, nnz, _ = _lib.cluster_text(buf, off, " ", d)
    assert nnz == len(indices) and np.array_equal(lab_t, want)


@pytest.mark.parametrize("d", [3, 5])
def test_default_configuration_equals_the_oracle_at_20k_rows(d):
    """what a caller gets with no knob set at max-dist 3 and 5 on 20k rows with indels kept — automatic generator choice,
    candidates of already connected rows dropped unchecked (k_verify_connected) — against the FULL oracle; and the prefix
    groups forced (below their automatic threshold at d = 3) on the same input"""
    uf = list(dict.fromkeys(generate_profiles(20000, p_del=0.05, p_ins=0.01)))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    want = oracle_labels_indel(20000, d)
    labels, st = _lib.cluster_csr(indptr, indices, d)
    assert np.array_equal(labels, want)
    assert st["n_retry_slices"] == 0
    for pos in ("0", "1"):  # whole-group walk / positional filter (what inputs from 60k rows get)
        os.environ["BFK_PG_POS"] = pos
        try:
            ctx = _lib.Context(0)
            ctx.set_candidate_path("prefix")
            ctx.upload_csr(indptr, indices)
            d_out = ctx.alloc(4 * len(uf))
            ctx.cluster(d, d_out)
            st2 = ctx.sync()
            assert st2["path"] == 2 and np.array_equal(ctx.download_i32(d_out, len(uf)), want)
            assert st2["n_connected"] > 0  # the pruning verify was the one that ran
            ctx.close()
        finally:
            del os.environ["BFK_PG_POS"]


# ---- prefix groups (max-dist >= 4 on large inputs; any max-dist 2..7 when forced) against the band kernels and the oracle ----
@pytest.mark.exact_edges
@pytest.mark.parametrize("pos", ["0", "1"])
@pytest.mark.parametrize("n,d,indels", [(3000, 2, False), (20000, 3, True), (20000, 5, True), (60000, 4, True), (777, 7, True)])
def test_prefix_groups_equal_the_band_path(n, d, indels, pos, monkeypatch):
    """bfk_ctx_set_candidate_path(3): candidates from the groups of the rows' prefix elements instead of (k,f,g)
    bands — same labels, same number of edges (every pair counted once, at the first element its rows share) — with the
    whole-group walk (pos 0) and with the positional filter (pos 1: a record walks one range of position sub-groups)"""
    monkeypatch.setenv("BFK_PG_POS", pos)
    kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
    uf = list(dict.fromkeys(generate_profiles(n, **kw)))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    nu = len(uf)
    ctxb = _lib.Context(0)
    ctxb.set_candidate_path("allpairs")
    ctxb.upload_csr(indptr, indices)
    d_b = ctxb.alloc(4 * nu)
    for _ in range(2):
        ctxb.cluster(d, d_b)
        st_band = ctxb.sync()
    want = ctxb.download_i32(d_b, nu).copy()
    ctxb.close()
    assert st_band["n_work_items"] > 0
    ctx = _lib.Context(0)
    ctx.set_candidate_path("prefix")
    ctx.upload_csr(indptr, indices)
    d_out = ctx.alloc(4 * nu)
    for _ in range(2):  # (a dense first run may grow the candidate queue: the second is a single pass)
        ctx.cluster(d, d_out)
        st = ctx.sync()
    assert np.array_equal(ctx.download_i32(d_out, nu), want)
    assert st["n_edges"] == st_band["n_edges"] and st["n_retry_slices"] == 0
    assert st["n_work_items"] == (nu + 63) // 64  # work items of the prefix-group path: blocks of 64 rows
    ctx.close()
    if n <= 3000:
        assert np.array_equal(want, orc.cluster_csr(indptr, indices, d, n_threads=ORACLE_THREADS)["labels"])
    elif n <= 20000 and indels:  # the oracle for every size it can afford
        assert np.array_equal(want, oracle_labels_indel(n, d))


@pytest.mark.exact_edges
@pytest.mark.parametrize("top", [2**31 - 1, 2**24 + 5, 70_000])
def test_prefix_groups_with_sparse_token_ids(top, monkeypatch):
    """the record keys are token + 1 and the sort looks at as many bits as the largest id needs (up to 31): the same rows
    with their token ids spread over [0, top] — and the largest possible id present — must give the same labels and edges"""
    uf = list(dict.fromkeys(generate_profiles(4000, p_del=0.05, p_ins=0.01)))
    indptr, indices, nv = _lib.build_csr(uf, " ")
    rng = np.random.default_rng(top % 1000)
    ids = np.unique(rng.integers(0, top, size=4 * nv))[: nv - 1]           # ascending, distinct, below top
    remap = np.concatenate([ids, [top]]).astype(np.int32)       # monotone: "higher id = newer" keeps its meaning
    spread = remap[indices]
    monkeypatch.setenv("BFK_PG", "0")
    want, st_band = _lib.cluster_csr(indptr, indices, 4)
    monkeypatch.setenv("BFK_PG", "1")
    got, st = _lib.cluster_csr(indptr, spread, 4)
    assert st["path"] == 2 and np.array_equal(got, want) and st["n_edges"] == st_band["n_edges"]


@pytest.mark.parametrize("seed", range(3))
def test_prefix_groups_on_random_multisets(seed, monkeypatch):
    """shuffled rows, repeated tokens, empty rows, rows shorter than the prefix (they meet in the SHORT group)"""
    monkeypatch.setenv("BFK_PG", "1")
    for d in (2, 3, 6):
        indptr, indices = _random_multisets(900, 100 * seed + d, alphabet=60, kmax=3 * d)
        want = orc.cluster_csr(indptr, indices, d, n_threads=ORACLE_THREADS)
        got, st = _lib.cluster_csr(indptr, indices, d)
        assert np.array_equal(got, want["labels"]), (seed, d)


@pytest.mark.exact_edges
def test_prefix_groups_hub_row_overflows_the_lds_set(monkeypatch):
    """one row with 3000 neighbours at distance 1..2: the wave's LDS set (1024 slots) fills up and the rest of the row's
    members are de-duplicated by the slow exact test; every pair still counted once"""
    monkeypatch.setenv("BFK_PG", "1")
    rng = np.random.default_rng(5)
    hub = np.arange(100, 140, dtype=np.int32)
    rows = [hub]
    for i in range(3000):
        extra = rng.integers(1000, 200000, size=int(rng.integers(1, 3)))
        rows.append(np.concatenate([hub, extra.astype(np.int32)]))
    indptr = np.zeros(len(rows) + 1, np.int32)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows).astype(np.int32)
    for d in (2, 4):
        want = orc.cluster_csr(indptr, indices, d, n_threads=ORACLE_THREADS)
        got, st = _lib.cluster_csr(indptr, indices, d)
        assert np.array_equal(got, want["labels"])
        monkeypatch.setenv("BFK_PG", "0")
        _, st_band = _lib.cluster_csr(indptr, indices, d)
        monkeypatch.setenv("BFK_PG", "1")
        assert st["n_edges"] == st_band["n_edges"]


@pytest.mark.exact_edges
def test_prefix_groups_give_up_on_big_groups(monkeypatch):
    """random rows over a small alphabet: every token is in a quarter of the rows, so the 'rarest' tokens of a row still
    group thousands of rows and walking the groups would be quadratic.  The sampled counts say so before any walk:
    k_pgjoin does nothing, the step is redone on the band kernels (n_retry_slices = 1) and the CSR stays there"""
    rng = np.random.default_rng(11)
    n = 20000
    rows = [np.sort(rng.choice(60, size=15, replace=False)).astype(np.int32) for _ in range(n)]
    indptr = (np.arange(n + 1) * 15).astype(np.int32)
    indices = np.concatenate(rows)
    ctxb = _lib.Context(0)
    ctxb.set_candidate_path("allpairs")
    ctxb.upload_csr(indptr, indices)
    d_b = ctxb.alloc(4 * n)
    ctxb.cluster(4, d_b)
    st_band = ctxb.sync()
    want = ctxb.download_i32(d_b, n).copy()
    ctxb.close()
    ctx = _lib.Context(0)
    ctx.set_candidate_path("prefix")
    ctx.upload_csr(indptr, indices)
    d_out = ctx.alloc(4 * n)
    ctx.cluster(4, d_out)
    st = ctx.sync()
    assert st["n_retry_slices"] == 1 and st["path"] == 0      # redone on the band kernels
    assert np.array_equal(ctx.download_i32(d_out, n), want) and st["n_edges"] == st_band["n_edges"]
    ctx.cluster(4, d_out)
    st2 = ctx.sync()
    assert st2["n_retry_slices"] == 0 and st2["path"] == 0    # and stays there for this CSR
    assert np.array_equal(ctx.download_i32(d_out, n), want)
    ctx.close()
