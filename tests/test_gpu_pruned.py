"""The library's default for labels-only steps at max-dist >= 3: a candidate pair whose rows are already in one component is
dropped without its distance being computed (k_verify_connected).  The reference computes every distance
(src/breakfast/breakfast.py:261-276) and then keeps only the connected components (:325-329), so the labels must be the
same bit for bit; what changes is bfk_stats: n_edges counts the edges that went through the exact check, n_connected the
candidates that were dropped.  The tests that compare n_edges with the number of edges of the graph carry the
`exact_edges` marker (every candidate checked); here the default runs against the golden vectors, the oracle and the
exact mode.  BFK_SKIP_CONNECTED=1 forces the pruning kernel at max-dist 1 and 2 as well."""

import os

import numpy as np
import pytest
from conftest import load_stage, stage_names, set_generator
from test_gpu_parity import fuzz_case

from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles
from oracle import ref_port as orc

pytestmark = pytest.mark.gpu


def _invariants(st, labels):
    n = len(labels)
    comps = len(np.unique(labels))
    assert st["n_edges"] + st["n_connected"] <= st["n_candidates"]
    if st["n_retry_slices"] == 0:           # (after a queue overflow the counts are those of the recovery slices, which find
        assert st["n_edges"] >= n - comps    #  the first pass's unions done) a spanning forest went through the exact check


@pytest.mark.parametrize("force", [False, True])
@pytest.mark.parametrize("name", stage_names())
def test_golden_labels_with_pruning(name, force, monkeypatch):
    g = load_stage(name)
    if force:
        monkeypatch.setenv("BFK_SKIP_CONNECTED", "1")
    elif g["max_dist"] < 3:
        pytest.skip("default: no pruning below max-dist 3 (covered by test_labels_match_reference_golden)")
    labels, st = _lib.cluster_csr(g["indptr"], g["indices"], g["max_dist"])
    assert np.array_equal(labels, g["labels"])
    assert st["n_edges"] <= len(g["edges"])
    _invariants(st, labels)


def test_dense_max_dist_2_graphs_are_verified_by_the_pruning_kernel_by_themselves():
    """max-dist 2, labels only: both verify kernels are launched and the candidate queue's fill decides on the device which one
    works — a star phylogeny (hubs with hundreds of neighbours at distance 1: all of them within 2 of each other) takes the
    pruning kernel (n_connected > 0), the default generator's sparse forest the exact one; labels equal the oracle's either way
    and equal the exact mode's"""
    from breakfast_amd.synth import generate_family

    for fam, dense in (("star", True), ("default", False)):  # (star, 60k rows: ~11 candidates per row; default: ~2)
        uf = list(dict.fromkeys(generate_family(fam, 60000) if fam != "default" else generate_profiles(20000)))
        indptr, indices, _ = _lib.build_csr(uf, " ")
        want = orc.cluster_csr(indptr, indices, 2, n_threads=16)["labels"]
        got, st = _lib.cluster_csr(indptr, indices, 2)
        assert np.array_equal(got, want) and st["n_retry_slices"] == 0
        assert (st["n_connected"] > 0) == dense, (fam, st)
        assert st["n_edges"] + st["n_connected"] <= st["n_candidates"]
        ctx = _lib.Context(0)
        ctx.set_exact_edges(True)
        ctx.upload_csr(indptr, indices)
        d_out = ctx.alloc(4 * len(uf))
        ctx.cluster(2, d_out)
        stx = ctx.sync()
        assert stx["n_connected"] == 0 and np.array_equal(ctx.download_i32(d_out, len(uf)), want)
        ctx.close()


def test_a_dense_max_dist_2_step_sends_the_next_ones_on_that_csr_to_the_prefix_groups():
    """what only a step reveals: a band step at max-dist 2 that queued more than 8 candidates per row (star phylogeny) makes
    the context take the prefix groups for the following steps on that CSR (1.2-1.45x faster there); a new bind starts over,
    a forced generator is never overridden, the labels are the same either way"""
    from breakfast_amd.synth import generate_family

    uf = list(dict.fromkeys(generate_family("star", 60000)))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    ctx = _lib.Context(0)
    ctx.upload_csr(indptr, indices)
    d_out = ctx.alloc(4 * len(uf))
    ctx.cluster(2, d_out)
    st1 = ctx.sync()
    lab1 = ctx.download_i32(d_out, len(uf)).copy()
    assert st1["path"] == 0 and st1["n_candidates"] > 8 * len(uf)
    ctx.cluster(2, d_out)
    st2 = ctx.sync()
    assert st2["path"] == 2 and st2["n_retry_slices"] == 0 and np.array_equal(ctx.download_i32(d_out, len(uf)), lab1)
    ctx.set_candidate_path("allpairs")  # forced: stays
    ctx.cluster(2, d_out)
    assert ctx.sync()["path"] == 0
    ctx.set_candidate_path("auto")
    sparse = list(dict.fromkeys(generate_profiles(30000)))
    ip2, ix2, _ = _lib.build_csr(sparse, " ")
    ctx.upload_csr(ip2, ix2)  # a new CSR: what the old one taught is gone
    d2 = ctx.alloc(4 * len(sparse))
    for _ in range(2):
        ctx.cluster(2, d2)
        assert ctx.sync()["path"] == 0
    ctx.close()


@pytest.mark.parametrize("generator", ["band", "prefix", "prefix_pos"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("BFK_FUZZ_SEEDS", "4"))))
def test_fuzz_vs_oracle_with_pruning(seed, generator, monkeypatch):
    """the inputs of test_all_pairs_fuzz_vs_oracle (max-dist 2 .. 5) with the pruning kernel at every max-dist"""
    set_generator(monkeypatch, generator)
    monkeypatch.setenv("BFK_SKIP_CONNECTED", "1")
    rng = np.random.default_rng(5000 + seed)
    dropped = 0
    for it in range(20):
        rows, indptr, indices, d, alphabet = fuzz_case(rng)
        want = orc.cluster_csr(indptr, indices, d, n_threads=4)["labels"]
        got, st = _lib.cluster_csr(indptr, indices, d)
        assert np.array_equal(got, want), (seed, it, d)
        _invariants(st, got)
        dropped += st["n_connected"]
    assert dropped > 0


@pytest.mark.parametrize("d,indels,path", [(3, False, "allpairs"), (3, False, "auto"), (4, True, "auto"), (5, True, "auto")])
def test_100k_rows_pruned_equals_exact(d, indels, path):
    """both kernels on one resident context, alternating (the counters of one mode must not leak into the other); the
    prefix groups (the default at this size from max-dist 3), the band kernels (two-phase verify) at 3"""
    rows = generate_profiles(100_000, p_del=0.05, p_ins=0.01) if indels else generate_profiles(100_000)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    n = len(uf)
    ctx = _lib.Context(0)
    ctx.set_candidate_path(path)
    ctx.upload_csr(indptr, indices)
    d_out = ctx.alloc(4 * n)
    res = {}
    for exact in (True, False, True, False):
        ctx.set_exact_edges(exact)
        ctx.cluster(d, d_out)
        st = ctx.sync()
        labels = ctx.download_i32(d_out, n).copy()
        if exact in res:
            assert np.array_equal(labels, res[exact][0]) and st["n_candidates"] == res[exact][1]["n_candidates"]
            if exact:
                assert st["n_edges"] == res[exact][1]["n_edges"]
        res[exact] = (labels, st)
    ctx.close()
    (lx, sx), (lp, sp) = res[True], res[False]
    assert np.array_equal(lx, lp)
    assert sx["n_connected"] == 0 and sp["n_connected"] > 0
    # (prefix groups, labels-only: a hub row whose de-duplication set is half full lets its further pairs through, so a pair
    # can be queued — and dropped as connected — more than once)
    assert sp["n_candidates"] >= sx["n_candidates"] if sp["path"] == 2 else sp["n_candidates"] == sx["n_candidates"]
    assert sp["n_edges"] <= sx["n_edges"]
    _invariants(sp, lp)
    assert sp["path"] == (0 if path == "allpairs" else 2)


def test_pruned_sharded_runs_merge_to_the_one_shard_labels():
    rows = generate_profiles(30_000, p_del=0.05, p_ins=0.01)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    n = len(uf)
    for d in (3, 5):
        want, st1 = _lib.cluster_csr(indptr, indices, d)
        ctx = _lib.Context(0)
        ctx.upload_csr(indptr, indices)
        n_shards = 3
        d_gath = ctx.alloc(4 * n * n_shards)
        d_out = ctx.alloc(4 * n)
        dropped = 0
        for s in range(n_shards):
            ctx.cluster(d, d_gath + 4 * n * s, s, n_shards)
            dropped += ctx.sync()["n_connected"]
        ctx.merge_labels(d_gath, n_shards, d_out)
        ctx.sync()
        assert np.array_equal(ctx.download_i32(d_out, n), want)
        assert dropped > 0
        ctx.close()


def test_pruned_queue_overflow_is_recovered(monkeypatch):
    """the recovery slices run the pruning kernel too: its counters add up over the slices"""
    rows = generate_profiles(20_000, p_del=0.05, p_ins=0.01)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    want, st0 = _lib.cluster_csr(indptr, indices, 4)
    monkeypatch.setenv("BFK_CAND_CAP_SHARD", "64")
    got, st = _lib.cluster_csr(indptr, indices, 4)
    assert st["n_retry_slices"] > 0
    assert np.array_equal(got, want)
    _invariants(st, got)


@pytest.mark.parametrize("d,exact", [(4, False), (3, True), (2, True)])
def test_sharded_queue_overflow_is_recovered(d, exact, monkeypatch):
    """recovery slices of a SHARD: the walk's row blocks are the rank's blocks inside [t_begin, t_end), numbered through —
    every slice boundary and every shard offset must still cover each of the rank's rows exactly once"""
    rows = generate_profiles(20_000, p_del=0.05, p_ins=0.01)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    n = len(uf)
    ctx0 = _lib.Context(0)
    ctx0.set_candidate_path("prefix")
    ctx0.set_exact_edges(exact)
    ctx0.upload_csr(indptr, indices)
    d_ref = ctx0.alloc(4 * n)
    ctx0.cluster(d, d_ref)
    st0 = ctx0.sync()
    want = ctx0.download_i32(d_ref, n).copy()
    ctx0.close()
    monkeypatch.setenv("BFK_CAND_CAP_SHARD", "64")
    n_shards = 3
    ctx = _lib.Context(0)
    ctx.set_candidate_path("prefix")
    ctx.set_exact_edges(exact)
    ctx.upload_csr(indptr, indices)
    d_gath = ctx.alloc(4 * n * n_shards)
    d_out = ctx.alloc(4 * n)
    edges = cands = retries = 0
    for s in range(n_shards):
        ctx.cluster(d, d_gath + 4 * n * s, s, n_shards)
        st = ctx.sync()
        assert st["path"] == 2
        edges += st["n_edges"]
        cands += st["n_candidates"]
        retries += st["n_retry_slices"]
    ctx.merge_labels(d_gath, n_shards, d_out)
    ctx.sync()
    got = ctx.download_i32(d_out, n)
    ctx.close()
    assert retries > 0
    assert np.array_equal(got, want)
    if exact:  # every pair queued once, by exactly one shard, in exactly one slice
        assert cands == st0["n_candidates"] and edges == st0["n_edges"]


def test_edge_capture_and_neighbour_lists_are_never_pruned():
    """bfk_neighbours_csr (the cache path's lists) needs every edge: capture switches the pruning off"""
    g = load_stage("indel200_d5")
    ptr, idx = _lib.neighbours_csr(g["indptr"], g["indices"], g["max_dist"])
    n = len(g["indptr"]) - 1
    got = {(i, int(j)) for i in range(n) for j in idx[ptr[i]: ptr[i + 1]] if j > i}
    assert got == {tuple(e) for e in g["edges"].tolist()}


def test_pruned_prefix_groups_behind_n_gpus(monkeypatch):
    """bfk_cluster_csr(n_gpus = 4) — one context per device, every one with its own forest and its own pruning — on the
    prefix groups: the merged labels are those of one device"""
    rows = generate_profiles(30_000, p_del=0.05, p_ins=0.01)
    uf = list(dict.fromkeys(rows))
    indptr, indices, _ = _lib.build_csr(uf, " ")
    want, _ = _lib.cluster_csr(indptr, indices, 5)
    monkeypatch.setenv("BFK_PG", "1")
    if _lib.load().bfk_device_count() < 4:
        monkeypatch.setenv("BFK_MULTI_ONE_DEVICE", "1")
    got, st = _lib.cluster_csr(indptr, indices, 5, n_gpus=4)
    assert np.array_equal(got, want)
    assert st["path"] == 2 and st["n_connected"] > 0


def test_cli_bytes_at_max_dist_5_do_not_depend_on_the_generator_or_the_pruning(tmp_path, monkeypatch):
    """clusters.tsv of the CLI (native front end -> bfk_cluster_csr -> writer) at max-dist 5, indels kept: the default
    (prefix groups + pruning verify at this size), every candidate checked, and the band kernels give the same bytes"""
    import hashlib

    import click.testing

    from breakfast_amd import console
    from breakfast_amd.synth import generate_tsv

    inp = tmp_path / "in.tsv"
    generate_tsv(inp, 40_000, p_del=0.05, p_ins=0.01)
    digests = []
    for name, env in (("default", {}), ("exact", {"BFK_EXACT_EDGES": "1"}), ("band", {"BFK_PG": "0"})):
        for k in ("BFK_EXACT_EDGES", "BFK_PG"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = tmp_path / name
        res = click.testing.CliRunner().invoke(console.main, ["--input-file", str(inp), "--outdir", str(out), "--max-dist", "5",
                                                              "--var-type", "raw"])   # raw: every token counts, indels too
        assert res.exit_code == 0, (res.output, res.exception)
        digests.append(hashlib.sha256((out / "clusters.tsv").read_bytes()).hexdigest())
    assert digests[0] == digests[1] == digests[2]
