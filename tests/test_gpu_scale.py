"""GPU parity at BASELINE.json's full sizes — configs[3] (1M profiles, max-dist 1, the 8-GPU sharding run on one
GPU) and configs[4] (1M profiles, max-dist 5, indels kept) — through size-independent properties plus sampled rows
against the oracle's select_ind path (get_neighbours_batch, src/breakfast/breakfast.py:223-276: query rows x all
columns); the size switches of the host code (variant join up to 800k rows, third sort key from 600k) crossed with
NO environment knob; one 50k-row comparison against the full oracle; and the one-process-per-GPU driver
(GpuEngine + ShardedClusterer.step) on the HIP path.  Integer work: bit-exact."""

import os

import numpy as np
import pytest
import torch  # noqa: F401  (before libbfk: see conftest.py)

from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles
from oracle import ref_port as orc

pytestmark = pytest.mark.gpu
CORES = min(16, os.cpu_count() or 8)


@pytest.fixture(scope="module")
def million():
    """the 1M-row generator output (the smaller sizes below are its prefixes: the generator is sequential)"""
    return generate_profiles(1_000_000)


@pytest.fixture(scope="module")
def million_indels():
    return generate_profiles(1_000_000, p_del=0.05, p_ins=0.01)


def _csr(rows):
    uf = list(dict.fromkeys(rows))  # collapse_duplicates
    indptr, indices, _ = _lib.build_csr(uf, " ")
    return uf, indptr, indices


@pytest.fixture(scope="module")
def million_csr(million):
    """unique rows + CSR of the whole million, built ONCE: the unique rows of a prefix of the input are a prefix of these (first
    occurrences keep their order) and so is their CSR (vocabulary ids are handed out by first appearance) — every size below is a
    slice.  n_unique_of(n) = unique rows among the first n input rows."""
    seen, first = set(), np.zeros(len(million), dtype=np.int64)
    for i, r in enumerate(million):
        if r not in seen:
            seen.add(r)
            first[i] = 1
    uf, indptr, indices = _csr(million)
    return uf, indptr, indices, np.cumsum(first)


def _csr_prefix(million_csr, n_rows):
    uf, indptr, indices, cum = million_csr
    m = int(cum[n_rows - 1])
    return uf[:m], indptr[: m + 1].copy(), indices[: int(indptr[m])].copy()


def _check_fix_point(labels):
    n = len(labels)
    assert np.all(labels <= np.arange(n)) and np.array_equal(labels[labels], labels)


def _check_sampled_rows(indptr, indices, d, labels, n_sample, seed):
    """neighbour lists of sampled rows == the oracle's select_ind lists (one get_neighbours_batch call per query length
    q, like cluster_features' loop: the lists come back for the selected rows of the band of q, in row order); and
    every neighbour shares the row's label"""
    n = len(indptr) - 1
    rng = np.random.default_rng(seed)
    sel = np.sort(rng.choice(n, size=n_sample, replace=False)).astype(np.int64)
    ptr, idx = _lib.neighbours_csr(indptr, indices, d, sel)
    nf = np.diff(indptr).astype(np.int64)
    want = {int(i): set() for i in sel}
    for q in sorted({int(q) for i in sel for q in range(max(0, int(nf[i]) - d), int(nf[i]) + d + 1)}):
        rows_q = [int(i) for i in sel if abs(int(nf[i]) - q) <= d]
        lists = orc.get_neighbours_batch(indptr, indices, nf, q, d, select_ind=sel, n_threads=CORES)
        assert len(lists) == len(rows_q)
        for i, l in zip(rows_q, lists):
            assert i in l
            want[i] |= set(l.tolist())
    n_nb = 0
    for s, i in enumerate(sel.tolist()):
        got = idx[ptr[s]: ptr[s + 1]]
        assert set(got.tolist()) == want[i], f"row {i}"
        assert np.all(labels[got] == labels[i])
        n_nb += len(got) - 1
    return n_nb


def _check_permutation_invariance(uf, d, labels, seed):
    n = len(uf)
    perm = np.random.default_rng(seed).permutation(n)
    ip2, ix2, _ = _lib.build_csr([uf[i] for i in perm], " ")
    lp, _ = _lib.cluster_csr(ip2, ix2, d)
    back = np.empty(n, np.int64)
    back[perm] = np.arange(n)                      # original row -> permuted row
    canon = np.full(n, n, np.int64)
    np.minimum.at(canon, lp[back], np.arange(n))   # min original index per component of the permuted run
    assert np.array_equal(canon[lp[back]], labels)


def _sharded_labels(indptr, indices, d, n_shards):
    """the N-GPU path on one GPU: every shard into its own local forest, label arrays merged like after an
    all_gather"""
    n = len(indptr) - 1
    ctx = _lib.Context(0)
    ctx.upload_csr(indptr, indices)
    d_gath = ctx.alloc(4 * n * n_shards)
    d_out = ctx.alloc(4 * n)
    edges = 0
    for s in range(n_shards):
        for attempt in range(4):  # a dense shard may overflow the first queue: recovered inside sync, which grows it
            ctx.cluster(d, d_gath + 4 * n * s, s, n_shards)
            st = ctx.sync()
            if st["n_retry_slices"] == 0 or attempt == 3:
                break
        edges += st["n_edges"]
    ctx.merge_labels(d_gath, n_shards, d_out)
    ctx.sync()
    got = ctx.download_i32(d_out, n).copy()
    ctx.close()
    return got, edges


@pytest.mark.exact_edges
def test_config3_one_million_rows_max_dist_1(million_csr):
    uf, indptr, indices = _csr_prefix(million_csr, 1_000_000)
    assert len(uf) > 990_000
    l1, st1 = _lib.cluster_csr(indptr, indices, 1)
    assert st1["path"] == 1                 # short rows: the variant join up to 2M rows (round 4; the band kernels beyond)
    assert st1["n_retry_slices"] == 0
    assert st1["pairs_resolved"] == len(uf) * (len(uf) - 1) // 2
    _check_fix_point(l1)
    l2, _ = _lib.cluster_csr(indptr, indices, 2)
    _check_fix_point(l2)
    assert np.array_equal(l2[l1], l2)       # the partition at d refines the partition at d + 1
    assert _check_sampled_rows(indptr, indices, 1, l1, 48, seed=11) > 0
    _check_permutation_invariance(uf, 1, l1, seed=12)
    got8, edges8 = _sharded_labels(indptr, indices, 1, 8)  # configs[3]: row-sharded over 8 ranks, label merge
    assert np.array_equal(got8, l1)
    assert edges8 == st1["n_edges"]         # every edge found by exactly one shard
    # the band kernels with the third sort key (north_star's all-pairs design) agree at this size: an independent generator
    ctx = _lib.Context(0)
    ctx.set_candidate_path("allpairs")
    ctx.upload_csr(indptr, indices)
    d_out = ctx.alloc(4 * len(uf))
    ctx.cluster(1, d_out)
    stb = ctx.sync()
    assert stb["path"] == 0 and stb["n_work_items"] > 0 and stb["n_edges"] == st1["n_edges"]
    assert np.array_equal(ctx.download_i32(d_out, len(uf)), l1)
    ctx.close()


def test_config4_one_million_rows_max_dist_5_indels(million_indels, monkeypatch):
    uf, indptr, indices = _csr(million_indels)
    d = 5
    # the library's default for a labels-only step at max-dist >= 3: candidates of already connected rows are dropped
    l5, st5 = _lib.cluster_csr(indptr, indices, d)
    if st5["n_retry_slices"]:               # the first queue is sized for sparse graphs; sync grows it: run again clean
        l5b, st5 = _lib.cluster_csr(indptr, indices, d)
        assert np.array_equal(l5b, l5)
    _check_fix_point(l5)
    assert st5["n_connected"] > st5["n_candidates"] // 2    # the dense graph the configuration is about
    assert st5["n_edges"] + st5["n_connected"] <= st5["n_candidates"]
    assert st5["n_edges"] >= len(uf) - len(np.unique(l5))    # at least a spanning forest went through the exact check
    l4, _ = _lib.cluster_csr(indptr, indices, 4)
    assert np.array_equal(l5[l4], l5)
    assert _check_sampled_rows(indptr, indices, d, l5, 40, seed=21) > 40
    _check_permutation_invariance(uf, d, l5, seed=22)
    got8, _ = _sharded_labels(indptr, indices, d, 8)
    assert np.array_equal(got8, l5)
    # every candidate checked: same labels, and every edge is found by exactly one shard
    monkeypatch.setenv("BFK_EXACT_EDGES", "1")
    l5x, st5x = _lib.cluster_csr(indptr, indices, d)
    assert np.array_equal(l5x, l5) and st5x["n_connected"] == 0 and st5x["n_candidates"] <= st5["n_candidates"]
    assert st5x["n_edges"] > 5 * len(uf)
    got8, edges8 = _sharded_labels(indptr, indices, d, 8)
    assert np.array_equal(got8, l5)
    assert edges8 == st5x["n_edges"]
    # an INDEPENDENT candidate generator at full size: the band kernels (k_sig .. k_prefilter: every
    # pair of the (k,f,g) band through the signature levels, ~1.4e11 pair slots here) with every candidate checked exactly
    # must find the same edge set — the same labels AND the same number of edges as the prefix groups
    ctx = _lib.Context(0)
    ctx.set_candidate_path("allpairs")
    ctx.set_exact_edges(True)
    ctx.upload_csr(indptr, indices)
    d_out = ctx.alloc(4 * len(uf))
    ctx.cluster(d, d_out)
    stb = ctx.sync()
    assert stb["path"] == 0 and stb["n_work_items"] > 0 and stb["n_connected"] == 0
    assert np.array_equal(ctx.download_i32(d_out, len(uf)), l5)
    assert stb["n_edges"] == st5x["n_edges"]
    ctx.close()


@pytest.mark.exact_edges
@pytest.mark.parametrize("n_rows,join,third_key", [(590_000, True, False), (610_000, True, True), (790_000, True, True),
                                                   (815_000, True, True)])
def test_size_switches_without_knobs(million_csr, n_rows, join, third_key):
    """either side of the 600k third-key switch (the all-pairs kernels, max-dist 2) and of 800k rows, where the device-driven
    text step ends and — for rows of more than 64 tokens on average — the variant join (max-dist 1); the paths must agree
    with each other and with sampled oracle rows"""
    uf, indptr, indices = _csr_prefix(million_csr, n_rows)
    n = len(uf)
    assert (n >= 600_000) == third_key
    l1, st = _lib.cluster_csr(indptr, indices, 1)
    assert (st["n_work_items"] == 0) == join
    _check_fix_point(l1)
    ctx = _lib.Context(0)
    ctx.set_candidate_path("allpairs" if join else "join")   # the other generator, same labels and edges
    ctx.upload_csr(indptr, indices)
    d_out = ctx.alloc(4 * n)
    ctx.cluster(1, d_out)
    st_o = ctx.sync()
    assert np.array_equal(ctx.download_i32(d_out, n), l1) and st_o["n_edges"] == st["n_edges"]
    ctx.close()
    l2, st2 = _lib.cluster_csr(indptr, indices, 2)           # all-pairs with / without the third key
    assert np.array_equal(l2[l1], l2)
    # (six sampled rows per max-dist and size — 48 over the four sizes; the 1M-row tests sample 48 each: the oracle's select_ind
    # call scans every column of a length band per query length, ~1 s per sampled row at these sizes)
    _check_sampled_rows(indptr, indices, 2, l2, 6, seed=n_rows)
    _check_sampled_rows(indptr, indices, 1, l1, 6, seed=n_rows + 1)


@pytest.mark.parametrize("d,indels,n_rows", [(1, False, 50000), (2, True, 40000), (5, True, 24000)])
def test_50k_rows_against_the_full_oracle(d, indels, n_rows):
    """(max-dist 5: 24k rows, max-dist 2: 40k — the oracle's all-pairs work grows with the square of the rows and took 97 + 25 of the
    suite's 515 s at 50k; 20k rows at max-dist 2 .. 5 are compared in test_gpu_parity.py as well)"""
    kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
    uf, indptr, indices = _csr(generate_profiles(n_rows, **kw))
    got, st = _lib.cluster_csr(indptr, indices, d)
    want = orc.cluster_csr(indptr, indices, d, n_threads=CORES)["labels"]
    assert d > 3 or st["n_retry_slices"] == 0  # (a dense d = 5 graph may outgrow the first queue: recovered in slices, same labels)
    assert np.array_equal(got, want)


# ---- the one-process-per-GPU driver on the HIP path ---------------------------------------------------------------
def _driver_case():
    uf, indptr, indices = _csr(generate_profiles(30000, p_del=0.03, p_ins=0.01))
    return indptr, indices


@pytest.mark.exact_edges
@pytest.mark.parametrize("d", [1, 2])
def test_gpu_engine_world_1_matches_one_shot(d):
    import torch

    from breakfast_amd.distributed import GpuEngine, ShardedClusterer

    indptr, indices = _driver_case()
    want, st1 = _lib.cluster_csr(indptr, indices, d)
    eng = GpuEngine(0)
    sc = ShardedClusterer(eng, 0, 1)
    sc.bind(indptr, indices)
    for _ in range(3):  # steady-state steps (the join memoises the empty verify queue after the first sync)
        got = sc.step(d)
        st = eng.sync()
        torch.cuda.synchronize()
        assert np.array_equal(got[: len(want)].cpu().numpy(), want)
        assert st["n_edges"] == st1["n_edges"]


def _nccl_worker(rank, world, port, d, merge, out_dir):
    import torch
    import torch.distributed as dist

    from breakfast_amd.distributed import GpuEngine, ShardedClusterer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        indptr, indices = _driver_case()
        eng = GpuEngine(rank)
        sc = ShardedClusterer(eng, rank, world, merge, force_exchange=(world == 1))
        sc.bind(indptr, indices)
        for _ in range(2):
            got = sc.step(d)
            st = eng.sync()
        # (no device-wide synchronize here: step() orders the caller's stream behind the engine's, and .cpu() runs on the caller's)
        np.save(os.path.join(out_dir, f"labels_{rank}.npy"), got[: len(indptr) - 1].cpu().numpy())
        np.save(os.path.join(out_dir, f"edges_{rank}.npy"), np.array([st["n_edges"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.exact_edges
@pytest.mark.parametrize("d,merge", [(1, "allgather"), (2, "allgather"), (2, "allreduce")])
def test_gpu_engine_world_2_over_rccl(d, merge, tmp_path):
    """two ranks, two GPUs, RCCL label exchange: every rank ends with the 1-GPU labels (needs >= 2 devices)"""
    if _lib.load().bfk_device_count() < 2:
        pytest.skip("needs two gfx950 devices (the driver's 8-GPU node); one-device rehearsal: test below")
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    indptr, indices = _driver_case()
    want, st1 = _lib.cluster_csr(indptr, indices, d)
    mp.spawn(_nccl_worker, args=(2, port, d, merge, str(tmp_path)), nprocs=2, join=True)
    edges = 0
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"labels_{r}.npy"), want)
        edges += int(np.load(tmp_path / f"edges_{r}.npy")[0])
    assert edges == st1["n_edges"]


@pytest.mark.exact_edges
@pytest.mark.parametrize("d,merge", [(1, "allgather"), (2, "allgather"), (3, "allreduce")])
def test_world_1_over_rccl_runs_the_exchange_and_merge(d, merge, tmp_path):
    """the code an N-GPU node runs, executed once on hardware: a process group on backend "nccl" (= RCCL)
    with ONE rank, ShardedClusterer with the one-rank shortcut switched off — cluster_shard, the RCCL collective
    (all_gather_into_tensor, or all_reduce(MIN) + all_reduce(MAX) to the fix point) ordered on the engine's stream behind
    the kernels, k_merge behind the collective — and the labels equal the plain one-GPU call's"""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    indptr, indices = _driver_case()
    want, st1 = _lib.cluster_csr(indptr, indices, d)
    mp.spawn(_nccl_worker, args=(1, port, d, merge, str(tmp_path)), nprocs=1, join=True)
    assert np.array_equal(np.load(tmp_path / "labels_0.npy"), want)
    assert int(np.load(tmp_path / "edges_0.npy")[0]) == st1["n_edges"]


@pytest.mark.exact_edges
@pytest.mark.parametrize("d,world", [(1, 2), (2, 3)])
def test_gpu_engine_sharded_steps_on_one_device(d, world):
    """GpuEngine.cluster_shard + GpuEngine.merge for every rank of a `world`-rank run, executed one after the other on
    this one GPU with the all_gather done by hand: the exchange protocol of ShardedClusterer.step on the HIP engine"""
    import torch

    from breakfast_amd.distributed import GpuEngine

    indptr, indices = _driver_case()
    n = len(indptr) - 1
    want, st1 = _lib.cluster_csr(indptr, indices, d)
    engs = [GpuEngine(0) for _ in range(world)]
    local, edges = [], 0
    for r, e in enumerate(engs):
        e.bind(indptr, indices)
        lab = e.new_labels(1)
        e.cluster_shard(d, r, world, lab)
        edges += e.sync()["n_edges"]
        local.append(lab)
    gathered = torch.cat(local, dim=0).contiguous()  # what all_gather_into_tensor leaves on every rank
    for e in engs:
        out = e.new_labels(1)
        e.merge(gathered, world, out)
        e.sync()
        assert np.array_equal(out[0, :n].cpu().numpy(), want)
    assert edges == st1["n_edges"]


# ---- n_gpus > 1 behind the one-shot C-ABI and the CLI (one process, one context per device) ------------------------------
@pytest.mark.exact_edges
@pytest.mark.parametrize("d,n_gpus", [(1, 2), (2, 3), (3, 8)])
def test_cluster_csr_n_gpus_rehearsed_on_one_device(d, n_gpus, monkeypatch):
    """bfk_cluster_csr(n_gpus > 1): shards on their own contexts, label arrays gathered by peer copies, merged on the first
    device.  BFK_MULTI_ONE_DEVICE=1 puts every context on device 0 (the driver's 8-GPU node runs it on 8 devices)"""
    indptr, indices = _driver_case()
    want, st1 = _lib.cluster_csr(indptr, indices, d)
    if _lib.load().bfk_device_count() < n_gpus:
        monkeypatch.setenv("BFK_MULTI_ONE_DEVICE", "1")
    got, st = _lib.cluster_csr(indptr, indices, d, n_gpus=n_gpus)
    assert np.array_equal(got, want)
    assert st["n_edges"] == st1["n_edges"] and st["pairs_resolved"] == st1["pairs_resolved"]


def test_cluster_csr_rejects_more_gpus_than_devices():
    indptr, indices = _driver_case()
    have = _lib.load().bfk_device_count()
    with pytest.raises(_lib.BfkError) as e:
        _lib.cluster_csr(indptr, indices, 1, n_gpus=have + 1)
    assert e.value.code == -3


def test_cli_gpus_option(tmp_path, monkeypatch):
    import hashlib

    import click.testing

    from breakfast_amd import console
    from breakfast_amd.synth import generate_tsv

    inp = tmp_path / "in.tsv"
    generate_tsv(inp, 20000)
    outs = []
    for gpus in (1, 2):
        if gpus > _lib.load().bfk_device_count():
            monkeypatch.setenv("BFK_MULTI_ONE_DEVICE", "1")
        out = tmp_path / f"g{gpus}"
        res = click.testing.CliRunner().invoke(console.main, ["--input-file", str(inp), "--outdir", str(out), "--gpus", str(gpus)])
        assert res.exit_code == 0, (res.output, res.exception)
        outs.append(hashlib.sha256((out / "clusters.tsv").read_bytes()).hexdigest())
    assert outs[0] == outs[1]
