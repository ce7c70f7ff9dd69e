"""Soak of the bound-CSR entry points (development tooling; the oracle is the checker): random CSRs of random shapes on one context,
runs at random max-dist with the candidate generator forced or automatic, exact edges on and off, several runs without a sync
in between (the same run again, other labels, another max-dist), re-binds behind unread runs, the one-shot entries
(bfk_cluster_csr, bfk_neighbours_csr with and without select_ind, bfk_labels_from_lists) in between; labels and lists against
the oracle's.
usage (GPU box): python tools/soak_csr.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from breakfast_amd import _lib  # noqa: E402
from breakfast_amd.synth import generate_family, generate_profiles  # noqa: E402
from oracle import ref_port as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
POOL = {"default": list(dict.fromkeys(generate_profiles(20000, seed=seed))),
        "indels": list(dict.fromkeys(generate_profiles(10000, seed=seed + 1, p_del=0.05, p_ins=0.01))),
        "long": list(dict.fromkeys(generate_family("long", 3000, seed=seed + 2))),
        "star": list(dict.fromkeys(generate_family("star", 10000, seed=seed + 3))),
        "aa": list(dict.fromkeys(generate_family("aa", 6000, seed=seed + 4)))}


def rows_of():
    kind = str(rng.choice(["default", "default", "indels", "long", "star", "aa", "identical", "tiny", "shuffled"]))
    if kind in POOL:
        src = POOL[kind]
        n = int(rng.integers(1, min(len(src), 9000)))
        a = int(rng.integers(0, len(src) - n + 1))
        return kind, src[a:a + n]
    if kind == "identical":
        src = POOL["default"]
        n = int(rng.integers(300, 2500))
        k = int(rng.integers(1, n))
        return kind, src[:k] + [src[k]] * int(rng.integers(150, 400)) + src[k:n]
    if kind == "tiny":
        return kind, [["A1C", "A1C G2T", "G2T", "Q9R A1C G2T", "X", "X X", "X Y"][int(i)] for i in rng.integers(0, 7, size=int(rng.integers(1, 12)))]
    src = POOL["default"]   # the same multisets in another token order (the positional certificate cannot decide them)
    n = int(rng.integers(200, 2000))
    out = []
    for r in src[:n]:
        t = r.split(" ")
        out.append(" ".join(t[i] for i in rng.permutation(len(t))) if rng.random() < 0.3 else r)
    return kind, out


oracle_cache = {}


def want(key, ip, ix, d):
    k = (key, d)
    if k not in oracle_cache:
        oracle_cache[k] = orc.cluster_csr(ip, ix, d, n_threads=16)
    return oracle_cache[k]


def brute_lists(ip, ix, d):
    """every row's neighbours within d (itself included), ascending — from the dense distance matrix of the third-party kernel the
    reference calls (small inputs only)"""
    from scipy.sparse import csr_matrix
    from sklearn.metrics.pairwise import manhattan_distances

    # (copies: sum_duplicates works in place on the arrays the matrix was built from)
    x = csr_matrix((np.ones(len(ix), dtype=np.int64), ix.copy(), ip.copy()), shape=(len(ip) - 1, int(ix.max()) + 1))
    x.sum_duplicates()
    dist = manhattan_distances(x)
    return [np.flatnonzero(row <= d) for row in dist]


import os  # noqa: E402

os.environ.setdefault("BFK_MULTI_ONE_DEVICE", "1")
os.environ.setdefault("BFK_MULTI_FORCE", "1")
t_end = time.time() + budget
n_runs = n_binds = 0
ctx = _lib.Context(0)
while time.time() < t_end:
    kind, rows = rows_of()
    ip, ix, _ = _lib.build_csr(rows, " ")
    if len(ix) == 0:
        continue
    n = len(rows)
    key = n_binds
    oracle_cache = {}
    n_binds += 1
    if rng.random() < 0.15:
        ctx.close()
        ctx = _lib.Context(0)
    ctx.upload_csr(ip, ix)
    bufs = [ctx.alloc(4 * n) for _ in range(3)]
    pending = []   # (buffer, d) whose labels are due at the next sync
    for _ in range(int(rng.integers(1, 7))):
        d = int(rng.choice([1, 1, 2, 2, 3, 4, 5]))
        ctx.set_candidate_path(str(rng.choice(["auto", "auto", "allpairs", "join" if d == 1 else "auto", "prefix" if d >= 2 else "auto"])))
        ctx.set_exact_edges(bool(rng.integers(0, 2)))
        b = int(rng.integers(0, 3))
        for _ in range(int(rng.integers(1, 3))):   # (the same run twice: nothing to settle in between)
            ctx.cluster(d, bufs[b])
        pending = [(pb, pd) for pb, pd in pending if pb != b] + [(b, d)]
        n_runs += 1
        if rng.random() < 0.4:
            ctx.sync()
            for pb, pd in pending:
                got = ctx.download_i32(bufs[pb], n)
                assert np.array_equal(got, want(key, ip, ix, pd)["labels"]), ("labels differ", kind, n, pd, seed, n_binds)
            pending = []
    if rng.random() < 0.5:   # a re-bind behind unread runs: the next loop's upload settles them
        ctx.sync()
    for pb, pd in pending:
        if rng.random() < 0.5:
            ctx.sync()
        got = ctx.download_i32(bufs[pb], n)   # (a download is an entry point too: synchronous on the stream)
        ctx.sync()
        got = ctx.download_i32(bufs[pb], n)
        assert np.array_equal(got, want(key, ip, ix, pd)["labels"]), ("labels differ", kind, n, pd, seed, n_binds)
    # one-shot entries on the default context
    if rng.random() < 0.35:
        d = int(rng.choice([1, 2, 3]))
        lab, st = _lib.cluster_csr(ip, ix, d)
        assert np.array_equal(lab, want(key, ip, ix, d)["labels"]), ("cluster_csr", kind, n, d, seed)
    if rng.random() < 0.25:   # the one-process multi-device driver, every context on this device (BFK_MULTI_ONE_DEVICE / _FORCE, set below)
        d = int(rng.choice([1, 2, 3, 5]))
        g = int(rng.choice([2, 3, 4]))
        lab, st = _lib.cluster_csr(ip, ix, d, n_gpus=g)
        assert np.array_equal(lab, want(key, ip, ix, d)["labels"]), ("cluster_csr n_gpus", kind, n, d, g, seed)
    if rng.random() < 0.5 and n <= 2500:
        d = int(rng.choice([1, 2, 3]))
        sel = None if rng.random() < 0.4 else rng.integers(0, n, size=int(rng.integers(1, max(2, n // 3)))).astype(np.int64)  # (unsorted, repeats)
        ptr, idx = _lib.neighbours_csr(ip, ix, d, sel)
        w = brute_lists(ip, ix, d)
        q = range(n) if sel is None else sel.tolist()
        for pos, r in enumerate(q):
            assert idx[ptr[pos]: ptr[pos + 1]].tolist() == w[r].tolist(), ("neighbour list", kind, n, d, r, seed)
        if sel is None:
            lab = _lib.labels_from_csr(n, ptr, idx)
            assert np.array_equal(lab, want(key, ip, ix, d)["labels"]), ("labels_from_lists", kind, n, d, seed, rows if n < 30 else None)
    if n_binds % 10 == 0:
        print(f"[soak_csr] {n_binds} CSRs, {n_runs} runs checked", flush=True)
ctx.close()
print(f"[soak_csr] done: {n_binds} CSRs, {n_runs} runs, all equal the oracle's")
