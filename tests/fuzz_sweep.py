"""One-off extended fuzz sweep (more seeds than the test suite runs): device tokeniser, fused text -> labels, the clustering
kernels with every candidate generator, and the device prepare, each against the oracle / the host stage.
Run on a GPU box:  python tests/fuzz_sweep.py [first_seed] [n_seeds]   (not collected by pytest; lives here because only tests/
may use the oracle)"""
import os
import sys
from pathlib import Path

import numpy as np

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))  # (conftest helpers and the fuzz generators of the suite)
from oracle import ref_port as orc  # noqa: E402
from test_frontend import OPTS, _fuzz_tokens  # noqa: E402
from test_gpu_parity import fuzz_case  # noqa: E402
from test_gpu_prep import VAR_TYPES, _structured_tokens, assert_same, both, host_invalid_tokens  # noqa: E402

from breakfast_amd import _lib  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(770000 + seed)
    # 1. text -> CSR, and text -> labels
    sep = [" ", ",", ";", "\t", "|", "-", ", ", "--", "A1", " | ", "::"][seed % 11]   # (several bytes: folded on the device, k_sepfold)
    na = int(rng.choice([5, 50, 500, 5000]))
    alphabet = [f"{chr(65 + int(a))}{int(p)}{chr(65 + int(b))}" * int(rng.integers(1, 3)) for a, p, b in
                zip(rng.integers(0, 26, na), rng.integers(1, 30000, na), rng.integers(0, 26, na))]
    rows = []
    for _ in range(int(rng.integers(1, 4000))):
        k = int(rng.integers(0, int(rng.choice([4, 60, 200]))))
        toks = [alphabet[int(i)] for i in rng.integers(0, len(alphabet), k)]
        if rng.random() < 0.2:
            toks += [""] * int(rng.integers(1, 3))
            rng.shuffle(toks)
        rows.append(sep.join(toks))
    want = orc.sparse_feature_matrix(rows, sep)
    got = _lib.build_csr_device(rows, sep)
    if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and int(got[2]) == int(want[2])):
        print("TOKENISER MISMATCH seed", seed)
        bad += 1
    buf, off = _lib.pack_rows(rows)
    for d in (1, int(rng.integers(0, 4))):
        lab = _lib.cluster_text(buf, off, sep, d)[0]
        wl = orc.cluster_csr(want[0], want[1], d, n_threads=4)["labels"]
        if not np.array_equal(lab, wl):
            print("CLUSTER_TEXT MISMATCH seed", seed, "d", d)
            bad += 1
    # 2. clustering kernels, every generator
    for it in range(4):
        _, indptr, indices, d, _ = fuzz_case(rng)
        for dd in (d, 1):
            wl = orc.cluster_csr(indptr, indices, dd, n_threads=4)["labels"]
            for env in ({}, {"BFK_JOIN": "0", "BFK_PG": "0"}, {"BFK_PG": "1"}, {"BFK_EXACT_EDGES": "1"}):
                os.environ.update(env)
                try:
                    lab, st = _lib.cluster_csr(indptr, indices, dd)
                finally:
                    for k in env:
                        del os.environ[k]
                if not np.array_equal(lab, wl):
                    print("CLUSTER MISMATCH seed", seed, "it", it, "d", dd, env)
                    bad += 1
    # 3. device prepare (filter + collapse + CSR) against the host stage
    var_type = VAR_TYPES[int(rng.integers(len(VAR_TYPES)))]
    opts = OPTS[int(rng.integers(len(OPTS)))]
    toks = sorted({t for t in _fuzz_tokens(rng, 400) + _structured_tokens(rng, 400) if t and " " not in t})
    filtering = opts[0] or opts[1] or opts[2] > 0 or opts[3] > 0
    badt = host_invalid_tokens(toks, var_type, opts) if filtering else set()
    good = [t for t in toks if t not in badt]
    if good:
        sep2 = [" ", " ", ", ", "::", " | ", ";;"][seed % 6]   # (tokens may hold pieces of the separator: both stages split the same string)
        feats = [sep2.join(good[int(rng.integers(len(good)))] for _ in range(int(rng.integers(0, 12)))) for _ in range(int(rng.integers(1, 1500)))]
        feats += feats[: len(feats) // 3] + ["", sep2, sep2 + sep2, sep2[:1]]
        rng.shuffle(feats)
        try:
            assert_same(*both([f"s{i}" for i in range(len(feats))], feats, sep2, var_type, opts))
        except AssertionError as e:
            print("PREPARE MISMATCH seed", seed, var_type, opts, str(e)[:200])
            bad += 1
    if seed % 10 == 9:
        print("seed", seed, "done, mismatches so far:", bad, flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
