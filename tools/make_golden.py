#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, read-only).  Nothing from the
reference is copied into this repo except (a) data files its own tests hold (TSV inputs and
expected outputs -> tests/golden/ref_fixtures/) and (b) input/output VECTORS produced by
calling its functions.  The GPU box never sees the reference: tests read only tests/golden/.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--big]

--big additionally records the clusters.tsv sha256 of the 10k / 100k synthetic workloads
(the 100k reference run takes ~2 min on 8 threads).

Vectors:
  G1  ref_fixtures/            the reference's own fixture TSVs (+ CLI runs recorded in cli_runs.json)
  G2  stage_<name>.npz         per-stage vectors on synthetic inputs: CSR, n_features, neighbour
                               lists exactly as cluster_features builds them, canonical edges,
                               canonical labels, meta cluster ids, clusters.tsv bytes
  G3  kats.json["cluster"]     edge-case known answers for cluster()
  G4  kats.json["filter"]      filter_features known answers
  G5  sha256.json              clusters.tsv digests for the Appendix-A workloads
"""

from __future__ import annotations

import argparse
import gzip
import hashlib
import io
import json
import os
import pickle
import shutil
import sys
import tempfile
from contextlib import redirect_stdout
from pathlib import Path

import numpy as np
import pandas as pd

REPO = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF / "src"))
sys.path.insert(0, str(REPO))

from breakfast import breakfast as ref_bf  # noqa: E402  (the reference)
from breakfast import console as ref_console  # noqa: E402

from breakfast_amd.synth import generate_profiles  # noqa: E402

GOLD = REPO / "tests" / "golden"


def quiet(fn, *a, **kw):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def write_tsv(path, ids, feats):
    with open(path, "w", newline="") as f:
        f.write("accession\tdna_profile\n")
        for i, r in zip(ids, feats):
            f.write(f"{i}\t{r}\n")


def run_ref_cli(args):
    import click.testing

    runner = click.testing.CliRunner()
    res = runner.invoke(ref_console.main, args)
    return res


# --------------------------------------------------------------------------- G1
def g1_fixtures():
    out = GOLD / "ref_fixtures"
    out.mkdir(parents=True, exist_ok=True)
    for p in sorted((REF / "tests").glob("*.tsv")):
        shutil.copyfile(p, out / p.name)
    # record what the reference CLI writes for each fixture scenario (bytes)
    runs = {}
    scenarios = {
        "dist0": ["--input-file", "testfile.tsv", "--max-dist", "0"],
        "dist1": ["--input-file", "testfile.tsv", "--max-dist", "1"],
        "dist1_noskipdel": ["--input-file", "testfile.tsv", "--max-dist", "1", "--no-skip-del"],
        "raw_defaults": ["--input-file", "testfile.tsv", "--var-type", "raw"],
        "raw_explicit": ["--input-file", "testfile.tsv", "--trim-start", "0", "--trim-end", "0",
                         "--no-skip-del", "--no-skip-ins", "--var-type", "raw"],
        "nextclade_dist0": ["--input-file", "testfile_nextclade.tsv", "--sep2", ",", "--id-col", "seqName",
                            "--clust-col", "substitutions", "--max-dist", "0", "--var-type", "nextclade_dna"],
        "nextclade_dist1": ["--input-file", "testfile_nextclade.tsv", "--sep2", ",", "--id-col", "seqName",
                            "--clust-col", "substitutions", "--max-dist", "1", "--var-type", "nextclade_dna"],
        "dist2_mcs3": ["--input-file", "testfile.tsv", "--max-dist", "2", "--min-cluster-size", "3"],
        "dist1_mcs3_noskip": ["--input-file", "testfile.tsv", "--max-dist", "1", "--min-cluster-size", "3",
                              "--no-skip-del"],
    }
    cwd = os.getcwd()
    os.chdir(out)
    try:
        for name, args in scenarios.items():
            with tempfile.TemporaryDirectory() as td:
                res = run_ref_cli(args + ["--outdir", td])
                assert res.exit_code == 0, (name, res.output, res.exception)
                data = (Path(td) / "clusters.tsv").read_bytes()
            runs[name] = {"args": args, "clusters_tsv": data.decode(), "sha256": hashlib.sha256(data).hexdigest()}
        # cache scenarios (reference tests/test_caching.py): record final output only
        with tempfile.TemporaryDirectory() as td:
            cache = Path(td) / "cache"
            res = run_ref_cli(["--input-file", "testfile.tsv", "--outdir", td, "--output-cache", str(cache),
                               "--max-dist", "1"])
            assert res.exit_code == 0
            shutil.copyfile(cache, GOLD / "ref_cache_testfile_d1.pkl.gz")  # a cache written BY THE REFERENCE
            with gzip.open(cache, "rb") as f:
                c = pickle.load(f)
            runs["cache_init"] = {
                "neigh": [[int(x) for x in n] for n in c["neigh"]],
                "meta_id": [list(t) for t in c["meta"]["id"]],
                "meta_feature": list(c["meta"]["feature"]),
                "max_dist": int(c["max_dist"]),
            }
            for p in sorted(Path(".").glob("testfile_caching0*.tsv")):
                with tempfile.TemporaryDirectory() as td2:
                    res = run_ref_cli(["--input-file", p.name, "--outdir", td2, "--input-cache", str(cache),
                                       "--max-dist", "1"])
                    assert res.exit_code == 0, (p, res.output)
                    data = (Path(td2) / "clusters.tsv").read_bytes()
                runs["cache_" + p.stem.split("_")[1]] = {"input": p.name, "clusters_tsv": data.decode()}
        # error scenario
        res = run_ref_cli(["--input-file", "duplicate-ids.tsv", "--max-dist", "1", "--outdir", tempfile.mkdtemp()])
        runs["duplicate_ids"] = {"exit_nonzero": res.exit_code != 0, "exc_type": type(res.exception).__name__,
                                 "exc_msg": str(res.exception)}
    finally:
        os.chdir(cwd)
    (GOLD / "cli_runs.json").write_text(json.dumps(runs, indent=1))


# --------------------------------------------------------------------------- G2
def canonical_from_neigh(neigh, n):
    """canonical edge set {(i<j)} and min-index labels from the reference's neighbour lists"""
    import networkx
    from networkx.algorithms.components.connected import connected_components

    G = quiet(ref_bf._to_graph, neigh)
    labels = np.arange(n, dtype=np.int32)
    for comp in connected_components(G):
        comp = sorted(int(c) for c in comp)
        labels[comp] = comp[0]
    return labels


def stage_vectors(name, feats, sep, max_dist, min_cluster_size=2, ids=None):
    """Call the reference stage by stage on already-filtered feature strings."""
    n_in = len(feats)
    if ids is None:
        ids = [f"seq{i:07d}" for i in range(n_in)]
    meta = pd.DataFrame({"id": ids, "feature": feats})
    nodups = quiet(ref_bf.collapse_duplicates, meta)
    ufeats = list(nodups["feature"])
    group_size = np.array([len(t) for t in nodups["id"]], dtype=np.int32)
    csr = ref_bf.sparse_feature_matrix(nodups["feature"], sep)
    with tempfile.TemporaryDirectory() as td:
        cache = Path(td) / "c"
        clustered = quiet(ref_bf.cluster_features, nodups.copy(), sep, max_dist, min_cluster_size, None, cache)
        with gzip.open(cache, "rb") as f:
            neigh = pickle.load(f)["neigh"]
        with tempfile.TemporaryDirectory() as td2:
            quiet(ref_bf.write_output, clustered, meta, Path(td2))
            tsv = (Path(td2) / "clusters.tsv").read_bytes()
    n_u = len(ufeats)
    neigh_off = np.zeros(len(neigh) + 1, dtype=np.int64)
    neigh_off[1:] = np.cumsum([len(x) for x in neigh])
    neigh_flat = np.concatenate([np.asarray(x, dtype=np.int64) for x in neigh]) if neigh else np.zeros(0, np.int64)
    # canonical edges: all (i<j) that appear together as (query row, neighbour); the query row is not
    # stored by the reference, so derive edges from exact distances instead (dense, small N only)
    D = None
    if n_u <= 2500:
        from sklearn.metrics import pairwise_distances

        D = pairwise_distances(csr, metric="manhattan")
        iu = np.triu_indices(n_u, 1)
        m = D[iu] <= max_dist
        edges = np.stack([iu[0][m], iu[1][m]], axis=1).astype(np.int32)
    else:
        edges = np.zeros((0, 2), np.int32)
    labels = canonical_from_neigh(neigh, n_u)
    cid = np.array([0 if pd.isna(x) else int(x) for x in clustered["cluster_id"]], dtype=np.int32)
    np.savez_compressed(
        GOLD / f"stage_{name}.npz",
        sep=np.array(sep),
        max_dist=np.int32(max_dist),
        min_cluster_size=np.int32(min_cluster_size),
        features=np.array(feats, dtype=object) if False else np.array(feats),
        ufeatures=np.array(ufeats),
        group_size=group_size,
        indptr=csr.indptr.astype(np.int32),
        indices=csr.indices.astype(np.int32),
        n_vocab=np.int32(csr.shape[1]),
        n_features=np.asarray(clustered["n_features"]).astype(np.int64).ravel(),
        neigh_off=neigh_off,
        neigh_flat=neigh_flat,
        edges=edges,
        labels=labels,
        cluster_id=cid,
        clusters_tsv=np.frombuffer(tsv, dtype=np.uint8),
    )
    print(f"stage_{name}: N={n_in} N_u={n_u} nnz={csr.nnz} lists={len(neigh)} edges={len(edges)} "
          f"clusters={cid.max()}")


def multiset_rows(n, seed, alphabet=30, kmax=12):
    rng = np.random.default_rng(seed)
    toks = [f"T{i}" for i in range(alphabet)]
    rows = []
    base = [list(rng.choice(toks, size=int(rng.integers(0, kmax + 1)))) for _ in range(max(4, n // 10))]
    for _ in range(n):
        r = list(base[int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, 3))):
            op = rng.random()
            if op < 0.4 and r:
                r.pop(int(rng.integers(0, len(r))))
            elif op < 0.8:
                r.insert(int(rng.integers(0, len(r) + 1)), str(rng.choice(toks)))
            elif r:
                r.append(r[int(rng.integers(0, len(r)))])  # repeat a token -> multiset semantics
        if rng.random() < 0.3:
            rng.shuffle(r)
        s = " ".join(r)
        if rng.random() < 0.1:
            s = s.replace(" ", "  ", 1)  # double separator -> empty token
        rows.append(s)
    return rows


def g2_stage():
    filt = lambda rows, **kw: quiet(  # noqa: E731
        ref_bf.filter_features, rows, " ", "covsonar_dna", kw.get("skip_ins", True), kw.get("skip_del", True),
        264, 228, 29903)
    for n in (200, 2000):
        rows = generate_profiles(n)
        for d in (1, 2):
            stage_vectors(f"syn{n}_d{d}", filt(rows), " ", d)
    for n in (200, 2000):
        rows = generate_profiles(n, p_del=0.05, p_ins=0.01)
        for d in (1, 2, 5):
            stage_vectors(f"indel{n}_d{d}", filt(rows, skip_ins=False, skip_del=False), " ", d)
    rows = multiset_rows(300, 7)
    for d in (1, 2, 3):
        stage_vectors(f"multiset300_d{d}", rows, " ", d, min_cluster_size=3)
    # long rows (k up to ~700) to exercise the >64 / >256 element code paths
    rng = np.random.default_rng(11)
    base = [f"A{p}C" for p in range(1000, 1700)]
    rows = []
    for i in range(120):
        k = int(rng.choice([10, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 640]))
        sel = sorted(rng.choice(700, size=k, replace=False))
        r = [base[j] for j in sel]
        rows.append(" ".join(r))
        for _ in range(2):  # near-duplicates at distance 1..2
            r2 = list(r)
            if rng.random() < 0.5 and len(r2) > 1:
                r2.pop(int(rng.integers(0, len(r2))))
            else:
                r2.append(f"G{int(rng.integers(2000, 2100))}T")
            rows.append(" ".join(r2))
    for d in (1, 2):
        stage_vectors(f"longrows_d{d}", rows, " ", d)


# --------------------------------------------------------------------------- G3 / G4
def g3_g4_kats():
    cases = [
        (["X Y", "X X Y"], 1, 2), (["X Y", "X X X Y"], 1, 2), (["X Y", "X X X Y"], 2, 2),
        (["X Y", "Y X"], 0, 2), (["X Y", "Y X"], 1, 2),
        (["", "X", "X Y", "Q R S"], 1, 2), (["", "", "Q R S"], 1, 2), (["", "", "Q R S"], 0, 2),
        (["X", "X", "X Y", "Z"], 1, 3), (["X", "X", "X Y", "Z"], 1, 4),
        (["X  Y", "X Y"], 0, 2), (["X  Y", "X Y"], 1, 2),
        (["A", "A B", "A B C", "A B C D", "Z"], 1, 2),
        (["A B", "A C"], 1, 2), (["A B", "A C"], 2, 2),
        (["A"], 1, 1), (["A"], 1, 2), (["", ""], 1, 2), ([""], 1, 1),
        (["A B C", "A B D", "A B E", "F"], 2, 2), (["A B C", "C B A", "B A C"], 1, 2),
        (["A A", "A", "A A A", "B"], 1, 2),
        (["A,B", "A,B,C", "A"], 1, 2),
    ]
    out = []
    for feats, d, mcs in cases:
        sep = "," if any("," in f for f in feats) else " "
        ids = [f"s{i}" for i in range(len(feats))]
        meta = pd.DataFrame({"id": ids, "feature": feats})
        nod = quiet(ref_bf.collapse_duplicates, meta)
        try:
            cl = quiet(ref_bf.cluster, nod, sep, d, mcs, None, None)
        except Exception as e:  # e.g. all-empty input: scipy cannot infer the matrix shape
            out.append({"features": feats, "sep": sep, "max_dist": d, "min_cluster_size": mcs,
                        "error": type(e).__name__, "error_msg": str(e)})
            continue
        with tempfile.TemporaryDirectory() as td:
            quiet(ref_bf.write_output, cl, meta, Path(td))
            tsv = (Path(td) / "clusters.tsv").read_text()
        csr = ref_bf.sparse_feature_matrix(nod["feature"], sep)
        out.append({"features": feats, "sep": sep, "max_dist": d, "min_cluster_size": mcs, "clusters_tsv": tsv,
                    "indptr": csr.indptr.tolist(), "indices": csr.indices.tolist(),
                    "row_sums": np.asarray(csr.sum(axis=1)).ravel().tolist()})
    fcases = [
        ("covsonar_dna", True, True, 264, 228, " ", "A264C A265C A29674C A29675C"),
        ("covsonar_dna", True, True, 264, 228, " ", "C241TAT del:5:3 A100C"),
        ("covsonar_dna", False, False, 264, 228, " ", "C241TAT del:5:3 A100C"),
        ("covsonar_dna", False, False, 0, 1, " ", "S:N501Y foo a12c A12C"),
        ("covsonar_dna", False, True, 0, 0, " ", "A12C  A12C   T5G "),
        ("covsonar_dna", False, False, 0, 0, " ", "A12C  A12C   T5G "),
        ("nextclade_dna", True, True, 264, 228, ",", "C241T,11288-11297,22492,273:CTT,A500G"),
        ("nextclade_dna", False, True, 0, 0, ",", "C241T,11288-11297,22492,273:CTT,A500G"),
        ("nextclade_aa", True, True, 0, 0, " ", "S:N501Y ORF1a:T3255I S:V70- S:Y144* N:A34AK"),
        ("covsonar_aa", False, True, 0, 0, " ", "S:N501Y ORF1:del:12:7 N:A34AK xyz"),
        ("covsonar_aa", True, False, 0, 0, " ", "S:N501Y ORF1:del:12:7 N:A34AK xyz"),
        ("raw", True, True, 5, 5, " ", "anything goes  here"),
        ("covsonar_dna", False, False, 250, 0, " ", "  "),
        ("covsonar_dna", True, True, 100, 100, " ", "G24C C241T del:10:1 G533TT A990T"),
        ("covsonar_dna", True, True, 264, 228, " ", "C241T T606C del:11288:9 C13515T A29675G A29674G"),
    ]
    fout = []
    for vt, si, sd, ts, te, sep, s in fcases:
        buf = io.StringIO()
        with redirect_stdout(buf):
            r = ref_bf.filter_features([s], sep, vt, si, sd, ts, te, 29903 if ts != 100 else 1000)
        fout.append({"var_type": vt, "skip_ins": si, "skip_del": sd, "trim_start": ts, "trim_end": te,
                     "reference_length": 29903 if ts != 100 else 1000, "sep": sep, "input": s,
                     "output": list(r)[0], "stdout": buf.getvalue()})
    (GOLD / "kats.json").write_text(json.dumps({"cluster": out, "filter": fout}, indent=1))
    print(f"kats: {len(out)} cluster cases, {len(fout)} filter cases")


# --------------------------------------------------------------------------- G5
def g5_sha(big):
    res = {}
    sizes = [2000] + ([10000, 100000] if big else [])
    path = GOLD / "sha256.json"
    if path.exists():
        res = json.loads(path.read_text())
    os.environ["OMP_NUM_THREADS"] = "8"
    for n in sizes:
        with tempfile.TemporaryDirectory() as td:
            inp = Path(td) / "in.tsv"
            rows = generate_profiles(n)
            write_tsv(inp, [f"seq{i:07d}" for i in range(n)], rows)
            h_in = hashlib.sha256(inp.read_bytes()).hexdigest()
            r = run_ref_cli(["--input-file", str(inp), "--outdir", td, "--max-dist", "1", "--jobs", "8"])
            assert r.exit_code == 0, r.output
            data = (Path(td) / "clusters.tsv").read_bytes()
        df = pd.read_table(io.BytesIO(data))
        res[f"syn{n}_d1"] = {"n": n, "seed": 20240601, "input_sha256": h_in,
                             "clusters_sha256": hashlib.sha256(data).hexdigest(),
                             "n_clusters": int(df["cluster_id"].max()),
                             "n_unclustered": int(df["cluster_id"].isna().sum())}
        print(n, res[f"syn{n}_d1"])
        path.write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    GOLD.mkdir(parents=True, exist_ok=True)
    steps = {"g1": g1_fixtures, "g2": g2_stage, "g34": g3_g4_kats, "g5": lambda: g5_sha(a.big)}
    for k, fn in steps.items():
        if not a.only or k in a.only.split(","):
            fn()
