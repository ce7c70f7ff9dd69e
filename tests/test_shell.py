"""CPU: the Python shell around the hot path (reader, filter, collapse, writer, CLI plumbing) against the
reference's fixtures and the golden vectors.  Nothing here needs a GPU: max-dist 0 never reaches libbfk."""

import io
import os
from contextlib import redirect_stdout
from pathlib import Path

import click.testing
import numpy as np
import pandas as pd
import pytest
from conftest import GOLD, ROOT

from breakfast_amd import breakfast, console

FIX = GOLD / "ref_fixtures"


@pytest.fixture
def runner():
    return click.testing.CliRunner()


def test_entrypoint(runner):
    assert runner.invoke(console.main, ["--help"]).exit_code == 0


def test_filter_kats(kats):
    for c in kats["filter"]:
        buf = io.StringIO()
        with redirect_stdout(buf):
            r = breakfast.filter_features([c["input"]], c["sep"], c["var_type"], c["skip_ins"], c["skip_del"],
                                          c["trim_start"], c["trim_end"], c["reference_length"])
        assert list(r)[0] == c["output"], c
        assert buf.getvalue() == c["stdout"], c


@pytest.mark.parametrize("features,expected,args", [
    (["C241T"], ["C241T"], (False, False, 0, 0, 1000)),
    (["C241T"], [""], (False, False, 250, 0, 1000)),
    ([""], [""], (False, False, 250, 0, 1000)),
    (["  "], [""], (False, False, 250, 0, 1000)),
    (["C241T del:10:1 G5343TT"], ["C241T G5343TT"], (False, True, 0, 0, 1000)),
    (["C241T del:10:1 G5343TT"], ["C241T del:10:1"], (True, False, 0, 0, 1000)),
    (["C241T del:10:1 G5343TT"], ["C241T"], (True, True, 0, 0, 1000)),
    (["G24C C241T del:10:1 G533TT A990T"], ["C241T"], (True, True, 100, 100, 1000)),
])
def test_filter_like_reference_unit_tests(features, expected, args):
    """the cases of the reference's tests/test_filtering.py:4-71"""
    with redirect_stdout(io.StringIO()):
        assert breakfast.filter_features(features, " ", "covsonar_dna", *args)[0] == expected[0]


def test_filter_aa_like_reference_unit_tests():
    with redirect_stdout(io.StringIO()):
        assert breakfast.filter_features(["S:N501Y ORF1:del:12:7 N:A34AK"], " ", "covsonar_aa", True, True, 0, 0,
                                         1000)[0] == "S:N501Y"
        assert breakfast.filter_features(["S:N501Y S:V70-"], " ", "nextclade_aa", False, True, 0, 0,
                                         1000)[0] == "S:N501Y"


def test_filter_identity_returns_same_object():
    feats = ["A12C  A12C   T5G "]
    assert breakfast.filter_features(feats, " ", "covsonar_dna", False, False, 0, 0, 29903) is feats


def test_filter_bad_type_exits():
    with pytest.raises(SystemExit), redirect_stdout(io.StringIO()):
        breakfast.filter_features(["A1C"], " ", "nope", True, True, 0, 0, 100)


def test_sparse_feature_matrix_ignores_empty_features():
    m = breakfast.sparse_feature_matrix(["", "C241T"], " ")
    assert m.shape == (2, 1) and m[0].count_nonzero() == 0 and m[1].count_nonzero() == 1


def test_read_input_and_duplicates():
    with redirect_stdout(io.StringIO()):
        meta = breakfast.read_input(FIX / "testfile.tsv", "\t", "accession", "dna_profile")
    assert list(meta.columns) == ["id", "feature"] and len(meta) == 7
    with pytest.raises(ValueError, match="Duplicate sequence identifiers"), redirect_stdout(io.StringIO()):
        breakfast.read_input(FIX / "duplicate-ids.tsv", "\t", "accession", "dna_profile")


def test_read_input_na_rules(tmp_path):
    p = tmp_path / "x.tsv"
    p.write_text('accession\tdna_profile\nNA2\tA1C\n"q 1"\t\nz\tnull\n')
    with redirect_stdout(io.StringIO()):
        meta = breakfast.read_input(p, "\t", "accession", "dna_profile")
    assert meta["feature"].tolist() == ["A1C", "", ""] and meta["id"].tolist() == ["NA2", "q 1", "z"]


def test_collapse_and_write_roundtrip(tmp_path):
    meta = pd.DataFrame({"id": ["a", "b", "c", "d", "e"], "feature": ["X", "Y", "X", "Z", "Y"]})
    with redirect_stdout(io.StringIO()):
        nod = breakfast.collapse_duplicates(meta)
    assert nod["id"].tolist() == [("a", "c"), ("b", "e"), ("d",)]
    nod["cluster_id"] = pd.array([7, pd.NA, 3], dtype="Int64").astype(object)
    breakfast.write_output(nod, meta, tmp_path)
    assert (tmp_path / "clusters.tsv").read_bytes() == b"id\tcluster_id\na\t1\nb\t\nc\t1\nd\t2\ne\t\n"


@pytest.mark.parametrize("scenario", ["dist0", "nextclade_dist0"])
def test_cli_dist0_bytes(runner, tmp_path, monkeypatch, cli_runs, scenario):
    """max-dist 0 end to end (no GPU involved): byte-identical to what the reference CLI wrote"""
    monkeypatch.chdir(FIX)
    run = cli_runs[scenario]
    res = runner.invoke(console.main, run["args"] + ["--outdir", str(tmp_path)])
    assert res.exit_code == 0, res.output
    assert (tmp_path / "clusters.tsv").read_text() == run["clusters_tsv"]
    exp = pd.read_table(FIX / "expected_clusters_dist0.tsv", sep="\t")
    assert exp.equals(pd.read_table(tmp_path / "clusters.tsv", sep="\t"))


def test_cli_utf8_ids_take_the_native_path(runner, tmp_path, monkeypatch):
    """accession names with accents / CJK / emoji no longer push a run to the pandas reader — the native
    path reads, collapses and writes them (stdout and clusters.tsv byte-equal to the mirror's); invalid UTF-8 still goes to
    pandas, which raises the reference's UnicodeDecodeError"""
    from breakfast_amd import fastpath

    names = ["hCoV-19/C\u00f4te d\u2019Ivoire/{}", "M\u00fcnchen-{}", "\u6771\u4eac-{}", "\U0001f9ec{}", "plain{}"]
    rows = [(names[i % 5].format(i), f"A{300 + i % 4}T C{400 + i % 3}G del:5:3") for i in range(200)]
    inp = tmp_path / "in.tsv"
    inp.write_text("accession\tdna_profile\n" + "".join(f"{a}\t{f}\n" for a, f in rows), encoding="utf-8")
    calls = []
    real = fastpath.run
    monkeypatch.setattr(fastpath, "run", lambda *a, **k: calls.append(real(*a, **k)) or calls[-1])
    res = runner.invoke(console.main, ["--input-file", str(inp), "--max-dist", "0", "--outdir", str(tmp_path / "native")])
    assert res.exit_code == 0, (res.output, res.exception)
    assert calls == [True]                       # the fast path took the run
    monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    ref = runner.invoke(console.main, ["--input-file", str(inp), "--max-dist", "0", "--outdir", str(tmp_path / "mirror")])
    assert ref.exit_code == 0, (ref.output, ref.exception)
    a, b = (tmp_path / "native" / "clusters.tsv").read_bytes(), (tmp_path / "mirror" / "clusters.tsv").read_bytes()
    assert a == b and "M\u00fcnchen-1\t".encode() in a
    strip = lambda out: [ln for ln in out.splitlines() if "outdir" not in ln.lower()]
    assert strip(res.output) == strip(ref.output)
    monkeypatch.delenv("BFK_NO_FASTPATH")
    inp.write_bytes(inp.read_bytes().replace("M\u00fcnchen-1\t".encode(), b"M\xfcnchen-1\t"))   # Latin-1: not UTF-8
    calls.clear()
    bad = runner.invoke(console.main, ["--input-file", str(inp), "--max-dist", "0", "--outdir", str(tmp_path / "bad")])
    assert calls == [False] and bad.exit_code != 0 and isinstance(bad.exception, UnicodeDecodeError)


def test_cli_errors(runner, tmp_path, monkeypatch):
    monkeypatch.chdir(FIX)
    base = ["--outdir", str(tmp_path), "--max-dist", "0"]
    assert runner.invoke(console.main, ["--input-file", "duplicate-ids.tsv"] + base).exit_code != 0
    assert runner.invoke(console.main, ["--input-file", "testfile.tsv", "--clust-col", "missing"] + base).exit_code != 0
    assert runner.invoke(console.main, ["--input-file", "testfile.tsv", "--id-col", "missing"] + base).exit_code != 0
    assert runner.invoke(console.main, ["--input-file", "testfile.tsv", "--var-type", "raw", "--trim-start", "5"]
                         + base).exit_code != 0
    assert runner.invoke(console.main, ["--input-file", "testfile.tsv", "--var-type", "covsonar_aa", "--skip-del"]
                         + base).exit_code != 0
    assert runner.invoke(console.main, ["--input-file", "testfile.tsv", "--trim-start", "40000"] + base).exit_code != 0


def test_cluster_identical_features_min_size():
    meta = pd.DataFrame({"id": [("a", "b"), ("c",), ("d", "e", "f")], "feature": ["X", "Y", "Z"]})
    with redirect_stdout(io.StringIO()):
        out = breakfast.cluster(meta, " ", 0, 2, None, None)
    assert out["cluster_id"].tolist()[0] == 1 and pd.isna(out["cluster_id"].tolist()[1])
    assert out["cluster_id"].tolist()[2] == 2


def test_cache_host_logic(tmp_path):
    """map_features / update_neighbours / save+load (no GPU): rows are matched by feature string, deleted rows
    drop out of cached lists, new rows come back in the order the reference visits them"""
    from breakfast_amd import cache as ca

    c2n, new_rows = ca.map_features(["A", "B", "C", "D"], ["C", "zz", "A", "E", "D"])
    assert c2n.tolist() == [2, -1, 0, 4] and new_rows.tolist() == [3, 1]  # 'E' < 'zz'
    neigh = ca.update_neighbours([np.array([0, 1]), np.array([1]), np.array([2, 3, 0])], c2n)
    assert [x.tolist() for x in neigh] == [[2], [0, 4, 2]]
    meta = pd.DataFrame({"id": [("a",), ("b", "c")], "feature": ["X", "Y"], "n_features": [1, 1]})
    with redirect_stdout(io.StringIO()):
        ca.save(tmp_path / "d" / "c.gz", [np.array([0, 1])], meta, 2)
        got = ca.load(tmp_path / "d" / "c.gz", 2)
        assert got["max_dist"] == 2 and list(got["meta"].columns) == ["id", "feature"]
        with pytest.raises(ca.CacheMismatch):
            ca.load(tmp_path / "d" / "c.gz", 1)


def test_reference_cache_file_is_readable():
    from breakfast_amd import cache as ca

    with redirect_stdout(io.StringIO()):
        c = ca.load(GOLD / "ref_cache_testfile_d1.pkl.gz", 1)
    assert len(c["neigh"]) == 5 and list(c["meta"].columns) == ["id", "feature"]


def _cli_process(args, cwd):
    """the CLI as a real process (python -m breakfast_amd): exit status and stderr as a shell sees them"""
    import subprocess
    import sys

    return subprocess.run([sys.executable, "-m", "breakfast_amd", *args], cwd=str(cwd), capture_output=True, text=True,
                          env={**os.environ, "PYTHONPATH": str(ROOT)})


@pytest.mark.parametrize("args,ok", [
    (["--input-file", "testfile.tsv", "--max-dist=0"], True),            # '=' form: no preload, d = 0 path
    (["--input-file", "testfile.tsv", "--max-dist", "0"], True),
    (["--input-file", "duplicate-ids.tsv", "--max-dist", "0"], False),   # declined by the native reader -> the reference's ValueError
    (["--input-file", "testfile.tsv", "--id-col", "nope", "--max-dist", "0"], False),
    (["--input-file", "testfile.tsv", "--id-col", "nope"], False),       # preload started, the run dies in the reader
    (["--input-file", "testfile.tsv", "--help"], True),
])
def test_cli_process_never_aborts(tmp_path, args, ok):
    """every exit path joins the preload thread: the process ends with the CLI's own status — never SIGABRT (134) with
    'terminate called without an active exception' from a still-joinable std::thread"""
    res = _cli_process(args + ["--outdir", str(tmp_path / "o")], FIX)
    assert "terminate called" not in res.stderr, res.stderr[-500:]
    assert res.returncode not in (134, -6), (res.returncode, res.stderr[-500:])
    assert (res.returncode == 0) == ok, (res.returncode, res.stderr[-500:])


def test_cli_process_fast_exit_loses_nothing(tmp_path):
    """a successful run ends by os._exit after flushing (no interpreter / HIP teardown): the prints a pipe receives and the
    files are those of an ordinary exit (BFK_FAST_EXIT=0), the status is 0; a failing run ends the ordinary way"""
    import subprocess
    import sys

    outs = {}
    for mode in ("1", "0"):
        o = tmp_path / f"o{mode}"
        res = subprocess.run([sys.executable, "-m", "breakfast_amd", "--input-file", "testfile.tsv", "--max-dist", "0", "--outdir", str(o)],
                             cwd=str(FIX), capture_output=True, text=True, env={**os.environ, "PYTHONPATH": str(ROOT), "BFK_FAST_EXIT": mode})
        assert res.returncode == 0, res.stderr[-500:]
        outs[mode] = (res.stdout.replace(str(o), "OUT"), (o / "clusters.tsv").read_bytes())
    assert outs["1"] == outs["0"]
    assert "Number of clusters found" in outs["1"][0]
    res = _cli_process(["--input-file", "testfile.tsv", "--id-col", "nope", "--max-dist", "0", "--outdir", str(tmp_path / "x")], FIX)
    assert res.returncode != 0


def test_cli_process_max_dist_1_without_a_gpu_fails_loudly_not_by_abort(tmp_path):
    """with the preload thread running (max-dist 1): on a box without a GPU the run reports BFK_ENODEV and exits non-zero;
    on a GPU box it succeeds — either way through a normal exit"""
    res = _cli_process(["--input-file", "testfile.tsv", "--outdir", str(tmp_path / "o")], FIX)
    assert "terminate called" not in res.stderr and res.returncode not in (134, -6), (res.returncode, res.stderr[-500:])
    import torch

    assert (res.returncode == 0) == torch.cuda.is_available(), res.stderr[-500:]
