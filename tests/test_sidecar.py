"""The side-car cache container (breakfast_amd/sidecar.py): CPU tests of its host logic (hashes, row matching, list update,
file round trip, reading the reference's pickle) and — marked gpu — the reference's 8 cache scenarios through it."""

import json

import numpy as np
import pandas as pd
import pytest
from conftest import GOLD

from breakfast_amd import _lib, cache as ca, sidecar

FIX = GOLD / "ref_fixtures"
CACHE_INPUTS = ["AddedSeqs", "DisorderedOnly", "AddedAndDisorderedSeqs", "DeletedSingleSeq", "DeletedProfile", "ModifiedSeqs",
                "MultipleTests", "NoChanges"]


def test_hashes_identify_feature_strings():
    feats = ["A1C G2T", "", "A1C G2T", "A1C  G2T", "G2T A1C", "x" * 1000, "x" * 1001]
    h = _lib.hash_rows(feats)
    assert h.shape == (7, 2) and (h[0] == h[2]).all()
    assert len({tuple(r) for r in h.tolist()}) == 6  # every other pair differs (in both words, as it happens)
    # the table's hashes are those of the filtered feature strings it would hand out
    t = _lib.Table.from_lists([f"s{i}" for i in range(5)], ["A1C G2T", "G2T", "A1C G2T", "A1C G2T Q", ""])
    t.prepare(" ", "raw", False, False, 0, 0, 29903)
    assert np.array_equal(t.feature_hashes(), _lib.hash_rows(t.features()))
    t2 = _lib.Table.from_lists([f"s{i}" for i in range(4)], ["A1C C241T del:5:3", "A1C", "A300C A1C", "A300C  A1C"])
    t2.prepare(" ", "covsonar_dna", True, True, 264, 228, 29903)  # filtered: the strings are rebuilt from the kept tokens
    assert np.array_equal(t2.feature_hashes(), _lib.hash_rows(t2.features()))


def test_match_and_update_follow_the_reference_semantics():
    cached = _lib.hash_rows(["a", "b", "c", "d"])
    new = _lib.hash_rows(["d", "x", "b"])
    c2n = sidecar.match_rows(cached, new)
    assert c2n.tolist() == [-1, 2, -1, 0]
    # lists [a b], [c], [b c d], [d]: rows that are gone leave their lists, a list that is left empty is dropped (cache.py:51-71)
    off = np.array([0, 2, 3, 6, 7], dtype=np.int64)
    flat = np.array([0, 1, 2, 1, 2, 3, 3], dtype=np.int32)
    o2, f2 = sidecar.update_lists(off, flat, c2n)
    assert o2.tolist() == [0, 1, 3, 4] and f2.tolist() == [2, 2, 0, 0]
    want = ca.update_neighbours([flat[off[i]: off[i + 1]] for i in range(4)], c2n)
    assert [x.tolist() for x in want] == [f2[o2[i]: o2[i + 1]].tolist() for i in range(len(o2) - 1)]
    assert sidecar.match_rows(cached, _lib.hash_rows([])).tolist() == [-1] * 4


def test_file_round_trip_and_magic(tmp_path, capsys):
    h = _lib.hash_rows(["a", "b"])
    off = np.array([0, 2, 3], dtype=np.int64)
    flat = np.array([0, 1, 1], dtype=np.int32)
    p = tmp_path / "deep" / "c.bfkc"
    sidecar.save(p, 3, h, off, flat)
    assert sidecar.is_sidecar(p) and not sidecar.is_sidecar(GOLD / "ref_cache_testfile_d1.pkl.gz") and not sidecar.is_sidecar(tmp_path / "nope")
    h2, o2, f2 = sidecar.load(p, 3)
    assert np.array_equal(h, h2) and np.array_equal(off, o2) and np.array_equal(flat, f2)
    with pytest.raises(ca.CacheMismatch):  # another max-dist: the reference's warning, then a full computation
        sidecar.load(p, 1)
    assert "differnt max-dist" in capsys.readouterr().out
    p.write_bytes(p.read_bytes()[:-2])
    with pytest.raises(ValueError):
        sidecar.load(p, 3)
    assert sidecar.wants_sidecar(None, tmp_path / "x.bfkc") and not sidecar.wants_sidecar(None, tmp_path / "x.pkl.gz")


def test_damaged_sidecar_files_are_refused(tmp_path):
    """a short header, negative counts, offsets that are not a non-decreasing run from 0 to `total`, members that are no rows
    of the cached input: ValueError every time, never a clustering from wrapped indices"""
    h = _lib.hash_rows(["a", "b", "c"])
    good_off, good_flat = np.array([0, 2, 3], dtype=np.int64), np.array([0, 1, 2], dtype=np.int32)

    def write(name, head, off, flat):
        p = tmp_path / name
        with open(p, "wb") as f:
            f.write(sidecar.MAGIC)
            f.write(head)
            np.ascontiguousarray(h, dtype="<u8").tofile(f)
            np.ascontiguousarray(off, dtype="<i8").tofile(f)
            np.ascontiguousarray(flat, dtype="<i4").tofile(f)
        return p

    head = sidecar._HEAD.pack(1, 3, 2, 3)
    assert sidecar.load(write("ok.bfkc", head, good_off, good_flat), 1)[2].tolist() == [0, 1, 2]
    cases = {
        "short_header.bfkc": (head[:10], good_off, good_flat),
        "negative_rows.bfkc": (sidecar._HEAD.pack(1, -3, 2, 3), good_off, good_flat),
        "negative_total.bfkc": (sidecar._HEAD.pack(1, 3, 2, -1), good_off, good_flat),
        "off_not_from_zero.bfkc": (head, np.array([1, 2, 3]), good_flat),
        "off_decreasing.bfkc": (sidecar._HEAD.pack(1, 3, 3, 3), np.array([0, 2, 1, 3]), good_flat),
        "off_short_of_total.bfkc": (head, np.array([0, 1, 2]), good_flat),
        "member_negative.bfkc": (head, good_off, np.array([0, -1, 2])),
        "member_too_large.bfkc": (head, good_off, np.array([0, 1, 3])),
    }
    for name, (hd, off, flat) in cases.items():
        with pytest.raises(ValueError):
            sidecar.load(write(name, hd, off, flat), 1)


def test_reference_pickle_reads_as_flat_arrays(cli_runs):
    """a cache written by the reference itself (tests/golden/ref_cache_testfile_d1.pkl.gz), loaded as the side-car's arrays"""
    h, off, flat = sidecar._load_any(GOLD / "ref_cache_testfile_d1.pkl.gz", 1)
    ref = cli_runs["cache_init"]
    assert np.array_equal(h, _lib.hash_rows(ref["meta_feature"]))
    assert [flat[off[i]: off[i + 1]].tolist() for i in range(len(off) - 1)] == ref["neigh"]


# ---- the reference's cache scenarios through the side-car (GPU: the lists and the components come from the device) -----------
def _cli(args):
    import click.testing

    from breakfast_amd import console

    res = click.testing.CliRunner().invoke(console.main, args)
    assert res.exit_code == 0, (res.output, res.exception)
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("reader", ["native", "pandas"])
@pytest.mark.parametrize("start", ["sidecar", "reference_pickle"])
@pytest.mark.parametrize("idx,name", list(enumerate(CACHE_INPUTS, 1)))
def test_cache_scenarios_through_the_sidecar(idx, name, start, reader, cli_runs, tmp_path, monkeypatch):
    """test_caching.py:58-103 with the side-car as the container: written by this build from testfile.tsv, or a pickle the
    reference wrote continued as a side-car; both host paths; clusters.tsv equals what the reference produced — and the
    side-car written by the run is itself a valid starting point for the same input again"""
    if reader == "pandas":
        monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    monkeypatch.chdir(FIX)
    if start == "sidecar":
        cache = tmp_path / "c0" / "init.bfkc"
        _cli(["--input-file", "testfile.tsv", "--outdir", str(tmp_path / "init"), "--output-cache", str(cache), "--max-dist", "1"])
        assert sidecar.is_sidecar(cache)
        assert (tmp_path / "init" / "clusters.tsv").read_text() == cli_runs["dist1"]["clusters_tsv"]
    else:
        cache = GOLD / "ref_cache_testfile_d1.pkl.gz"
    inp = f"testfile_caching{idx:02d}_{name}.tsv"
    nxt = tmp_path / "next.bfkc"
    res = _cli(["--input-file", inp, "--outdir", str(tmp_path / "out"), "--input-cache", str(cache), "--output-cache", str(nxt),
                "--max-dist", "1"])
    assert ("Import from side-car cache" in res.output) == (start == "sidecar")
    assert (tmp_path / "out" / "clusters.tsv").read_text() == cli_runs[f"cache_caching{idx:02d}"]["clusters_tsv"]
    exp = pd.read_table(f"expected_clusters_caching{idx:02d}_dist1.tsv", sep="\t")
    assert exp.equals(pd.read_table(tmp_path / "out" / "clusters.tsv", sep="\t"))
    _cli(["--input-file", inp, "--outdir", str(tmp_path / "again"), "--input-cache", str(nxt), "--max-dist", "1"])
    assert (tmp_path / "again" / "clusters.tsv").read_text() == cli_runs[f"cache_caching{idx:02d}"]["clusters_tsv"]


@pytest.mark.gpu
def test_sidecar_other_max_dist_and_pickle_output(tmp_path, monkeypatch, cli_runs):
    monkeypatch.chdir(FIX)
    cache = tmp_path / "init.bfkc"
    _cli(["--input-file", "testfile.tsv", "--outdir", str(tmp_path / "init"), "--output-cache", str(cache), "--max-dist", "1"])
    res = _cli(["--input-file", "testfile.tsv", "--outdir", str(tmp_path / "o"), "--input-cache", str(cache), "--max-dist", "2",
                "--min-cluster-size", "3"])
    assert "differnt max-dist" in res.output
    assert (tmp_path / "o" / "clusters.tsv").read_text() == cli_runs["dist2_mcs3"]["clusters_tsv"]
    import click.testing

    from breakfast_amd import console

    bad = click.testing.CliRunner().invoke(console.main, ["--input-file", "testfile.tsv", "--outdir", str(tmp_path / "p"),
                                                          "--input-cache", str(cache), "--output-cache", str(tmp_path / "x.pkl.gz")])
    assert bad.exit_code != 0 and "side-car" in str(bad.exception)


@pytest.mark.gpu
def test_sidecar_incremental_at_100k_rows_equals_a_fresh_run(tmp_path):
    """grow a 90k-profile input to 100k through the side-car (and drop 5k old rows on the way): same clusters.tsv as a fresh
    run of the final input, with lists computed only for the rows that are new"""
    from breakfast_amd.synth import generate_profiles

    rows = generate_profiles(100000)

    def write(path, idx):
        with open(path, "w") as f:
            f.write("accession\tdna_profile\n")
            for i in idx:
                f.write(f"seq{i:07d}\t{rows[i]}\n")

    write(tmp_path / "a.tsv", range(90000))
    keep = list(range(0, 40000)) + list(range(45000, 100000))
    write(tmp_path / "b.tsv", keep)
    cache = tmp_path / "c.bfkc"
    _cli(["--input-file", str(tmp_path / "a.tsv"), "--outdir", str(tmp_path / "oa"), "--output-cache", str(cache)])
    assert cache.stat().st_size < 8_000_000  # 16 B of hashes per row + the lists: not the 30 MB of feature strings
    _cli(["--input-file", str(tmp_path / "b.tsv"), "--outdir", str(tmp_path / "ob"), "--input-cache", str(cache)])
    _cli(["--input-file", str(tmp_path / "b.tsv"), "--outdir", str(tmp_path / "fresh")])
    got, want = (tmp_path / "ob" / "clusters.tsv").read_bytes(), (tmp_path / "fresh" / "clusters.tsv").read_bytes()
    # (a deleted row's cached list still chains its surviving neighbours — the reference's semantics, cache.py:51-71 — so the
    # cached run may join what a fresh run keeps apart; with this generator a deleted parent's children are within max-dist 2 of
    # each other, never within 1, and the partitions can differ there: compare as the reference would, component by component)
    if got != want:
        g = pd.read_table(tmp_path / "ob" / "clusters.tsv")
        w = pd.read_table(tmp_path / "fresh" / "clusters.tsv")
        assert g["id"].equals(w["id"])
        # every fresh cluster lies inside one cached cluster (the cache can only ADD connections)
        both = pd.DataFrame({"g": g["cluster_id"], "w": w["cluster_id"]}).dropna(subset=["w"])
        assert both.groupby("w")["g"].nunique(dropna=False).max() == 1


# ---- side-car runs on the device stages (fastpath._run_sidecar_on_device, bfk_table_cluster_write_device_cache) ----------------
def _synth_tsv(path, rows, ids=None):
    with open(path, "w") as f:
        f.write("accession\tdna_profile\n")
        for i, r in enumerate(rows):
            f.write(f"{ids[i] if ids else f'seq{i:07d}'}\t{r}\n")


@pytest.mark.gpu
@pytest.mark.parametrize("var_type,filt,d,sep2", [("covsonar_dna", True, 1, " "), ("covsonar_dna", True, 2, " "), ("covsonar_dna", False, 1, " "),
                                                  ("raw", True, 1, " "), ("covsonar_dna", True, 1, ", "), ("covsonar_dna", False, 1, "::"),
                                                  ("raw", True, 2, " | ")])
def test_device_written_sidecar_holds_the_host_stage_hashes_and_the_lists_of_all_rows(var_type, filt, d, sep2, tmp_path):
    """bfk_table_cluster_write_device_cache: the two hashes of every unique row's feature string come from the device
    (k_row_hashes: kept tokens re-joined, or the raw bytes when nothing is filtered) and equal bfk_table_feature_hashes of the
    host stage; the lists are bfk_neighbours_csr's for all rows; the file is marked exact; clusters.tsv is the plain run's"""
    from breakfast_amd import synth

    rows = synth.generate_profiles(20000, seed=31, p_del=0.05, p_ins=0.02)
    rows[7] += " notatoken"
    rows[8] = "  " + rows[8] + "  C241T"   # empty tokens, a token the trim drops
    rows[9] = ""
    rows += rows[100:400]
    if sep2 != " ":   # (a separator of several bytes: folded on the device, the hashes are those of the strings that hold it)
        rows = [sep2.join(r.split(" ")) for r in rows]
    inp = tmp_path / "in.tsv"
    _synth_tsv(inp, rows)
    opts = (sep2, var_type, filt, filt, 264 if filt else 0, 228 if filt else 0, 29903)
    t = _lib.Table.open(inp, "\t", "accession", "dna_profile")
    info, k = t.cluster_write_device(*opts, d, 2, tmp_path / "dev.tsv", cache_path=tmp_path / "c.bfkc")
    t.close()
    t = _lib.Table.open(inp, "\t", "accession", "dna_profile")
    info_h = t.prepare(*opts)
    assert (info.n_unique, info.nnz, info.n_invalid) == (info_h.n_unique, info_h.nnz, info_h.n_invalid)
    assert sidecar.is_exact(tmp_path / "c.bfkc")
    h, off, flat = sidecar.load(tmp_path / "c.bfkc", d)
    assert np.array_equal(h, t.feature_hashes())
    ptr, idx = _lib.neighbours_csr(t.indptr, t.indices, d, None)
    assert np.array_equal(off, ptr) and np.array_equal(flat, idx)
    t.close()
    t = _lib.Table.open(inp, "\t", "accession", "dna_profile")
    _, k2 = t.cluster_write_device(*opts, d, 2, tmp_path / "host.tsv")   # (the plain run: labels only, no edge recorded)
    t.close()
    assert k == k2 and (tmp_path / "dev.tsv").read_bytes() == (tmp_path / "host.tsv").read_bytes()


@pytest.mark.gpu
def test_sidecar_runs_take_the_device_stages_when_that_is_the_same_run(tmp_path, monkeypatch):
    """A growing input continued through exact side-cars runs as the no-cache run on the device stages (the list path is never
    entered), with the stdout and clusters.tsv of the list path (BFK_CACHE_REUSE=1); an input that LOST a cached row, a
    format-1 cache, and a cache of another max-dist go where the reference's semantics need them to."""
    from breakfast_amd import fastpath, synth

    rows = synth.generate_profiles(30000, seed=77, p_del=0.05, p_ins=0.02)
    _synth_tsv(tmp_path / "a.tsv", rows[:24000])
    _synth_tsv(tmp_path / "b.tsv", rows)                                        # grown: every cached row still there
    lost = list(range(0, 9000)) + list(range(9500, 30000))
    _synth_tsv(tmp_path / "c.tsv", [rows[i] for i in lost], [f"seq{i:07d}" for i in lost])   # 500 cached rows gone
    list_path = []
    real = fastpath._run_with_cache
    monkeypatch.setattr(fastpath, "_run_with_cache", lambda *a, **k: list_path.append(1) or real(*a, **k))

    def run(name, inp, *extra, reuse=False):
        monkeypatch.setenv("BFK_CACHE_REUSE", "1" if reuse else "0")
        list_path.clear()
        res = _cli(["--input-file", str(tmp_path / inp), "--outdir", str(tmp_path / name), *extra])
        # (the prints from the first result line on: the parameter echo in front names the run's own files)
        return res.output[res.output.index("Number of sequences"):], (tmp_path / name / "clusters.tsv").read_bytes(), bool(list_path)

    out, tsv, lp = run("a", "a.tsv", "--output-cache", str(tmp_path / "a.bfkc"))
    assert not lp and sidecar.is_exact(tmp_path / "a.bfkc")
    out_r, tsv_r, lp_r = run("a_r", "a.tsv", "--output-cache", str(tmp_path / "a_r.bfkc"), reuse=True)
    assert lp_r and (out, tsv) == (out_r, tsv_r) and sidecar.is_exact(tmp_path / "a_r.bfkc")
    for x, y in zip(sidecar.load(tmp_path / "a.bfkc", 1), sidecar.load(tmp_path / "a_r.bfkc", 1)):
        assert np.array_equal(x, y)
    # grown input: device stages, same prints and clusters as the list path and as a run without a cache
    out, tsv, lp = run("b", "b.tsv", "--input-cache", str(tmp_path / "a.bfkc"), "--output-cache", str(tmp_path / "b.bfkc"))
    out_r, tsv_r, lp_r = run("b_r", "b.tsv", "--input-cache", str(tmp_path / "a.bfkc"), "--output-cache", str(tmp_path / "b_r.bfkc"), reuse=True)
    fresh = run("b_f", "b.tsv")
    assert not lp and lp_r and (out, tsv) == (out_r, tsv_r) and tsv == fresh[1]
    assert "Import from side-car cache" in out and "not available" not in out
    assert sidecar.is_exact(tmp_path / "b.bfkc") and sidecar.is_exact(tmp_path / "b_r.bfkc")
    # the input lost cached rows: their lists still chain their neighbours (cache.py:51-71) — the list path, format 1 out
    out, tsv, lp = run("c", "c.tsv", "--input-cache", str(tmp_path / "b.bfkc"), "--output-cache", str(tmp_path / "c.bfkc"))
    out_r, tsv_r, _ = run("c_r", "c.tsv", "--input-cache", str(tmp_path / "b.bfkc"), reuse=True)
    assert lp and tsv == tsv_r and not sidecar.is_exact(tmp_path / "c.bfkc")
    # ... and a format-1 cache is never proof of anything: list path even though no row is missing
    out, tsv, lp = run("c2", "c.tsv", "--input-cache", str(tmp_path / "c.bfkc"))
    assert lp and tsv == tsv_r
    # another max-dist: announced, not used (cache.py:35-48) — the device stages
    out, tsv, lp = run("d2", "b.tsv", "--input-cache", str(tmp_path / "c.bfkc"), "--max-dist", "2")
    out_r, tsv_r, _ = run("d2_r", "b.tsv", "--input-cache", str(tmp_path / "c.bfkc"), "--max-dist", "2", reuse=True)
    assert not lp and (out, tsv) == (out_r, tsv_r) and "differnt max-dist" in out and "not available" in out
