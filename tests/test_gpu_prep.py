"""GPU parity of the DEVICE prepare — filter_features (breakfast.py:116-190) inside the tokeniser, collapse_duplicates (:72-79)
by row hash + exact comparison, sparse_feature_matrix (:193-215) of the unique rows — against the host stage
(bfk_table_prepare), which tests/test_frontend.py pins to the regex / pandas mirror of the reference and to the golden vectors.
Everything is compared field by field: group index of every input row, weights, CSR of the unique rows, vocabulary size, the
number of "Skipping invalid feature" lines; and clusters.tsv byte for byte through the one-call pipeline and the CLI."""

import hashlib
import io
import zlib
from contextlib import redirect_stdout

import numpy as np
import pytest
from test_frontend import OPTS, _fuzz_tokens

from breakfast_amd import _lib, fastpath, synth

pytestmark = pytest.mark.gpu

VAR_TYPES = ["covsonar_dna", "covsonar_aa", "nextclade_dna", "nextclade_aa", "raw"]


def both(ids, feats, sep2, var_type, opts):
    th = _lib.Table.from_lists(ids, feats)
    ih = th.prepare(sep2, var_type, *opts)
    td = _lib.Table.from_lists(ids, feats)
    try:
        idv = td.prepare_device(sep2, var_type, *opts)
    except _lib.Unsupported:
        return th, ih, None, None
    return th, ih, td, idv


def invalid_lines(t, info):
    """what the CLI prints for the table's invalid tokens, in order (a device prepare lists nothing when all of them are empty)"""
    listed = t.invalid_count()
    assert listed in (0, info.n_invalid)
    return [t.invalid(i) if listed else "" for i in range(info.n_invalid)]


def assert_same(th, ih, td, idv):
    assert td is not None, "the device stage declined the input"
    for k in ("n_rows", "n_unique", "nnz", "n_invalid", "n_vocab", "filtered"):
        assert getattr(ih, k) == getattr(idv, k), k
    assert invalid_lines(td, idv) == invalid_lines(th, ih)  # the reference's print order (breakfast.py:182-184)
    np.testing.assert_array_equal(td.group, th.group)
    np.testing.assert_array_equal(td.weight, th.weight)
    np.testing.assert_array_equal(td.indptr, th.indptr)
    np.testing.assert_array_equal(td.indices, th.indices)


def _structured_tokens(rng, n):
    """well-formed tokens of every grammar (and near misses): substitutions, insertions, deletions of the DNA and AA dialects"""
    genes = ["S", "N", "ORF1a", "orf1b", "E", "M", "ORF8", "x9", "7a"]
    L = "ACDEFGHIKLMNPQRSTVWYZ"
    out = []
    for _ in range(n):
        g, a, b = genes[int(rng.integers(len(genes)))], L[int(rng.integers(len(L)))], L[int(rng.integers(len(L)))]
        pos = int(rng.choice([0, 1, 99, 100, 101, 263, 264, 265, 771, 772, 899, 900, 999, 1000, 29674, 29675, int(rng.integers(0, 40000))]))
        k = int(rng.integers(1, 40))
        p = f"{pos:0{int(rng.integers(1, 8))}d}" if rng.random() < 0.2 else str(pos)
        out.append([f"{g}:{a}{p}{b}", f"{g}:{a}{p}{b}{a}", f"{g}:{a}{p}{b}{a}{b}", f"{g}:del:{p}:{k}", f"{g}:{a}{p}-", f"{g}:{a}{p}*",
                    f"{a}{p}{b}", f"{a}{p}{b}{a}", f"{a}{p}{b}{a}{b}{a}", f"del:{p}:{k}", f"{p}:{a}{b}", f"{p}:{a}", f"{p}-{pos + k}",
                    f"{p}", f"{g}:{a}{p}", f"{g}{a}{p}{b}", f"{g}:{p}{b}", f"{a}{p}", f"del:{p}", f"{p}-", f"{g}:{a}{p}b"][int(rng.integers(21))])
    return out


def host_invalid_tokens(tokens, var_type, opts):
    """which of `tokens` the host classifier calls invalid (one token per row, no separators inside)"""
    t = _lib.Table.from_lists([f"t{i}" for i in range(len(tokens))], tokens)
    info = t.prepare("\x01", var_type, *opts)  # (a separator no token holds: every row is one token)
    return {t.invalid(i) for i in range(info.n_invalid)}


@pytest.mark.parametrize("var_type", VAR_TYPES)
@pytest.mark.parametrize("opts", OPTS)
def test_device_filter_and_collapse_vs_host_on_the_token_grammar_fuzz(var_type, opts):
    """the five grammars as device byte matchers: rows built from the fuzz tokens of tests/test_frontend.py that the host calls
    valid (kept or dropped: trims, indel switches, 19+ digit positions, leading zeros ...) plus empty tokens in every place —
    identical group / weight / CSR / counts; with tokens the host calls invalid in the rows (and empty ones around them) the
    device stage lists the same tokens in the same order"""
    rng = np.random.default_rng(zlib.crc32(repr((var_type, opts, 1)).encode()))
    toks = sorted({t for t in _fuzz_tokens(rng, 1500) + _structured_tokens(rng, 1500) if t and " " not in t})
    filtering = opts[0] or opts[1] or opts[2] > 0 or opts[3] > 0
    bad = host_invalid_tokens(toks, var_type, opts) if filtering else set()
    good = [t for t in toks if t not in bad]
    feats = []
    for _ in range(400):
        k = int(rng.integers(0, 9))
        feats.append(" ".join(good[int(rng.integers(0, len(good)))] for _ in range(k)))
    g = lambda i: good[i % len(good)]
    feats += ["", " ", "  ", g(0) + "  " + g(1), " " + g(2), g(3) + " ", feats[5], feats[7], feats[5]]
    ids = [f"s{i}" for i in range(len(feats))]
    assert_same(*both(ids, feats, " ", var_type, opts))
    some = list(sorted(bad))[:: max(1, len(bad) // 12)]
    for tok in some:
        th, ih, td, idv = both(ids + ["x"], feats + [g(0) + " " + tok], " ", var_type, opts)
        assert ih.n_invalid >= 1, tok
        assert_same(th, ih, td, idv)
    if some:  # invalid tokens, empty tokens and kept ones interleaved in many rows: the order of the lines is the text's
        feats2 = list(feats)
        for k in range(0, len(feats2), 3):
            b1, b2 = some[k % len(some)], some[(k // 3 + 1) % len(some)]
            feats2[k] = [b1 + " " + feats2[k], feats2[k] + "  " + b1 + " ", " " + b1 + "  " + g(k) + " " + b2, b1 + " " + b1, "  " + b2 + "  "][k % 5]
        assert_same(*both(ids, feats2, " ", var_type, opts))


def test_device_filter_kats(kats):
    """the reference's own filter cases (tests/golden/kats.json, captured by importing it): the device stage either gives
    the host stage's result, the printed tokens included"""
    for c in kats["filter"]:
        if len(c["sep"]) != 1:
            continue
        opts = (c["skip_ins"], c["skip_del"], c["trim_start"], c["trim_end"], c["reference_length"])
        th, ih, td, idv = both(["a", "b"], [c["input"], c["input"]], c["sep"], c["var_type"], opts)
        names = [ln.split("'", 1)[1].rsplit("'", 1)[0] for ln in c["stdout"].splitlines() if ln.startswith("Skipping invalid")]
        assert_same(th, ih, td, idv)
        assert invalid_lines(td, idv) == names + names  # (two rows with the case's input)


@pytest.mark.parametrize("opts", OPTS[:4])
@pytest.mark.parametrize("indels", [False, True])
def test_device_prepare_synthetic_profiles_with_duplicates(opts, indels):
    kw = dict(p_del=0.05, p_ins=0.02) if indels else {}
    feats = synth.generate_profiles(20000, seed=11, **kw)
    feats += feats[:3000] + feats[100:200] * 30 + ["", ""]  # duplicates, a hub of 30 copies, empty profiles
    rng = np.random.default_rng(3)
    order = rng.permutation(len(feats))
    feats = [feats[i] for i in order]
    ids = [f"q{i}" for i in range(len(feats))]
    assert_same(*both(ids, feats, " ", "covsonar_dna", opts))


def test_device_prepare_identity_is_the_string_when_nothing_is_filtered():
    """no filter: the reference groups by the RAW string (:128-129) — "A  B" and "A B" stay apart, equal strings collapse"""
    feats = ["A1T  C2G", "A1T C2G", "A1T C2G", " A1T C2G", "A1T C2G ", "A1T  C2G", "", "", " ", "C2G A1T"]
    ids = [f"s{i}" for i in range(len(feats))]
    th, ih, td, idv = both(ids, feats, " ", "covsonar_dna", (False, False, 0, 0, 29903))
    assert_same(th, ih, td, idv)
    assert idv.n_unique == 7 and idv.filtered == 0
    # the same rows with the filter on: identity = the kept tokens in order
    th, ih, td, idv = both(ids, feats, " ", "covsonar_dna", (True, True, 0, 0, 29903))
    assert_same(th, ih, td, idv)
    assert idv.n_unique == 3


def test_device_prepare_a_hub_of_identical_rows_and_long_tokens():
    """60 000 copies of one profile (one slot of the row table for all of them) between other rows; insertions of 8+ bytes
    (hashed vocabulary entries) kept and dropped"""
    base = synth.generate_profiles(2000, seed=5, p_del=0.1, p_ins=0.1)
    feats = base[:1000] + [base[17]] * 60000 + base[1000:] + [base[3]] * 500
    ids = [f"h{i}" for i in range(len(feats))]
    for opts in ((True, True, 264, 228, 29903), (False, False, 264, 228, 29903), (False, True, 0, 0, 29903)):
        th, ih, td, idv = both(ids, feats, " ", "covsonar_dna", opts)
        assert_same(th, ih, td, idv)
        assert int(td.weight.max()) >= 60000


@pytest.mark.parametrize("max_dist", [2, 3])
def test_pipeline_on_several_devices_rehearsed_on_one(max_dist, tmp_path, monkeypatch):
    """bfk_table_cluster_write_device_gpus (`--gpus N`): filter + collapse + CSR on the device, then the unique rows' CSR through the
    multi-device driver — every rank's contexts on device 0 here (BFK_MULTI_ONE_DEVICE=1 also forces the split, which the rule would
    not take at this size): the same clusters.tsv as the one-device call"""
    inp = tmp_path / "in.tsv"
    synth.generate_tsv(inp, 30000, p_del=0.05, p_ins=0.01)
    opts = (False, False, 264, 228, 29903)
    t1 = _lib.Table.open(inp, "\t", "accession", "dna_profile")
    i1, n1 = t1.cluster_write_device(" ", "covsonar_dna", *opts, max_dist, 2, tmp_path / "one.tsv")
    monkeypatch.setenv("BFK_MULTI_ONE_DEVICE", "1")
    t4 = _lib.Table.open(inp, "\t", "accession", "dna_profile")
    i4, n4 = t4.cluster_write_device(" ", "covsonar_dna", *opts, max_dist, 2, tmp_path / "four.tsv", n_gpus=4)
    assert n1 == n4 and (tmp_path / "one.tsv").read_bytes() == (tmp_path / "four.tsv").read_bytes()
    for k in ("n_rows", "n_unique", "nnz", "n_invalid", "n_vocab", "filtered"):
        assert getattr(i1, k) == getattr(i4, k), k


def test_device_prepare_lists_invalid_tokens_and_declines_only_beyond_its_queue():
    """up to 65536 invalid token occurrences are noted on the device ({offset, length}) and ordered on the host; more than that
    (a file in another dialect altogether) is the host stage's"""
    rng = np.random.default_rng(5)
    feats = synth.generate_profiles(30000, seed=3)
    ids = [f"s{i}" for i in range(len(feats))]
    for k in rng.choice(len(feats), 100, replace=False):   # 100 stray tokens in 30k rows
        toks = feats[k].split(" ")
        toks.insert(int(rng.integers(0, len(toks) + 1)), ["S:N501Y", "stray", "n/a", "A12", "del:5"][int(rng.integers(5))])
        feats[k] = " ".join(toks)
    opts = (True, True, 264, 228, 29903)
    th, ih, td, idv = both(ids, feats, " ", "covsonar_dna", opts)
    assert ih.n_invalid == 100
    assert_same(th, ih, td, idv)
    many = ["x%d y%d" % (i, i) for i in range(40000)]    # 80 000 invalid tokens: beyond the queue
    th, ih, td, idv = both([f"m{i}" for i in range(len(many))], many, " ", "covsonar_dna", opts)
    assert td is None and ih.n_invalid == 80000


def test_device_prepare_declines_what_it_does_not_restate():
    t = _lib.Table.from_lists(["a", "b"], ["A1T||C2G", "A1T"])
    with pytest.raises(_lib.Unsupported):
        t.prepare_device("|" * 17, "raw", False, False, 0, 0, 0)  # (up to 16 bytes are folded: the tests below)
    with pytest.raises(_lib.Unsupported):
        t.prepare_device("|\n", "raw", False, False, 0, 0, 0)
    with pytest.raises(ValueError):
        t.prepare_device("", "raw", False, False, 0, 0, 0)
    t = _lib.Table.from_lists(["é", "b"], ["A1T C２G", "A1T"])  # non-ASCII under a grammar
    with pytest.raises(_lib.Unsupported):
        t.prepare_device(" ", "covsonar_dna", True, True, 0, 0, 29903)
    info = t.prepare_device(" ", "raw", True, True, 5, 5, 29903)
    assert info.n_unique == 2
    # accents in the ids (or in other columns) are nobody's business: only a FEATURE with non-ASCII bytes goes to the mirror
    ids, feats = ["é", "Köln-1", "b", "c"], ["A1T C2G bogus", "A1T C2G", "A1T  C2G", "A300T"]
    assert_same(*both(ids, feats, " ", "covsonar_dna", (True, True, 0, 0, 29903)))
    t = _lib.Table.from_lists(["a"], ["A1T C2G"])
    t.prepare_device(" ", "covsonar_dna", True, True, 0, 0, 29903)
    with pytest.raises(_lib.BfkError) as ei:  # the filtered strings stay with the host stage
        t.feature(0)
    assert ei.value.code == -6


@pytest.mark.parametrize("sep2", [", ", "::", " | ", "--", "ab", ";;;", "0123456789abcdef"])
@pytest.mark.parametrize("var_type,opts", [("covsonar_dna", OPTS[0]), ("covsonar_aa", OPTS[1]), ("nextclade_dna", OPTS[0]), ("raw", OPTS[2]),
                                           ("nextclade_aa", OPTS[0]), ("covsonar_dna", OPTS[-1])])
def test_device_prepare_takes_token_separators_of_several_bytes(sep2, var_type, opts):
    """--sep2 may be any string (breakfast.py:164, str.split): the device stage folds every occurrence — leftmost, never
    overlapping, as str.split finds them — into a byte the table does not hold and gives what the host stage gives: groups,
    CSR, counts, and the invalid tokens (empty ones among them) in the reference's order.  Separators that overlap themselves
    ('::' in ':::', ' | ' in ' | | '), tokens that hold pieces of the separator, empty tokens in front, between and behind."""
    rng = np.random.default_rng(zlib.crc32(repr((sep2, var_type, opts)).encode()))
    toks = sorted({t for t in _fuzz_tokens(rng, 800) + _structured_tokens(rng, 800) if t})
    piece = [sep2[:1], sep2[-1:], sep2[:-1], sep2[1:]]
    feats = []
    for _ in range(500):
        k = int(rng.integers(0, 9))
        row = [toks[int(rng.integers(0, len(toks)))] for _ in range(k)]
        for _ in range(int(rng.integers(0, 3))):  # empty tokens, and tokens made of (or ending in) pieces of the separator
            row.insert(int(rng.integers(0, len(row) + 1)), ["", "", piece[int(rng.integers(4))], toks[int(rng.integers(len(toks)))] + piece[0]][int(rng.integers(4))])
        feats.append(sep2.join(row))
    feats += ["", sep2, sep2 + sep2, sep2[:-1], sep2 + sep2[:1], toks[0] + sep2, sep2 + toks[1], feats[3], feats[4], feats[3]]
    ids = [f"s{i}" for i in range(len(feats))]
    th, ih, td, idv = both(ids, feats, sep2, var_type, opts)
    assert_same(th, ih, td, idv)
    assert ih.n_unique < len(feats)


def test_device_prepare_finds_a_stand_in_byte_the_table_does_not_hold():
    """the first candidates (0x1F, 0x1E, ...) occur in the features: the next free one is taken; with every candidate in the table
    the device stage declines and the host stage takes the input"""
    feats = ["A1T, C2G\x1f, G3A", "A1T, C2G\x1e", "A1T, C2G\x1f, G3A", ", \x1d"]
    th, ih, td, idv = both(["a", "b", "c", "d"], feats, ", ", "raw", (False, False, 0, 0, 0))
    assert_same(th, ih, td, idv)
    assert ih.n_unique == 3
    every = "".join(chr(b) for b in list(range(1, 9)) + [0x0B, 0x0C] + list(range(0x0E, 0x20)) + [0x7F])
    th, ih, td, idv = both(["a", "b"], ["A1T, " + every, "C2G"], ", ", "raw", (False, False, 0, 0, 0))
    assert td is None and ih.n_unique == 2


def test_cli_with_a_token_separator_of_several_bytes_runs_the_device_stages(tmp_path, monkeypatch):
    """--sep2 ', ' through fastpath.run: the device stages (the default) print and write what the host stages do"""
    rows = synth.generate_profiles(20000, seed=11, p_del=0.05, p_ins=0.01)
    lines = ["accession\tdna_profile"] + [f"q{i}\t" + ", ".join(r.split(" ")) for i, r in enumerate(rows)]
    lines[50] += ", , notatoken, "
    lines[9000] = "q8999\t, A300T,, C400G ,"
    lines += ["e1\t", "d1\t" + lines[5].split("\t")[1]]
    inp = tmp_path / "in.tsv"
    inp.write_text("\n".join(lines) + "\n")

    def run(outdir, device):
        monkeypatch.setenv("BFK_DEVICE_PREP", "1" if device else "0")
        buf = io.StringIO()
        with redirect_stdout(buf):
            ok = fastpath.run(inp, "\t", "accession", "dna_profile", "covsonar_dna", ", ", True, True, 264, 228, 29903, 1, 2, outdir)
        assert ok
        return buf.getvalue(), (outdir / "clusters.tsv").read_bytes()

    out_d, tsv_d = run(tmp_path / "dev", True)
    out_h, tsv_h = run(tmp_path / "host", False)
    assert out_d == out_h and tsv_d == tsv_h
    inv = [ln.split("'")[1] for ln in out_d.splitlines() if ln.startswith("Skipping invalid")]
    assert "notatoken" in inv and "A300T," in inv and "C400G ," in inv and inv.count("") >= 4
    td = _lib.Table.open(inp, "\t", "accession", "dna_profile")  # and the device stage did not decline
    td.prepare_device(", ", "covsonar_dna", True, True, 264, 228, 29903)


@pytest.mark.parametrize("max_dist,indels,opts", [(1, False, OPTS[0]), (3, True, OPTS[1]), (2, True, OPTS[0])])
def test_pipeline_on_the_device_writes_the_host_path_s_clusters_tsv(max_dist, indels, opts, tmp_path):
    """bfk_table_cluster_write_device (filter + collapse + CSR + clustering in HBM, writer) against host prepare +
    bfk_table_cluster_write on the same table: clusters.tsv byte for byte, the same counts"""
    kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
    inp = tmp_path / "in.tsv"
    synth.generate_tsv(inp, 30000, **kw)
    text = inp.read_text().splitlines()
    text += [ln.replace("seq", "dup") for ln in text[1:4000]] + ["empty1\t", "empty2\t"]
    inp.write_text("\n".join(text) + "\n")
    th = _lib.Table.open(inp, "\t", "accession", "dna_profile")
    ih = th.prepare(" ", "covsonar_dna", *opts)
    labels, _ = _lib.cluster_csr(th.indptr, th.indices, max_dist)
    cid, n_host = fastpath.cluster_ids(labels, th.weight, 2)
    th.write(tmp_path / "host.tsv", cid)
    td = _lib.Table.open(inp, "\t", "accession", "dna_profile")
    idv, n_dev = td.cluster_write_device(" ", "covsonar_dna", *opts, max_dist, 2, tmp_path / "dev.tsv")
    assert (tmp_path / "dev.tsv").read_bytes() == (tmp_path / "host.tsv").read_bytes()
    assert n_dev == n_host
    for k in ("n_rows", "n_unique", "nnz", "n_invalid", "n_vocab", "filtered"):
        assert getattr(ih, k) == getattr(idv, k), k


def test_cli_runs_the_device_stages_and_prints_what_the_host_stages_print(tmp_path, monkeypatch):
    """python -m breakfast_amd through fastpath.run: with the device prepare (the default) and with BFK_DEVICE_PREP=0 (host
    tokeniser / filter / collapse) — the same stdout, the same clusters.tsv; an input with a token that has to be listed
    is listed by the device stages too, in the reference's order"""
    inp = tmp_path / "in.tsv"
    synth.generate_tsv(inp, 20000)
    lines = inp.read_text().splitlines()
    lines += ["e1\t", "e2\t", "d1\t" + lines[5].split("\t")[1], "d2\t" + lines[5].split("\t")[1]]
    inp.write_text("\n".join(lines) + "\n")

    def run(outdir, device):
        monkeypatch.setenv("BFK_DEVICE_PREP", "1" if device else "0")
        buf = io.StringIO()
        with redirect_stdout(buf):
            ok = fastpath.run(inp, "\t", "accession", "dna_profile", "covsonar_dna", " ", True, True, 264, 228, 29903, 1, 2, outdir)
        assert ok
        return buf.getvalue(), hashlib.sha256((outdir / "clusters.tsv").read_bytes()).hexdigest()

    from breakfast_amd import _front

    _front.preload(inp)
    out_d, sha_d = run(tmp_path / "dev", True)
    out_h, sha_h = run(tmp_path / "host", False)
    assert out_d == out_h and sha_d == sha_h
    assert out_d.count("Skipping invalid feature: ''") == 2
    lines.insert(7000, "bad1\tA300T notatoken C400G")
    lines.insert(300, "bad2\t  S:N501Y A301T  xyz ")
    lines.append("bad3\tfoo")
    inp.write_text("\n".join(lines) + "\n")
    out_d, sha_d = run(tmp_path / "dev2", True)  # listed by the device stage: offsets noted by the kernels, ordered by the host
    out_h, sha_h = run(tmp_path / "host2", False)
    assert out_d == out_h and sha_d == sha_h and "Skipping invalid feature: 'notatoken'" in out_d
    inv = [ln for ln in out_d.splitlines() if ln.startswith("Skipping invalid")]
    assert [ln.split("'")[1] for ln in inv] == ["", "", "S:N501Y", "", "xyz", "", "notatoken", "", "", "foo"]


@pytest.mark.parametrize("quoting", ["all", "minimal"])
def test_cli_on_a_quoted_table_equals_the_pandas_mirror(quoting, tmp_path, monkeypatch):
    """pandas' quoting dialect in the native reader (read_table's defaults, breakfast.py:16-21): a table written with
    csv.QUOTE_ALL, and one whose fields hold quotes, tabs and line breaks where the dialect allows them — ids with doubled
    quotes, a feature with bytes behind its closing quote (rewritten in place: the device prepare reads the image's spans), a
    quoted feature that no pattern matches, a free-text column with line breaks — go down the device stages, the host stages
    and the pandas mirror to the same stdout and the same clusters.tsv"""
    import click.testing
    import csv

    from breakfast_amd import console

    rows = synth.generate_profiles(6000, seed=11, p_del=0.05, p_ins=0.02)
    inp = tmp_path / "in.tsv"
    with open(inp, "w", newline="") as f:
        w = csv.writer(f, delimiter="\t", quoting=csv.QUOTE_ALL if quoting == "all" else csv.QUOTE_MINIMAL, lineterminator="\n")
        w.writerow(["accession", "note", "dna_profile"])
        for i, r in enumerate(rows):
            acc = f's{i}' if i % 50 else f'hCoV "x"/{i}' if i % 100 else f"two\nlines {i}"
            note = "n" if i % 7 else 'said "so"\tthen\nleft'
            w.writerow([acc, note, r])
        w.writerow(["dupe", "n", rows[3]])
    if quoting == "minimal":
        with open(inp, "a") as f:
            f.write('tail1\tn\t"C300T"  bogus"\n')          # bytes behind the closing quote: feature is C300T  bogus"
            f.write('tail2\tn\t"C300T ""q"" A400G"\n')       # doubled quotes inside: the token "q" matches no pattern
            f.write('"tail3"\t"n"\t""\n')
            f.write('tail4\tn\t"C300T\nA400G C500T"\n')     # a line break inside the feature: part of a token
    args = ["--input-file", str(inp), "--max-dist", "1"]
    outs = {}
    for name, env in (("device", {}), ("host", {"BFK_DEVICE_PREP": "0"}), ("pandas", {"BFK_NO_FASTPATH": "1"})):
        for k in ("BFK_DEVICE_PREP", "BFK_NO_FASTPATH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        taken = []
        real = fastpath.run
        monkeypatch.setattr(fastpath, "run", lambda *a, **k: taken.append(real(*a, **k)) or taken[-1])
        res = click.testing.CliRunner().invoke(console.main, args + ["--outdir", str(tmp_path / name)])
        monkeypatch.setattr(fastpath, "run", real)
        assert res.exit_code == 0, (res.output, res.exception)
        assert taken == ([] if name == "pandas" else [True])
        outs[name] = (res.output.replace(str(tmp_path / name), ""), (tmp_path / name / "clusters.tsv").read_bytes())
    assert outs["device"] == outs["pandas"] and outs["host"] == outs["pandas"]
    if quoting == "minimal":
        assert "Skipping invalid feature: 'bogus\"'" in outs["device"][0] and "Skipping invalid feature: '\"q\"'" in outs["device"][0]
    assert b'"hCoV ""x""/50"\t' in outs["device"][1] and b'"two\nlines 0"\t' in outs["device"][1]


@pytest.mark.parametrize("seed", range(3))
def test_device_stages_on_random_quoted_tables_equal_the_host_stages(seed, tmp_path, monkeypatch):
    """random tables whose id / feature fields are plain, quoted, quoted with doubled quotes or with a tail behind the closing
    quote (rewritten in place by the reader), NA strings and empty features among them: the device stages read the features through
    the reader's {offset, length} spans of the one image — same stdout and clusters.tsv as the host stages"""
    rng = np.random.default_rng(700 + seed)
    toks = synth.generate_profiles(400, seed=seed, p_del=0.05, p_ins=0.02)

    def field(text, uniq=False):
        kind = int(rng.integers(0, 5))
        if kind == 0 or (uniq and kind == 4):
            return text
        if kind == 1:
            return '"' + text + '"'
        if kind == 2:   # a doubled quote inside: part of the content (for a feature: a token that matches no pattern)
            return '"' + text + (' ""x""' if not uniq else '""') + '"'
        if kind == 3:   # bytes behind the closing quote
            return '"' + text + '"' + ("z" if uniq else " C5T")
        return rng.choice(["NA", '"NA"', '""', "", "null"])
    for trial in range(8):
        n = int(rng.integers(50, 400))
        lines = ["accession\tnote\tdna_profile"]
        for r in range(n):
            lines.append("\t".join([field(f"s{trial}_{r}", True), field("n"), field(toks[int(rng.integers(0, len(toks)))])]))
        inp = tmp_path / f"in{trial}.tsv"
        inp.write_text(("\r\n" if trial % 3 == 0 else "\n").join(lines) + "\n")
        outs = []
        for dev in ("1", "0"):
            monkeypatch.setenv("BFK_DEVICE_PREP", dev)
            buf = io.StringIO()
            with redirect_stdout(buf):
                ok = fastpath.run(inp, "\t", "accession", "dna_profile", "covsonar_dna", " ", True, True, 264, 228, 29903, 1, 2,
                                  tmp_path / f"o{trial}_{dev}")
            assert ok
            outs.append((buf.getvalue(), (tmp_path / f"o{trial}_{dev}" / "clusters.tsv").read_bytes()))
        assert outs[0] == outs[1], trial
