"""CPU: the native front end / writer (bfk_table_*, SURVEY.md 8 f1 + f3) against the pandas-based mirror of the
reference's functions (breakfast_amd.breakfast, itself pinned to the reference by tests/test_shell.py and the
golden vectors).  No GPU: only --max-dist 0 reaches the clustering step here."""

import io
import re
import zlib
from contextlib import redirect_stdout

import click.testing

import numpy as np
import pandas as pd
import pytest
from conftest import GOLD, load_stage, stage_names

from breakfast_amd import _lib, breakfast, console, fastpath, synth

FIX = GOLD / "ref_fixtures"
OPTS = [  # (skip_ins, skip_del, trim_start, trim_end, reference_length)
    (True, True, 264, 228, 29903), (False, False, 0, 0, 29903), (True, False, 0, 0, 1000), (False, True, 100, 0, 1000),
    (False, False, 0, 100, 1000), (True, True, 0, 0, 0),
]


def mirror(ids, features, sep2, var_type, opts):
    """the pandas path: filter -> collapse -> CSR; -> (group, weight, indptr, indices, ufeatures, stdout)"""
    buf = io.StringIO()
    with redirect_stdout(buf):
        meta = pd.DataFrame({"id": ids, "feature": features})
        meta["feature"] = breakfast.filter_features(meta["feature"], sep2, var_type, *opts)
        nod = breakfast.collapse_duplicates(meta)
    ufeat = nod["feature"].tolist()
    where = {f: i for i, f in enumerate(ufeat)}
    group = np.array([where[f] for f in meta["feature"]], dtype=np.int32)
    weight = np.array([len(t) for t in nod["id"]], dtype=np.int32)
    indptr, indices, nv = _lib.build_csr(ufeat, sep2)
    return group, weight, indptr, indices, ufeat, buf.getvalue(), meta, nod


def check_against_mirror(ids, features, sep2, var_type, opts):
    t = _lib.Table.from_lists(ids, features)
    info = t.prepare(sep2, var_type, *opts)
    group, weight, indptr, indices, ufeat, out, meta, nod = mirror(ids, features, sep2, var_type, opts)
    assert info.n_rows == len(ids) and info.n_unique == len(ufeat)
    np.testing.assert_array_equal(t.group, group)
    np.testing.assert_array_equal(t.weight, weight)
    np.testing.assert_array_equal(t.indptr, indptr)
    np.testing.assert_array_equal(t.indices, indices)
    assert [t.feature(u) for u in range(info.n_unique)] == ufeat
    printed = [m for m in re.findall(r"Skipping invalid feature: '(.*)'", out)]
    assert [t.invalid(i) for i in range(info.n_invalid)] == printed
    return t, meta, nod


# ---- token grammar fuzz: the hand-written matchers vs the reference's regular expressions ----------------
def _fuzz_tokens(rng, n):
    pieces = ["A", "C", "T", "Z", "a", "z", "0", "1", "9", "27", "00300", "123456789012345678901", ":", "-", "*", "del",
              "del:", "ORF1a", "S", "N", "AB", ".", "_", "x:"]
    toks = []
    for _ in range(n):
        k = int(rng.integers(0, 6))
        toks.append("".join(pieces[int(rng.integers(0, len(pieces)))] for _ in range(k)))
    # well-formed examples of every class
    toks += ["C241T", "A23403G", "G5343TT", "del:11288:9", "del:1:", "del::1", "S:N501Y", "ORF1a:del:12:7", "N:A34AK",
             "S:V70-", "S:Q493*", "22204:GAGCCAGAA", "21765-21770", "28271", "1-", "-1", "S:", ":N501Y", "C0T", "C264T",
             "C265T", "C29674T", "C29675T", "C772T", "C771T", "A1000000000000000000000T", "A0000000000000000000300T"]
    return toks


@pytest.mark.parametrize("var_type", ["covsonar_dna", "covsonar_aa", "nextclade_dna", "nextclade_aa", "raw"])
@pytest.mark.parametrize("opts", OPTS)
def test_classifier_fuzz_vs_regex_mirror(var_type, opts):
    rng = np.random.default_rng(zlib.crc32(repr((var_type, opts)).encode()))  # (hash() of a str is salted per process)
    toks = _fuzz_tokens(rng, 1500)
    feats = []
    for _ in range(300):
        k = int(rng.integers(0, 8))
        feats.append(" ".join(toks[int(rng.integers(0, len(toks)))] for _ in range(k)))
    feats += ["", " ", "  ", "C241T  A23403G", " C241T", "C241T "]
    ids = [f"s{i}" for i in range(len(feats))]
    check_against_mirror(ids, feats, " ", var_type, opts)


@pytest.mark.parametrize("sep2", [" ", ",", "::", "ab"])
def test_token_separators(sep2):
    rng = np.random.default_rng(7)
    toks = ["C241T", "A23403G", "G5343TT", "del:11288:9", "x", ""]
    feats = [sep2.join(toks[int(rng.integers(0, len(toks)))] for _ in range(int(rng.integers(0, 6)))) for _ in range(200)]
    ids = [f"s{i}" for i in range(len(feats))]
    check_against_mirror(ids, feats, sep2, "covsonar_dna", (True, False, 0, 0, 29903))
    check_against_mirror(ids, feats, sep2, "raw", (False, False, 0, 0, 29903))


@pytest.mark.parametrize("name", stage_names())
def test_stage_vectors(name):
    """golden stage vectors made by the reference: filtered unique features and their CSR"""
    g = load_stage(name)
    ids = [f"s{i}" for i in range(len(g["ufeatures"]))]
    t = _lib.Table.from_lists(ids, g["ufeatures"])
    info = t.prepare(g["sep"], "raw", False, False, 0, 0, 0)
    assert info.n_unique == len(ids)
    np.testing.assert_array_equal(t.indptr, g["indptr"])
    np.testing.assert_array_equal(t.indices, g["indices"])


def test_synthetic_with_indels_and_duplicates():
    feats = synth.generate_profiles(3000, seed=11, p_del=0.05, p_ins=0.02)
    feats += feats[:50]  # exact duplicates
    ids = [f"q{i}" for i in range(len(feats))]
    for opts in OPTS[:3]:
        check_against_mirror(ids, feats, " ", "covsonar_dna", opts)


# ---- reader ---------------------------------------------------------------------------------------------
def _both_readers(path, sep="\t", id_col="id", feat="f"):
    with redirect_stdout(io.StringIO()):
        meta = breakfast.read_input(path, sep, id_col, feat)
    t = _lib.Table.open(path, sep, id_col, feat)
    info = t.prepare(" ", "raw", False, False, 0, 0, 0)
    assert [t.id(r) for r in range(len(t))] == meta["id"].tolist()
    got = [t.feature(int(u)) for u in t.group]
    assert got == meta["feature"].tolist()
    return t, meta, info


@pytest.mark.parametrize("text,sep", [
    ("id\tf\na\tX Y\nb\tX\n", "\t"),
    ("id\tf\na\tX Y\n\nb\t\n", "\t"),           # blank line, empty feature
    ("\n\nid\tf\na\tb\n", "\t"),                # leading blank lines
    ("id\tf\na\tb", "\t"),                      # no trailing newline
    ("x\tf\ty\tid\n1\tNA\t2\ta\n1\tnull\t2\tb\n3\tN/A\t4\tc\n5\tq\t6\td\n", "\t"),  # NA features, other columns
    ("f,id\nX Y,a\nX,b b\n", ","),              # other separator, swapped columns
    ("id\tf\n a \t X \n", "\t"),                # spaces are data
    ("id\tf\r\na\tX\r\n\r\nb\tY Z\r\n", "\t"),   # CRLF file (the reference's fixtures are)
    ("id\tf\r\na\tX\r", "\t"),                 # CR at end of file
    ("id\tf\nhCoV-19/Cote d\u2019Ivoire/\u00e9\u00e8/2021\tX Y\nM\u00fcnchen-7\tX\n\u6771\u4eac\U0001f9ec\tZ\n", "\t"),  # UTF-8 ids (2-, 3-, 4-byte forms)
    ("id\tf\na\tX\u00e9 Y\nb\tY X\u00e9\nc\tX\u00e9 Y\n", "\t"),   # UTF-8 in the feature column, nothing filtered: opaque bytes
    ("n\u00e4me\tid\tf\n\u00e9\ta\tX\n\u00e8\tb\tY\n", "\t"),     # UTF-8 in a column that is not used, and in its name
    ("\ufeffid\tf\na\tX\n", "\t"),            # byte-order mark (pandas drops it)
    ('\ufeff"id"\tf\n\u00e9\tX\n', "\t"),     # ... in front of a quoted name
    ("\ufeff\n  \nid\tf\na\tX\n", "\t"),      # ... in front of skipped lines
    ("id\tf\n   \nc\td\n \n", "\t"),         # lines of blanks are skipped like empty ones
    ("  \nid,f\na,X\n \t \nb,Y\n\t\n", ","),   # ... blanks and tabs, when the tab is not the separator
    ("id\tf\na\tX\n \t \nc\tZ\n", "\t"),     # ... but not when it is: a row of two blank fields
])
@pytest.mark.parametrize("chunk_bytes", [None, "1", "5"])
def test_reader_accepts_and_matches_pandas(text, sep, chunk_bytes, tmp_path, monkeypatch):
    """(chunk_bytes: the reader and the tokeniser cut their input into one slice per so many bytes — normally 1 MiB / 256 KiB;
    with 1 or 5 bytes even these tiny files are read in several slices, cut at line starts)"""
    if chunk_bytes:
        monkeypatch.setenv("BFK_CHUNK_BYTES", chunk_bytes)
        monkeypatch.setenv("BFK_THREADS", "6")
    p = tmp_path / "in.tsv"
    p.write_text(text)
    _both_readers(p, sep)


@pytest.mark.parametrize("text", [
    'id\tf\n"a\tX\n',                 # a quoted field that never ends (pandas: EOF inside string)
    'id\tf\na\t"X\nb\tY\n\nc\tZ',       # ... over several lines
    "id\tf\na\rb\tX\n",               # lone CR
    "id\tf\na\tX\tz\nb\tq\n",         # long row
    "id\tf\na\tX\nb\n",               # short row
    "id\tf\nNA\tX\n",                 # NA id
    "id\tf\na\tX\na\tY\n",            # duplicate ids
    "id\tg\na\tX\n",                  # missing column
    "id\tf\tf\na\tX\tY\n",            # duplicate column names
    "id\tf\n",                        # no rows
    "",                               # empty
    "id\tf\na\tX\n\t\nb\tY\n",          # a row of two empty fields: NA-valued id
])
@pytest.mark.parametrize("chunk_bytes", [None, "3"])
def test_reader_declines(text, chunk_bytes, tmp_path, monkeypatch):
    if chunk_bytes:
        monkeypatch.setenv("BFK_CHUNK_BYTES", chunk_bytes)
        monkeypatch.setenv("BFK_THREADS", "6")
    p = tmp_path / "in.tsv"
    p.write_text(text)
    with pytest.raises(_lib.Unsupported):
        _lib.Table.open(p, "\t", "id", "f")


@pytest.mark.parametrize("threads", ["1", "7"])
@pytest.mark.parametrize("dup_at", [None, (3, 39_990), (20_000, 20_001), (0, 16_384)])
def test_reader_checks_ids_in_parallel(dup_at, threads, tmp_path, monkeypatch):
    """the id check is row-parallel from 16k rows on (an insert-only table filled by compare-and-swap): a duplicate is found
    wherever its two rows lie — one thread's slice, two slices, the first and a late row — and distinct ids pass"""
    monkeypatch.setenv("BFK_THREADS", threads)
    n = 40_000
    ids = [f"seq{i:07d}" for i in range(n)]
    if dup_at:
        ids[dup_at[1]] = ids[dup_at[0]]
    p = tmp_path / "in.tsv"
    p.write_text("id\tf\n" + "".join(f"{x}\tA{i % 97}T C{i % 13}G\n" for i, x in enumerate(ids)))
    if dup_at:
        with pytest.raises(_lib.Unsupported, match="duplicate sequence identifiers"):
            _lib.Table.open(p, "\t", "id", "f")
    else:
        t = _lib.Table.open(p, "\t", "id", "f")
        assert len(t) == n
        t.close()


@pytest.mark.parametrize("bad,at", [("\x00", 5_000_000), ("\xe9", 123_457), ("\xc3", 999_999), ("\xa9", 1_000_000),
                                    ("\xed\xa0\x80", 2_000_003), ("\xc0\xaf", 3_000_001), ("\xf4\x90\x80\x80", 4_000_002),
                                    ("\r", 7_654_321)])
def test_reader_byte_check_finds_a_bad_byte_anywhere(bad, at, tmp_path, monkeypatch):
    """the byte check looks at eight bytes at a time in parallel slices of a file read in parallel slices: a NUL,
    a lone CR or bytes that are not valid UTF-8 (a Latin-1 byte, a lead byte without its tail — also right at a slice cut —, a
    stray continuation byte, a surrogate, an overlong form, a code point beyond U+10FFFF) are found at any offset
    (word-aligned or not), and the clean file is accepted"""
    monkeypatch.setenv("BFK_THREADS", "5")
    monkeypatch.setenv("BFK_CHUNK_BYTES", "1000000")
    line = "s{:07d}\tA1T C22G del:333:4\n"
    body = "".join(line.format(i) for i in range(400_000))
    text = ("id\tf\n" + body).encode()
    p = tmp_path / "in.tsv"
    p.write_bytes(text)
    t = _lib.Table.open(p, "\t", "id", "f")
    assert len(t) == 400_000
    t.close()
    at = min(at, len(text) - 10)
    p.write_bytes(text[:at] + bad.encode("latin-1") + text[at + len(bad.encode("latin-1")):])
    with pytest.raises(_lib.Unsupported):
        _lib.Table.open(p, "\t", "id", "f")


@pytest.mark.parametrize("threads", [1, 3, 5, 8])
def test_reader_finds_a_stray_continuation_byte_at_a_slice_start(threads, tmp_path, monkeypatch):
    """the UTF-8 check runs in parallel slices cut at n * q / threads: a continuation byte right at a cut that no lead byte in
    front of it covers (the slice in front ends in ASCII, or in a complete sequence) is invalid for every thread count — pandas
    raises UnicodeDecodeError on it — and a sequence that does straddle the cut is accepted"""
    monkeypatch.setenv("BFK_THREADS", str(threads))
    line = "s{:07d}\tA1T C22G del:333:4\n"
    text = bytearray(("id\tf\n" + "".join(line.format(i) for i in range(60_000))).encode())
    n = len(text)
    p = tmp_path / "in.tsv"
    for q in range(1, 8):
        for parts in (3, 5, 8):
            cut = n * q // parts
            if cut + 4 >= n or any(text[j] in b"\t\n" for j in range(cut - 2, cut + 3)):
                continue
            for bad in (b"\xa9", b"\xc3\xa9\xa9"[1:], b"\xa9\xa9"):   # stray continuation bytes at the cut, ASCII in front
                t2 = bytearray(text)
                t2[cut:cut + len(bad)] = bad
                p.write_bytes(bytes(t2))
                with pytest.raises(_lib.Unsupported):
                    _lib.Table.open(p, "\t", "id", "f")
            t2 = bytearray(text)   # a complete two-byte sequence in front of the cut, then a stray continuation byte AT the cut
            t2[cut - 2:cut + 1] = b"\xc3\xa9\xa9"
            p.write_bytes(bytes(t2))
            with pytest.raises(_lib.Unsupported):
                _lib.Table.open(p, "\t", "id", "f")
            t2 = bytearray(text)   # a three-byte sequence across the cut: valid
            t2[cut - 1:cut + 2] = "\u6771".encode()
            p.write_bytes(bytes(t2))
            t = _lib.Table.open(p, "\t", "id", "f")
            assert len(t) == 60_000
            t.close()


def test_reader_takes_utf8_at_any_offset_and_across_slice_cuts(tmp_path, monkeypatch):
    """valid multi-byte sequences wherever the parallel slices are cut (a slice that starts inside a sequence leaves its head
    to the slice in front): accepted, ids and clusters.tsv bytes equal to the pandas mirror's"""
    monkeypatch.setenv("BFK_THREADS", "7")
    names = ["s\u00e9q", "M\u00fcnchen", "\u6771\u4eac", "\U0001f9ec", "plain"]
    text = "id\tf\n" + "".join(f"{names[i % 5]}{i}\tA{i % 11}T C{i % 7}G\n" for i in range(3000))
    p = tmp_path / "in.tsv"
    p.write_text(text, encoding="utf-8")
    for cb in ("1", "7", "64", "1000"):
        monkeypatch.setenv("BFK_CHUNK_BYTES", cb)
        t, meta, info = _both_readers(p)
        cid = (np.arange(info.n_unique) % 3).astype(np.int32)
        t.write(tmp_path / "native.tsv", cid)
        t.close()
    raw = (tmp_path / "native.tsv").read_bytes()
    assert raw.decode("utf-8").split("\n")[1].startswith("s\u00e9q0\t")


def test_prepare_declines_non_ascii_features_under_a_grammar(tmp_path):
    """the reference's token patterns are str patterns (\\d matches the digits of every script): a feature with non-ASCII
    bytes is the pandas path's whenever a grammar is applied; ids may hold anything; raw / unfiltered runs take the bytes as
    they are"""
    t = _lib.Table.from_lists(["\u00e9a", "b"], ["A1T C\uff12G", "A1T"])
    with pytest.raises(_lib.Unsupported):
        t.prepare(" ", "covsonar_dna", True, True, 0, 0, 29903)
    info = t.prepare(" ", "covsonar_dna", False, False, 0, 0, 29903)   # nothing to filter: verbatim
    assert info.n_unique == 2 and t.feature(0) == "A1T C\uff12G"
    info = t.prepare(" ", "raw", True, True, 5, 5, 29903)
    assert info.n_unique == 2
    t.close()
    t = _lib.Table.from_lists(["\u00e9a", "b\u00fc"], ["A1T C2G", "A1T"])   # non-ASCII ids only: the grammar runs
    info = t.prepare(" ", "covsonar_dna", True, True, 0, 0, 29903)
    assert info.n_unique == 2 and t.id(0) == "\u00e9a" and t.id(1) == "b\u00fc"
    t.close()


def test_reader_declines_what_is_not_a_regular_file(tmp_path):
    with pytest.raises(_lib.Unsupported):
        _lib.Table.open(tmp_path, "\t", "id", "f")  # a directory


@pytest.mark.parametrize("fixture,idc,fc", [("testfile.tsv", "accession", "dna_profile"),
                                            ("testfile_nextclade.tsv", "seqName", "substitutions")])
def test_reader_on_reference_fixtures(fixture, idc, fc):
    path = FIX / fixture
    if not path.exists():
        pytest.skip("fixture not captured")
    hdr = path.read_text().split("\n")[0].split("\t")
    if idc not in hdr or fc not in hdr:
        pytest.skip("columns differ")
    _both_readers(path, "\t", idc, fc)


# ---- pandas' quoting dialect (read_table's defaults: '"', doubled quotes, QUOTE_MINIMAL) -------------------------------------
@pytest.mark.parametrize("text,sep", [
    ('id\tf\n"a"\tX Y\n"b"\t"X"\n', "\t"),                         # every field quoted
    ('"id"\t"f"\t"z"\n"a"\t"X Y"\t"1"\n"b"\t""\t"2"\n', "\t"),      # quoted names, a quoted empty feature (NaN -> "")
    ('id,f\n"Doe, J.",X\n"say ""hi""",Y\n', ","),                  # separator and doubled quotes inside
    ('id\tf\na"b\tX"Y\nc""d\tZ"\n', "\t"),                         # quotes that do not start a field are data
    ('id\tf\n"a"b\t"X"" Y"Z"\n', "\t"),                            # bytes behind the closing quote run on, verbatim
    ('id\tf\tnote\na\tX\t"line one\nline two"\nb\tY\t"t\tt"\n', "\t"),   # a line break / a separator inside another column
    ('id\tf\n"a\nb"\tX\n"c\r\nd"\tY Z\nq\tW\n', "\t"),               # ... inside the id (written back quoted)
    ('id\tf\na\t"X\nY"\nb\t"X\nY"\nc\tX\n', "\t"),                  # ... inside the feature
    ('id\tf\n"NA"x\tq\nb\t"NA"\nc\t"nu""ll"\n', "\t"),              # NA strings are NaN quoted or not
    ('"i""d"\tf\nb\tX\n', "\t"),                                   # (id column is literally i"d)
    ('id\tf\n""""\tX\n"\t"\tY\n', "\t"),                           # an id that is one quote, one that is one tab
    ('id\tf\r\n"a"\tX\r\n"b\r\nb"\t"Y"\r\n', "\t"),                 # CRLF file
    ('id\tf\n"a"\t"X"', "\t"),                                     # closing quote is the last byte
])
@pytest.mark.parametrize("chunk_bytes", [None, "1", "7"])
def test_reader_quoting_matches_pandas(text, sep, chunk_bytes, tmp_path, monkeypatch):
    if chunk_bytes:
        monkeypatch.setenv("BFK_CHUNK_BYTES", chunk_bytes)
        monkeypatch.setenv("BFK_THREADS", "6")
    p = tmp_path / "in.tsv"
    p.write_bytes(text.encode())
    idc = 'i"d' if text.startswith('"i""d"') else "id"
    t, meta, info = _both_readers(p, sep, idc)
    # and back out: clusters.tsv with these ids, against pandas' writer
    with redirect_stdout(io.StringIO()):
        nod = breakfast.collapse_duplicates(meta)
    cid = (np.arange(info.n_unique) % 2 + 1).astype(np.int32)
    nod["cluster_id"] = pd.array(cid, dtype="Int64").astype(object)
    breakfast.write_output(nod, meta, tmp_path / "a")
    t.write(tmp_path / "b.tsv", cid)
    t.close()
    assert (tmp_path / "b.tsv").read_bytes() == (tmp_path / "a" / "clusters.tsv").read_bytes()


def _pandas_or_none(path, sep):
    try:
        with redirect_stdout(io.StringIO()):
            return breakfast.read_input(path, sep, "id", "f")
    except Exception:
        return None


@pytest.mark.parametrize("seed", range(6))
def test_reader_quoting_fuzz_vs_pandas(seed, tmp_path, monkeypatch):
    """Random tables written with the csv module (QUOTE_ALL / QUOTE_MINIMAL / hand-quoted at random, fields that hold quotes,
    separators, line breaks, NA strings), and random byte soup from the same alphabet: whatever pandas reads, the native reader
    reads the same or declines; what pandas refuses (ragged rows, EOF inside a string, duplicates), it declines; well-formed
    tables are not declined.  Slices of a few bytes put the cuts inside quoted fields."""
    import csv
    rng = np.random.default_rng(100 + seed)
    pieces = ['"', '""', "\t", "\n", "\r\n", "a", "b", "NA", " ", "x y", ","]
    p = tmp_path / "in.tsv"
    accepted = 0
    for trial in range(60):
        sep = "\t" if trial % 3 else ","
        well_formed = trial % 2 == 0
        if well_formed:
            n = int(rng.integers(1, 12))
            def cell(k, unique=None):
                s = "".join(pieces[int(i)] for i in rng.integers(0, len(pieces), size=int(rng.integers(0, 4)))).replace("\r\n", "\n" if trial % 4 else "\r\n")
                return (f"{unique}{s}" if unique is not None else s)
            rows = [[cell(0, unique=f"r{r}"), cell(1), cell(2)] for r in range(n)]
            buf = io.StringIO()
            w = csv.writer(buf, delimiter=sep, quoting=[csv.QUOTE_ALL, csv.QUOTE_MINIMAL, csv.QUOTE_NONNUMERIC][trial % 3], lineterminator="\r\n" if trial % 5 == 0 else "\n")
            w.writerow(["id", "f", "z"])
            w.writerows(rows)
            text = buf.getvalue()
        else:
            body = "".join(pieces[int(i)] for i in rng.integers(0, len(pieces), size=int(rng.integers(1, 40))))
            text = f"id{sep}f\n" + body.replace(",", sep)
        p.write_bytes(text.encode())
        for cb in (None, "3"):
            if cb:
                monkeypatch.setenv("BFK_CHUNK_BYTES", cb)
                monkeypatch.setenv("BFK_THREADS", "5")
            else:
                monkeypatch.delenv("BFK_CHUNK_BYTES", raising=False)
            meta = _pandas_or_none(p, sep)
            try:
                t = _lib.Table.open(p, sep, "id", "f")
            except _lib.Unsupported:
                # (well-formed tables may still hold what is declined for other reasons: NA ids cannot occur — every id starts
                # with r<k> — but a lone CR can not either; so only pandas' own refusals remain)
                assert not (well_formed and meta is not None and not meta["id"].isna().any()), text
                continue
            assert meta is not None, f"pandas refuses what the native reader took: {text!r}"
            accepted += 1
            t.prepare(" ", "raw", False, False, 0, 0, 0)
            assert [t.id(r) for r in range(len(t))] == meta["id"].tolist(), text
            assert [t.feature(int(u)) for u in t.group] == meta["feature"].tolist(), text
            t.close()
    assert accepted >= 30


# ---- writer ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sep", ["\t", ","])
def test_writer_matches_pandas_writer(sep, tmp_path):
    rng = np.random.default_rng(3)
    n = 500
    feats = [" ".join(f"T{int(x)}" for x in rng.integers(0, 30, size=int(rng.integers(0, 4)))) for _ in range(n)]
    ids = [f"id {i}" if i % 7 else f"i\td{i}" for i in range(n)] if sep == "," else [f"id {i}" for i in range(n)]
    p = tmp_path / "in.txt"
    p.write_text(f"id{sep}f\n" + "".join(f"{i}{sep}{f}\n" for i, f in zip(ids, feats)))
    t, meta, info = _both_readers(p, sep)
    with redirect_stdout(io.StringIO()):
        nod = breakfast.collapse_duplicates(meta)
    assert len(nod) == info.n_unique
    cid = rng.integers(0, 40, size=info.n_unique).astype(np.int32)  # 0 = none, arbitrary numbering otherwise
    col = pd.array(cid, dtype="Int64")
    col[cid == 0] = pd.NA
    nod["cluster_id"] = col.astype(object)
    breakfast.write_output(nod, meta, tmp_path / "a")
    k = t.write(tmp_path / "b.tsv", cid)
    assert (tmp_path / "b.tsv").read_bytes() == (tmp_path / "a" / "clusters.tsv").read_bytes()
    assert k == len(set(cid[cid != 0].tolist()))


# ---- the whole CLI at --max-dist 0 (no GPU involved): fast path == pandas path == reference ----------------
@pytest.mark.parametrize("scenario", ["dist0", "nextclade_dist0"])
def test_cli_fastpath_dist0(scenario, cli_runs, tmp_path, monkeypatch):
    monkeypatch.chdir(FIX)
    run = cli_runs[scenario]
    taken = []
    real = fastpath.run
    monkeypatch.setattr(fastpath, "run", lambda *a, **k: taken.append(real(*a, **k)) or taken[-1])
    res_fast = click.testing.CliRunner().invoke(console.main, run["args"] + ["--outdir", str(tmp_path / "f")])
    assert res_fast.exit_code == 0, res_fast.output
    assert taken == [True]
    monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    res_slow = click.testing.CliRunner().invoke(console.main, run["args"] + ["--outdir", str(tmp_path / "s")])
    assert res_slow.exit_code == 0
    assert (tmp_path / "f" / "clusters.tsv").read_text() == run["clusters_tsv"]
    assert (tmp_path / "s" / "clusters.tsv").read_text() == run["clusters_tsv"]
    strip = lambda s: s.replace(str(tmp_path / "f"), "").replace(str(tmp_path / "s"), "")
    assert strip(res_fast.output) == strip(res_slow.output)  # same prints, same order


def test_cli_fastpath_synthetic_dist0_with_invalid_tokens(tmp_path, monkeypatch):
    rows = synth.generate_profiles(2000, seed=5, p_del=0.05, p_ins=0.02)
    rows[10] += " bogus!"
    rows[11] = ""
    rows += rows[:20]
    p = tmp_path / "in.tsv"
    p.write_text("accession\tdna_profile\tx\n" + "".join(f"s{i}\t{r}\t1\n" for i, r in enumerate(rows)))
    args = ["--input-file", str(p), "--max-dist", "0"]
    r1 = click.testing.CliRunner().invoke(console.main, args + ["--outdir", str(tmp_path / "f")])
    monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    r2 = click.testing.CliRunner().invoke(console.main, args + ["--outdir", str(tmp_path / "s")])
    assert r1.exit_code == 0 and r2.exit_code == 0, (r1.output, r2.output)
    assert (tmp_path / "f" / "clusters.tsv").read_bytes() == (tmp_path / "s" / "clusters.tsv").read_bytes()
    strip = lambda s: s.replace(str(tmp_path / "f"), "").replace(str(tmp_path / "s"), "")
    assert strip(r1.output) == strip(r2.output)
    assert "Skipping invalid feature: 'bogus!'" in r1.output


def test_cli_fastpath_on_a_quoted_table_at_dist0(tmp_path, monkeypatch):
    """a csv.QUOTE_ALL table with quotes, tabs and line breaks inside its fields through the native stages and through the pandas
    mirror: same stdout, same clusters.tsv (ids written back quoted where to_csv quotes them)"""
    import csv
    rows = synth.generate_profiles(1500, seed=9, p_del=0.05, p_ins=0.02)
    rows += rows[:30]
    p = tmp_path / "in.tsv"
    with open(p, "w", newline="") as f:
        w = csv.writer(f, delimiter="\t", quoting=csv.QUOTE_ALL, lineterminator="\n")
        w.writerow(["accession", "note", "dna_profile"])
        for i, r in enumerate(rows):
            w.writerow([f's{i}' if i % 50 else f'hCoV "x"/{i}' if i % 100 else f"two\nlines {i}", "n" if i % 7 else 'said "so"\tthen\nleft',
                        r if i != 77 else r + ' "odd"'])
    args = ["--input-file", str(p), "--max-dist", "0"]
    taken = []
    real = fastpath.run
    monkeypatch.setattr(fastpath, "run", lambda *a, **k: taken.append(real(*a, **k)) or taken[-1])
    r1 = click.testing.CliRunner().invoke(console.main, args + ["--outdir", str(tmp_path / "f")])
    assert taken == [True]
    monkeypatch.setenv("BFK_NO_FASTPATH", "1")
    r2 = click.testing.CliRunner().invoke(console.main, args + ["--outdir", str(tmp_path / "s")])
    assert r1.exit_code == 0 and r2.exit_code == 0, (r1.output, r2.output)
    assert (tmp_path / "f" / "clusters.tsv").read_bytes() == (tmp_path / "s" / "clusters.tsv").read_bytes()
    strip = lambda s: s.replace(str(tmp_path / "f"), "").replace(str(tmp_path / "s"), "")
    assert strip(r1.output) == strip(r2.output)
    assert "Skipping invalid feature: '\"odd\"'" in r1.output
    assert b'"hCoV ""x""/50"\t' in (tmp_path / "f" / "clusters.tsv").read_bytes()


def test_cli_fastpath_declines_dialects_and_ignores_the_cache_at_dist0(tmp_path):
    p = tmp_path / "in.tsv"
    p.write_text('accession\tdna_profile\na\rb\tC300T\n')   # (a lone CR: pandas takes it as a line end)
    ok = fastpath.run(p, "\t", "accession", "dna_profile", "covsonar_dna", " ", True, True, 264, 228, 29903, 0, 2,
                      tmp_path / "o")
    assert ok is False and not (tmp_path / "o").exists()
    # max-dist 0 never touches a cache (cluster() goes to cluster_identical_features, breakfast.py:82-89): the native path runs
    p.write_text("accession\tdna_profile\na\tC300T\n")
    assert fastpath.run(p, "\t", "accession", "dna_profile", "covsonar_dna", " ", True, True, 264, 228, 29903, 0, 2,
                        tmp_path / "o", None, tmp_path / "cache.pkl") is True
    assert (tmp_path / "o" / "clusters.tsv").exists() and not (tmp_path / "cache.pkl").exists()


@pytest.mark.parametrize("indels", [False, True])
def test_thread_count_never_changes_the_result(indels, monkeypatch):
    """the text stages are row-parallel (BFK_THREADS); first-appearance vocabulary ids, the order of the invalid tokens and
    the first-appearance unique rows must not depend on the number of chunks"""
    from breakfast_amd.synth import generate_profiles

    kw = dict(p_del=0.05, p_ins=0.02) if indels else {}
    rows = generate_profiles(6000, **kw)
    rows = rows + rows[:500] + ["bad:token " + rows[7], "", rows[3] + "  " + rows[5]]
    ids = [f"s{i}" for i in range(len(rows))]
    ref = None
    for threads in ("1", "2", "3", "8", "16"):
        monkeypatch.setenv("BFK_THREADS", threads)
        ip, ix, nv = _lib.build_csr(rows, " ")
        t = _lib.Table.from_lists(ids, rows)
        info = t.prepare(" ", "covsonar_dna", True, not indels, 264, 228, 29903)
        got = (ip.copy(), ix.copy(), nv, t.group.copy(), t.weight.copy(), t.indptr.copy(), t.indices.copy(),
               [t.invalid(i) for i in range(int(info.n_invalid))], int(info.n_vocab))
        t.close()
        if ref is None:
            ref = got
            continue
        for a, b in zip(ref, got):
            assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b


def test_reader_slices_give_the_rows_of_one_pass(tmp_path, monkeypatch):
    """the reader cuts files over 1 MiB into slices at line starts (one thread each): same rows, same order, CRLF line
    ends and blank lines included, as the single-threaded pass"""
    from breakfast_amd.synth import generate_profiles

    rows = generate_profiles(12000)
    inp = tmp_path / "in.tsv"
    with open(inp, "w", newline="") as f:
        f.write("x\taccession\tdna_profile\r\n")
        for i, r in enumerate(rows):
            f.write(f"{i % 7}\tseq{i:07d}\t{r}\r\n")
            if i % 1000 == 999:
                f.write("\r\n")
    assert inp.stat().st_size > 3 << 20
    outs = []
    for threads in ("1", "2", "7"):
        monkeypatch.setenv("BFK_THREADS", threads)
        t = _lib.Table.open(inp, "\t", "accession", "dna_profile")
        assert len(t) == len(rows)
        info = t.prepare(" ", "covsonar_dna", True, True, 264, 228, 29903)
        out = tmp_path / f"o{threads}.tsv"
        t.write(out, (np.arange(int(info.n_unique)) % 5).astype(np.int32))
        outs.append((out.read_bytes(), t.group.copy(), [t.id(0), t.id(len(rows) - 1)]))
        t.close()
    for o in outs[1:]:
        assert o[0] == outs[0][0] and np.array_equal(o[1], outs[0][1]) and o[2] == outs[0][2] == ["seq0000000", f"seq{len(rows) - 1:07d}"]
