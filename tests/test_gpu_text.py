"""GPU parity of the DEVICE tokeniser (bfk_text.hip behind bfk_build_csr_device / bfk_ctx_build_csr / bfk_cluster_text):
text -> first-appearance vocabulary -> CSR must equal the reference's sparse_feature_matrix (breakfast.py:193-215) entry
for entry — golden stage vectors and KATs produced by the imported reference, the oracle's restatement on seeded inputs,
and the host tokeniser (bfk_build_csr) at BASELINE's sizes.  Integer / byte work: bit-exact."""

import os

import numpy as np
import pytest
from conftest import load_stage, stage_names

from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles
from oracle import ref_port as orc

pytestmark = pytest.mark.gpu
ORACLE_THREADS = min(16, os.cpu_count() or 8)  # (the oracle's OpenMP threads: a GPU box gives a test run 16 cores)


def _same(got, want):
    assert np.array_equal(got[0], want[0]), "indptr differs"
    assert np.array_equal(got[1], want[1]), "indices differ"
    assert int(got[2]) == int(want[2]), "n_vocab differs"


@pytest.mark.parametrize("name", stage_names())
def test_device_csr_matches_reference_golden(name):
    g = load_stage(name)   # (a separator of several bytes is folded on the device: k_sepfold)
    _same(_lib.build_csr_device(g["ufeatures"], g["sep"]), (g["indptr"], g["indices"], g["n_vocab"]))
    # the raw (uncollapsed) feature column too: duplicates rows, other first appearances
    want = orc.sparse_feature_matrix(g["features"], g["sep"])
    _same(_lib.build_csr_device(g["features"], g["sep"]), want)


def test_device_csr_kats(kats):
    for c in kats["cluster"]:
        if "error" in c or len(c["sep"]) < 1:
            continue
        uf = list(dict.fromkeys(c["features"]))
        indptr, indices, _ = _lib.build_csr_device(uf, c["sep"])
        assert indptr.tolist() == c["indptr"] and indices.tolist() == c["indices"], c


def test_device_csr_edge_cases():
    """the inputs of tests/test_abi.py::test_build_csr_edge_cases, through the device path"""
    indptr, indices, nv = _lib.build_csr_device(["", "C241T"], " ")  # reference tests/test_filtering.py:94-99
    assert indptr.tolist() == [0, 0, 1] and indices.tolist() == [0] and nv == 1
    indptr, indices, nv = _lib.build_csr_device([], " ")
    assert indptr.tolist() == [0] and len(indices) == 0 and nv == 0
    indptr, indices, nv = _lib.build_csr_device(["", "", ""], " ")
    assert indptr.tolist() == [0, 0, 0, 0] and len(indices) == 0 and nv == 0
    indptr, indices, nv = _lib.build_csr_device(["a|b||||a", float("nan"), "||", "|a"], "|")
    assert indptr.tolist() == [0, 3, 3, 3, 4] and indices.tolist() == [0, 1, 0, 0] and nv == 2
    # separators of several bytes (str.split takes any string, breakfast.py:204): folded on the device, leftmost matches, never
    # across a row boundary; over 16 bytes, or with every stand-in byte in the text: the host tokeniser's
    for sep, rows in (("||", ["a||b", "|||a", "a|", "|b||", "", "||||", "b|||"]), (", ", ["A, B,C , ", ", ,", "A,  B", " ,A"]),
                      ("aba", ["xabababay", "aba", "abaaba", "ab", "a"]), ("0123456789abcdef", ["x0123456789abcdefy0123456789abcde"])):
        _same(_lib.build_csr_device(rows, sep), orc.sparse_feature_matrix(rows, sep))
    rng = np.random.default_rng(12)
    for sep in ("::", " | ", "--", ", "):
        alphabet = ["A1T", "C22G", ":", "-", "|", " ", ",", "del:5:1", "S:N501Y", "x" * 9]
        rows = [sep.join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 12)))) for _ in range(3000)]
        _same(_lib.build_csr_device(rows, sep), orc.sparse_feature_matrix(rows, sep))
        lab, _, nnz, nv = _lib.cluster_text(*_lib.pack_rows(rows), sep, 1, want_stats=False)
        want = orc.sparse_feature_matrix(rows, sep)
        assert nnz == len(want[1]) and np.array_equal(lab[:len(rows)], orc.cluster_csr(want[0], want[1], 1, n_threads=ORACLE_THREADS)["labels"])
    with pytest.raises(_lib.Unsupported):
        _lib.build_csr_device(["a||b"], "|" * 17)
    with pytest.raises(_lib.Unsupported):
        _lib.build_csr_device(["a||b" + "".join(chr(b) for b in list(range(1, 9)) + [0x0B, 0x0C] + list(range(0x0E, 0x20)) + [0x7F])], "||")
    with pytest.raises(ValueError):
        _lib.build_csr_device(["a b"], "")
    # rows that abut without a separator: the row boundary ends a token ("ab" + "cd" are two tokens, not "abcd")
    indptr, indices, nv = _lib.build_csr_device(["ab", "cd", "ab cd", "abcd"], " ")
    assert indptr.tolist() == [0, 1, 2, 4, 5] and indices.tolist() == [0, 1, 0, 1, 2] and nv == 3
    # one-byte rows: a token per byte
    rows = [chr(65 + (i * 7) % 26) for i in range(5000)]
    _same(_lib.build_csr_device(rows, " "), orc.sparse_feature_matrix(rows, " "))
    big = " ".join(f"T{i}" for i in range(200000))  # 200k distinct tokens in one row: the table grows
    indptr, indices, nv = _lib.build_csr_device([big, big], " ")
    assert nv == 200000 and np.array_equal(indices[:200000], np.arange(200000)) and indptr.tolist() == [0, 200000, 400000]
    assert np.array_equal(indices[200000:], np.arange(200000))


def test_device_csr_long_tokens_and_window_borders():
    """tokens across the 16-byte lane, 1 KiB window and 16 KiB block borders; tokens of 1 .. 5000 bytes; runs of separators"""
    rng = np.random.default_rng(11)
    rows = []
    for r in range(400):
        toks = []
        for _ in range(int(rng.integers(0, 40))):
            ln = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100, 1023, 1024, 1025, 5000], p=[.1] * 7 + [.03] * 10))
            pool = int(rng.integers(1, 60))  # few distinct tokens per length: repeats, first appearances matter
            toks.append((chr(97 + pool % 26) * ln)[: ln - 1] + chr(65 + pool // 26) if ln > 1 else chr(97 + pool % 26))
        seps = [" " * int(rng.integers(1, 4)) for _ in toks]
        rows.append(" " * int(rng.integers(0, 3)) + "".join(t + s for t, s in zip(toks, seps)))
    want = orc.sparse_feature_matrix(rows, " ")
    _same(_lib.build_csr_device(rows, " "), want)
    with pytest.raises(_lib.Unsupported):  # a token of 64 KiB: declined, nothing done
        _lib.build_csr_device(["x" * 70000 + " y"], " ")
    indptr, indices, nv = _lib.build_csr_device(["x" * 65534 + " y " + "x" * 65534], " ")
    assert indices.tolist() == [0, 1, 0] and nv == 2


@pytest.mark.parametrize("seed", range(6))
def test_device_csr_fuzz_vs_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    sep = [" ", ",", ";", "\t", "|", "-"][seed]
    alphabet = [f"{chr(65 + int(a))}{int(p)}{chr(65 + int(b))}" for a, p, b in
                zip(rng.integers(0, 26, 500), rng.integers(1, 30000, 500), rng.integers(0, 26, 500))]
    rows = []
    for _ in range(int(rng.integers(1, 3000))):
        k = int(rng.integers(0, 60))
        toks = [alphabet[int(i)] for i in rng.integers(0, len(alphabet), k)]
        if rng.random() < 0.2:
            toks += [""] * int(rng.integers(1, 3))  # empty tokens (double separators)
            rng.shuffle(toks)
        rows.append(sep.join(toks))
    _same(_lib.build_csr_device(rows, sep), orc.sparse_feature_matrix(rows, sep))


def test_device_csr_row_off_with_a_base_and_malformed_offsets():
    rows = ["A1C G5T", "", "G5T", "Q9R A1C"]
    buf, off = _lib.pack_rows(rows)
    want = orc.sparse_feature_matrix(rows, " ")
    shifted = b"junk " + buf + b" tail"
    _same(_lib.build_csr_bytes(shifted, off + 5, " ", device=True), want)
    bad = off.copy()
    bad[2] = bad[1] - 1
    with pytest.raises(_lib.BfkError) as ei:
        _lib.build_csr_bytes(buf, bad, " ", device=True)
    assert ei.value.code == -1


@pytest.mark.parametrize("n,indels", [(10000, False), (100000, False), (100000, True)])
def test_device_csr_equals_host_tokeniser_at_scale(n, indels):
    kw = dict(p_del=0.05, p_ins=0.01) if indels else {}
    rows = list(dict.fromkeys(generate_profiles(n, **kw)))
    buf, off = _lib.pack_rows(rows)
    host = _lib.build_csr_bytes(buf, off, " ")
    _same(_lib.build_csr_bytes(buf, off, " ", device=True), host)
    if n == 10000:
        _same(host, orc.sparse_feature_matrix(rows, " "))


def test_ctx_build_csr_binds_what_it_built_and_cluster_text_agrees():
    rows = list(dict.fromkeys(generate_profiles(20000)))
    buf, off = _lib.pack_rows(rows)
    ip, ix, nv = _lib.build_csr_bytes(buf, off, " ")
    ctx = _lib.Context(0)
    ctx.set_profiling(True)
    nnz, nv2 = ctx.build_csr(buf, off, " ")
    assert (nnz, nv2) == (len(ix), nv)
    d_ip, d_ix = ctx.download_csr()
    assert np.array_equal(d_ip, ip) and np.array_equal(d_ix, ix)
    ts = ctx.text_stats()
    assert ts["nnz"] == nnz and ts["n_vocab"] == nv and ts["host_fallback"] == 0 and ts["ms_total"] > 0
    d_out = ctx.alloc(4 * len(rows))
    for d in (1, 2):
        ctx.cluster(d, d_out)
        ctx.sync()
        lab = ctx.download_i32(d_out, len(rows)).copy()
        want, _ = _lib.cluster_csr(ip, ix, d)
        assert np.array_equal(lab, want)
        lab2, st, nnz3, nv3 = _lib.cluster_text(buf, off, " ", d)
        assert np.array_equal(lab2, want) and (nnz3, nv3) == (nnz, nv)
    # a second, smaller input on the same context: nothing of the first build may leak
    rows2 = rows[:777]
    buf2, off2 = _lib.pack_rows(rows2)
    ctx.build_csr(buf2, off2, " ")
    d_ip, d_ix = ctx.download_csr()
    ip2, ix2, _ = _lib.build_csr_bytes(buf2, off2, " ")
    assert np.array_equal(d_ip, ip2) and np.array_equal(d_ix, ix2)
    ctx.close()


def _device_text(rows):
    """the rows' bytes in a torch CUDA buffer of bfk_text_device_bytes (the tail deliberately poisoned: the library must pad
    it itself) + the offsets as a CUDA int64 tensor"""
    import torch

    buf, off = _lib.pack_rows(rows)
    need = _lib.text_device_bytes(len(buf))
    d_text = torch.full((need,), 0x41, dtype=torch.uint8, device="cuda")  # 'A': a token byte, not a separator
    if len(buf):
        d_text[: len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    d_off = torch.from_numpy(off).cuda()
    torch.cuda.synchronize()
    return buf, off, d_text, d_off


@pytest.mark.parametrize("case", ["synthetic", "indels", "kat_shapes", "empty_rows", "no_rows"])
def test_text_resident_in_hbm_gives_the_same_csr_and_labels(case):
    """bfk_ctx_build_csr_device / bfk_ctx_cluster_text_device — the step bench.py times: profile strings in HBM -> labels in
    HBM, no copy of the text — against the host-buffer entry and the oracle"""
    import torch

    rows = {"synthetic": list(dict.fromkeys(generate_profiles(20000))),
            "indels": list(dict.fromkeys(generate_profiles(5000, p_del=0.05, p_ins=0.01))),
            "kat_shapes": ["X Y", "X X Y", "", "Y X", "X  Y", "Q R S", "ab", "cd", "ab cd", "abcd", " ", "X"],
            "empty_rows": ["", "", ""], "no_rows": []}[case]
    buf, off, d_text, d_off = _device_text(rows)
    want = orc.sparse_feature_matrix(rows, " ") if rows else (np.zeros(1, np.int32), np.zeros(0, np.int32), 0)
    ctx = _lib.Context(0)
    nnz, nv = ctx.build_csr_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ")
    assert (nnz, nv) == (len(want[1]), want[2])
    ip, ix = ctx.download_csr()
    assert np.array_equal(ip, want[0]) and np.array_equal(ix, want[1])
    assert d_text[len(buf):].cpu().numpy().tolist() == [32] * (len(d_text) - len(buf))  # the library padded the caller's buffer
    if len(ix):
        d_lab = torch.empty(max(len(rows), 1), dtype=torch.int32, device="cuda")
        for d in (1, 3):
            want_labels = orc.cluster_csr(want[0], want[1], d, n_threads=ORACLE_THREADS)["labels"]
            for _ in range(2):  # twice: a second step on the same buffers finds nothing left over from the first
                ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", d, d_lab.data_ptr())
                st = ctx.sync()
                assert st["n_retry_slices"] == 0
                assert np.array_equal(d_lab.cpu().numpy()[: len(rows)], want_labels)
    ctx.close()


@pytest.mark.parametrize("case", ["synthetic", "indels", "kat_shapes", "long_rows", "big_vocabulary"])
def test_labels_only_steps_may_keep_table_slots_as_column_ids(case, monkeypatch):
    """bfk_ctx_set_token_ids(ctx, 1): a labels-only text step at max_dist 1 stops at the vocabulary table's slot numbers (no
    first-occurrence walk, no k_voc_count / k_voc_ids / k_tok_ids).  The labels are the oracle's; the CSR it leaves bound has
    the reference's indptr and an INJECTIVE renaming of the reference's column ids; the same context numbers by first
    appearance again as soon as it is asked for a CSR or for another max_dist (device-driven steps, steps that wait once, the
    band kernels, rows longer than the device-driven form assumes, a vocabulary that makes the table grow)."""
    import torch

    rng = np.random.default_rng(5)
    rows = {"synthetic": list(dict.fromkeys(generate_profiles(12000))),
            "indels": list(dict.fromkeys(generate_profiles(5000, p_del=0.05, p_ins=0.01))),
            "kat_shapes": ["X Y", "X X Y", "", "Y X", "X  Y", "Q R S", "ab", "cd", "ab cd", "abcd", " ", "X", "X Y Z", "Y Z"],
            "long_rows": [" ".join(f"T{int(x)}" for x in rng.integers(0, 400, size=int(rng.integers(120, 200)))) for _ in range(300)],
            "big_vocabulary": [" ".join(f"U{i}_{j}" for j in range(20)) for i in range(6000)] + ["U5_1 U5_2", "U5_1 U5_2 U5_3"]}[case]
    if case == "long_rows":
        rows += [r + " EXTRA" for r in rows[:50]]
    buf, off, d_text, d_off = _device_text(rows)
    want = orc.sparse_feature_matrix(rows, " ")
    want_labels = {d: orc.cluster_csr(want[0], want[1], d, n_threads=ORACLE_THREADS)["labels"] for d in (1, 2, 3)}
    for path in ("auto", "allpairs"):
        _slots_as_ids_on_one_context(rows, buf, off, d_text, d_off, want, want_labels, path)
    # the one-shot entry: host text -> labels, without the vocabulary count (slots) and with it (first appearance)
    monkeypatch.setenv("BFK_TOK_ANY_IDS", "1")
    got = _lib.cluster_text(buf, off, " ", 1, want_vocab=False)
    assert np.array_equal(got[0], want_labels[1]) and got[2] == len(want[1]) and got[3] == -1
    got = _lib.cluster_text(buf, off, " ", 1)
    assert np.array_equal(got[0], want_labels[1]) and got[3] == want[2]


def _slots_as_ids_on_one_context(rows, buf, off, d_text, d_off, want, want_labels, path):
    import torch

    ctx = _lib.Context(0)
    ctx.set_candidate_path(path)
    ctx.set_token_ids(True)
    d_lab = torch.empty(len(rows), dtype=torch.int32, device="cuda")
    for rep in range(3):  # (steps on one context: the row bits cleared by one step serve the next)
        ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())
        ctx.sync()
        assert np.array_equal(d_lab.cpu().numpy(), want_labels[1]), rep
    ip, ix = ctx.download_csr()
    assert np.array_equal(ip, want[0])
    fwd, back = {}, {}
    for a, b in zip(want[1].tolist(), ix.tolist()):   # the renaming is a bijection between the two vocabularies
        assert fwd.setdefault(a, b) == b and back.setdefault(b, a) == a
    assert ctx.text_stats()["n_vocab"] == -1
    # another max_dist, and any entry that hands out a CSR: first-appearance ids, as ever
    ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 2, d_lab.data_ptr())
    ctx.sync()
    assert np.array_equal(d_lab.cpu().numpy(), want_labels[2])
    ip, ix = ctx.download_csr()
    assert np.array_equal(ip, want[0]) and np.array_equal(ix, want[1])
    nnz, nv = ctx.build_csr_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ")
    assert (nnz, nv) == (len(want[1]), want[2])
    ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())   # and slots again
    ctx.sync()
    assert np.array_equal(d_lab.cpu().numpy(), want_labels[1])
    # a slot-numbered CSR that stays bound serves other max_dist too (the prefix groups find the largest id themselves)
    for d in (2, 3):
        ctx.cluster(d, d_lab.data_ptr())
        ctx.sync()
        assert np.array_equal(d_lab.cpu().numpy(), want_labels[d]), d
    ctx.close()


def test_builds_of_changing_size_on_one_context_leave_no_row_bits_behind():
    """the row-start bits of a build are cleared by the build itself when it is done with them, and the next build sets its
    own in k_tok_clear (no k_tok_rowbits launch) — as long as it is no longer than what was cleared: texts that shrink, grow,
    shift their row starts by a byte, and a failed build in between must all tokenise as a fresh context does"""
    import torch

    base = list(dict.fromkeys(generate_profiles(6000)))
    seqs = [base[:2000], base[:5000], ["X" + r for r in base[:1500]], base[100:101], base, [r[1:] for r in base[:5999]], [], base[:3]]
    ctx = _lib.Context(0)
    keep = []
    for k, rows in enumerate(seqs):
        buf, off, d_text, d_off = _device_text(rows)
        keep.append((d_text, d_off))
        nnz, nv = ctx.build_csr_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ")
        want = orc.sparse_feature_matrix(rows, " ")
        assert (nnz, nv) == (len(want[1]), want[2]), k
        d_ip, d_ix = ctx.download_csr()
        assert np.array_equal(d_ip, want[0]) and np.array_equal(d_ix, want[1]), k
        if k == 2:  # a build that fails (offsets decrease) between two good ones
            o = off.copy()
            o[5] = o[4] - 1
            d_bad = torch.from_numpy(o).cuda()
            with pytest.raises(_lib.BfkError):
                ctx.build_csr_device(d_text.data_ptr(), len(buf), d_bad.data_ptr(), len(rows), " ")
    ctx.close()


def test_text_resident_in_hbm_checks_the_offsets_on_the_device():
    import torch

    rows = ["A1C G5T", "", "G5T", "Q9R A1C"]
    buf, off, d_text, d_off = _device_text(rows)
    ctx = _lib.Context(0)
    for bad in ("not_from_zero", "short_of_the_text", "decreasing"):
        o = off.copy()
        if bad == "not_from_zero":
            o[0] = 1
        elif bad == "short_of_the_text":
            o[-1] -= 2
        else:
            o[2] = o[1] - 1
        d_bad = torch.from_numpy(o).cuda()
        with pytest.raises(_lib.BfkError) as ei:
            ctx.build_csr_device(d_text.data_ptr(), len(buf), d_bad.data_ptr(), len(rows), " ")
        assert ei.value.code == -1, bad
    with pytest.raises(_lib.BfkError):
        ctx.build_csr_device(0, len(buf), d_off.data_ptr(), len(rows), " ")
    with pytest.raises(_lib.Unsupported):
        ctx.build_csr_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), "||")
    nnz, nv = ctx.build_csr_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ")  # and the context still works
    assert (nnz, nv) == (5, 3)
    ctx.close()


@pytest.mark.parametrize("case", ["long_row", "all_empty", "table_grows", "bad_offsets", "shuffled_multisets"])
def test_device_driven_text_step_redoes_what_its_assumptions_do_not_cover(case, monkeypatch):
    """bfk_ctx_cluster_text_device at max-dist 1 enqueues the clustering kernels behind the tokeniser on device-resident
    counts, assuming no row over 128 tokens, at least one token and a vocabulary table that is big enough; bfk_ctx_sync
    completes the bind and redoes the step where that did not hold — labels always the oracle's, errors reported by the sync.
    With BFK_SPEC=0 (the host sizes the clustering kernels after a wait) the result is the same."""
    import torch

    rng = np.random.default_rng(7)
    if case == "long_row":
        base = [f"A{i}T" for i in range(300)]
        rows = [" ".join(base[:200]), " ".join(base[:199]), " ".join(base[1:200]), "A1T A2T", "A1T", " ".join(base[:150])]
    elif case == "all_empty":
        rows = ["", "", " ", ""]
    elif case == "table_grows":
        rows = [" ".join(f"T{i}" for i in range(r * 50000, (r + 1) * 50000)) for r in range(4)] + ["T1 T2", "T1"]
    elif case == "shuffled_multisets":
        toks = [f"C{i}G" for i in range(40)]
        rows = []
        for _ in range(300):
            k = int(rng.integers(1, 9))
            r = [toks[int(x)] for x in rng.integers(0, 40, size=k)]
            rows.append(" ".join(r))
            r2 = list(r)
            rng.shuffle(r2)
            rows.append(" ".join(r2[: max(1, k - 1)]))
    else:
        rows = ["A1C G5T", "", "G5T", "Q9R A1C"]
    buf, off, d_text, d_off = _device_text(rows)
    if case == "bad_offsets":
        o = off.copy()
        o[2] = o[1] - 1
        d_off = torch.from_numpy(o).cuda()
    results = []
    for spec in ("1", "0"):
        monkeypatch.setenv("BFK_SPEC", spec)
        ctx = _lib.Context(0)
        d_lab = torch.full((max(len(rows), 1),), -7, dtype=torch.int32, device="cuda")
        for rep in range(2):
            if case == "bad_offsets":
                with pytest.raises(_lib.BfkError) as ei:
                    ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())
                    ctx.sync()
                assert ei.value.code == -1
                continue
            ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())
            st = ctx.sync()
            results.append((d_lab.cpu().numpy()[: len(rows)].copy(), st["n_edges"]))
        if case != "bad_offsets":
            ip, ix = ctx.download_csr()
            want = orc.sparse_feature_matrix(rows, " ")
            assert np.array_equal(ip, want[0]) and np.array_equal(ix, want[1])
        ctx.close()
    if case == "bad_offsets":
        return
    want = orc.sparse_feature_matrix(rows, " ")
    if len(want[1]):
        lab = orc.cluster_csr(want[0], want[1], 1, n_threads=ORACLE_THREADS)["labels"]
    else:  # (the reference's csr_matrix cannot even be built from an all-empty input: every row is the same empty multiset)
        lab = np.zeros(len(rows), dtype=np.int32)
    for got, _ in results:
        assert np.array_equal(got, lab), case
    assert len({e for _, e in results}) == 1


def test_device_driven_step_then_other_entries_without_a_sync():
    """an entry that follows bfk_ctx_cluster_text_device without a bfk_ctx_sync in between completes the open bind first"""
    import torch

    rows = list(dict.fromkeys(generate_profiles(5000)))
    buf, off, d_text, d_off = _device_text(rows)
    want = orc.sparse_feature_matrix(rows, " ")
    ctx = _lib.Context(0)
    d_lab = torch.empty(len(rows), dtype=torch.int32, device="cuda")
    ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())
    ip, ix = ctx.download_csr(len(rows), len(want[1]))   # no sync before: the entry itself completes the open bind
    assert np.array_equal(ip, want[0]) and np.array_equal(ix, want[1])
    ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())
    ctx.cluster(2, d_lab.data_ptr())                 # another step on the CSR the open bind leaves
    ctx.sync()
    assert np.array_equal(d_lab.cpu().numpy(), orc.cluster_csr(want[0], want[1], 2, n_threads=ORACLE_THREADS)["labels"])
    ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())
    ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())  # back to back
    assert ctx.text_stats()["nnz"] == len(want[1])
    ctx.sync()
    assert np.array_equal(d_lab.cpu().numpy(), orc.cluster_csr(want[0], want[1], 1, n_threads=ORACLE_THREADS)["labels"])
    ctx.close()                                       # and a context destroyed with a bind still open
    ctx = _lib.Context(0)
    ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, d_lab.data_ptr())
    ctx.close()


def test_open_text_steps_complete_in_order_and_a_step_outside_the_assumptions_redoes_the_ones_behind_it():
    """up to four device-driven text steps may be open at once (a caller that streams batches); a step whose input breaks the
    launch's assumptions is redone at the sync — and so is every step behind it, in order: each batch's labels are its own"""
    import torch

    ok_a = list(dict.fromkeys(generate_profiles(3000)))
    ok_b = list(dict.fromkeys(generate_profiles(2000, seed=5)))
    base = [f"A{i}T" for i in range(260)]
    long_rows = [" ".join(base[:200]), " ".join(base[:199]), "A1T A2T", "A1T"] + ok_b[:50]      # a row of 200 tokens
    batches = [ok_a, ok_b, long_rows, ok_b, ok_a, ok_b, long_rows, ok_a]                          # more than the ring holds
    dev = [_device_text(r) for r in batches]
    labs = [torch.full((len(r),), -3, dtype=torch.int32, device="cuda") for r in batches]
    ctx = _lib.Context(0)
    for (buf, off, d_text, d_off), rows, lab in zip(dev, batches, labs):
        ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, lab.data_ptr())
    ctx.sync()
    for rows, lab in zip(batches, labs):
        ip, ix, _ = orc.sparse_feature_matrix(rows, " ")
        assert np.array_equal(lab.cpu().numpy(), orc.cluster_csr(ip, ix, 1, n_threads=ORACLE_THREADS)["labels"])
    # the same buffer for every step (what bench.py does): the last step's labels are in it
    lab = torch.empty(len(ok_a), dtype=torch.int32, device="cuda")
    buf, off, d_text, d_off = dev[0]
    for _ in range(11):
        ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(ok_a), " ", 1, lab.data_ptr())
    st = ctx.sync()
    ip, ix, _ = orc.sparse_feature_matrix(ok_a, " ")
    assert st["n_retry_slices"] == 0 and np.array_equal(lab.cpu().numpy(), orc.cluster_csr(ip, ix, 1, n_threads=ORACLE_THREADS)["labels"])
    ctx.close()


def test_a_join_give_up_in_an_open_text_step_that_is_not_the_last_is_repaired():
    """the variant join's outcome is per step: a step whose join gives up (300 identical rows: a probe chain beyond
    JOIN_MAX_PROBE) while later device-driven steps are open is redone on the all-pairs path when it is completed — the device
    words alone describe only the last step (k_flatten leaves every step's outcome in the step's own pinned slot)"""
    import torch

    ok_a = list(dict.fromkeys(generate_profiles(3000)))
    ok_b = list(dict.fromkeys(generate_profiles(2000, seed=5)))
    same = ["A1T A2T A3T"] * 300 + ["A1T A2T", "A1T A2T A3T A4T", "C5G"] + ok_b[:50]
    for batches in ([ok_a, same, ok_b], [same, ok_a, ok_b, ok_a], [ok_a, same, same, ok_b, same]):
        dev = [_device_text(r) for r in batches]
        labs = [torch.full((len(r),), -3, dtype=torch.int32, device="cuda") for r in batches]
        ctx = _lib.Context(0)
        for (buf, off, d_text, d_off), rows, lab in zip(dev, batches, labs):
            ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, lab.data_ptr())
        ctx.sync()
        for rows, lab in zip(batches, labs):
            ip, ix, _ = orc.sparse_feature_matrix(rows, " ")
            assert np.array_equal(lab.cpu().numpy(), orc.cluster_csr(ip, ix, 1, n_threads=ORACLE_THREADS)["labels"])
        ctx.close()


@pytest.mark.parametrize("d", [2, 3])
def test_a_step_that_waits_once_is_repaired_before_the_next_one_on_its_context_rebinds(d):
    """max-dist >= 2 text steps wait once between their halves and leave the clustering kernels' outcome unread; a step whose
    candidate queue overflows (a run of 330 identical rows among 3 000: ~54 000 pairs at distance 0 against a queue sized for
    the forest) is redone by bfk_ctx_sync — but the NEXT step on the context re-binds the CSR that redo needs.  Every entry that
    binds, and every run into other labels or at another max-dist, settles the pending run first (ctx_settle; found by
    tools/soak_text.py): three steps without a sync in between, the overflowing one first, in the middle, last."""
    import torch

    base = list(dict.fromkeys(generate_profiles(9000)))
    dense = base[:1500] + [base[1500]] * 330 + base[1501:3000]
    batches = {"dense": dense, "a": base[3000:6000], "b": base[6000:9000]}
    want = {}
    for k, rows in batches.items():
        ip, ix, _ = orc.sparse_feature_matrix(rows, " ")
        want[k] = orc.cluster_csr(ip, ix, d, n_threads=ORACLE_THREADS)["labels"]
    dev = {k: _device_text(rows) for k, rows in batches.items()}
    for order in (("dense", "a", "b"), ("a", "dense", "b"), ("a", "b", "dense")):
        ctx = _lib.Context(0)
        labs = {}
        for k in order:
            buf, off, d_text, d_off = dev[k]
            labs[k] = torch.full((len(batches[k]),), -3, dtype=torch.int32, device="cuda")
            ctx.cluster_text_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(batches[k]), " ", d, labs[k].data_ptr())
        ctx.sync()
        for k in order:
            assert np.array_equal(labs[k].cpu().numpy(), want[k]), (order, k)
        # the bound-CSR entries: a run into other labels, or at another max-dist, behind an unread one
        buf, off, d_text, d_off = dev["dense"]
        ctx.build_csr_device(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(dense), " ")
        l1 = torch.full((len(dense),), -3, dtype=torch.int32, device="cuda")
        l2 = torch.full((len(dense),), -3, dtype=torch.int32, device="cuda")
        ctx.cluster(d, l1.data_ptr())
        ctx.cluster(d, l2.data_ptr())
        ctx.cluster(d + 1, l2.data_ptr())
        ctx.sync()
        assert np.array_equal(l1.cpu().numpy(), want["dense"])
        ctx.close()


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_text_pipeline_overlaps_the_steps_of_different_batches_and_keeps_every_batch_s_labels(depth):
    """distributed.TextPipeline: `depth` contexts of one GPU (a stream and buffers each) take the text steps in turn, so steps of
    different batches run beside each other; every batch's labels are its own (oracle), whatever the depth — also for batches
    outside the device-driven launch's assumptions (a row of 200 tokens; 300 identical rows: the join gives up) in the middle"""
    import torch

    from breakfast_amd.distributed import TextPipeline

    ok_a = list(dict.fromkeys(generate_profiles(3000)))
    ok_b = list(dict.fromkeys(generate_profiles(2000, seed=5)))
    base = [f"A{i}T" for i in range(260)]
    long_rows = [" ".join(base[:200]), " ".join(base[:199]), "A1T A2T", "A1T"] + ok_b[:50]
    same = ["A1T A2T A3T"] * 300 + ["A1T A2T", "C5G"] + ok_a[:40]
    batches = [ok_a, ok_b, long_rows, ok_b, same, ok_a, ok_b, ok_a, same, ok_b, ok_a]
    dev = [_device_text(r) for r in batches]
    labs = [torch.full((len(r),), -3, dtype=torch.int32, device="cuda") for r in batches]
    pipe = TextPipeline(0, depth)
    evs = []
    for (buf, off, d_text, d_off), rows, lab in zip(dev, batches, labs):
        evs.append(pipe.step_text(d_text.data_ptr(), len(buf), d_off.data_ptr(), len(rows), " ", 1, lab))
    pipe.sync()
    assert all(e.query() for e in evs)
    want = {}
    for rows, lab in zip(batches, labs):
        key = id(rows)
        if key not in want:
            ip, ix, _ = orc.sparse_feature_matrix(rows, " ")
            want[key] = orc.cluster_csr(ip, ix, 1, n_threads=ORACLE_THREADS)["labels"]
        assert np.array_equal(lab.cpu().numpy(), want[key])
    pipe.close()


def test_pinned_host_buffer_for_the_text():
    """bfk_host_alloc / bfk_host_free: a caller builds its text in page-locked memory and hands that to bfk_cluster_text"""
    rows = list(dict.fromkeys(generate_profiles(20000)))
    buf, off = _lib.pack_rows(rows)
    want, _, nnz, nv = _lib.cluster_text(buf, off, " ", 1)
    pin = _lib.PinnedBuffer(len(buf))
    pin.view[:] = np.frombuffer(buf, dtype=np.uint8)
    out = np.empty(len(rows), dtype=np.int32)
    got, _, nnz2, nv2 = _lib.cluster_text(pin, off, " ", 1, labels_out=out)
    assert got is out[: len(rows)] or np.shares_memory(got, out)
    assert np.array_equal(got, want) and (nnz2, nv2) == (nnz, nv)
    pin.free()
    pin.free()  # idempotent
    z = _lib.PinnedBuffer(0)
    z.free()


def test_cluster_text_falls_back_to_the_host_tokeniser_for_inputs_the_device_declines():
    rows = ["A||B", "A||B||C", "Q", "A||B"]
    buf, off = _lib.pack_rows(rows)
    lab, st, nnz, nv = _lib.cluster_text(buf, off, "||", 1)
    assert lab.tolist() == [0, 0, 2, 0] and nnz == 8 and nv == 4


def test_cluster_text_matches_oracle_labels():
    rows = list(dict.fromkeys(generate_profiles(3000, p_del=0.05, p_ins=0.01)))
    buf, off = _lib.pack_rows(rows)
    ip, ix, _ = orc.sparse_feature_matrix(rows, " ")
    for d in (1, 3):
        lab, st, _, _ = _lib.cluster_text(buf, off, " ", d)
        assert np.array_equal(lab, orc.cluster_csr(ip, ix, d, n_threads=ORACLE_THREADS)["labels"])


def test_device_csr_with_the_text_sent_in_pieces(monkeypatch):
    """BFK_TOK_PIECES (opt-in: the text goes up in pieces on a copy stream, piece k is tokenised under the copy of piece
    k + 1; the last units of a piece wait for the next piece — a token may reach into it): the same CSR"""
    rows = list(dict.fromkeys(generate_profiles(100000)))
    # long tokens across the piece borders too
    rows[len(rows) // 2] += " " + "L" * 60000 + " " + "M" * 5000
    buf, off = _lib.pack_rows(rows)
    host = _lib.build_csr_bytes(buf, off, " ")
    for pieces in ("2", "5", "8"):
        monkeypatch.setenv("BFK_TOK_PIECES", pieces)
        _same(_lib.build_csr_bytes(buf, off, " ", device=True), host)


def test_device_csr_degenerate_shapes():
    """shapes that stress the byte-stream formulation: nothing but separators, a million empty rows, one row of megabytes,
    separators and token bytes above 0x7f, NUL bytes inside tokens"""
    # only separators / only empty rows
    _same(_lib.build_csr_device(["   ", " ", ""], " "), orc.sparse_feature_matrix(["   ", " ", ""], " "))
    ip, ix, nv = _lib.build_csr_device([""] * 1_000_000, " ")
    assert ip[-1] == 0 and len(ip) == 1_000_001 and nv == 0 and len(ix) == 0
    # one row of ~6 MB (800k tokens, 5k distinct) between two small ones
    rng = np.random.default_rng(3)
    big = " ".join(f"T{int(x)}" for x in rng.integers(0, 5000, 800_000))
    rows = ["T1 T2", big, "T2 T4999 X"]
    _same(_lib.build_csr_device(rows, " "), orc.sparse_feature_matrix(rows, " "))
    # a separator byte above 0x7f and token bytes of every value except it (latin-1 keeps one byte per character on both sides)
    sep = b"\xfe"
    toks = [bytes([b]) * (1 + b % 5) for b in range(256) if b != 0xFE]
    raw_rows = [sep.join(toks[i:i + 40]) for i in range(0, len(toks), 40)] + [sep.join(reversed(toks))]
    buf = b"".join(raw_rows)
    off = np.zeros(len(raw_rows) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(r) for r in raw_rows])
    lib = _lib.load()
    import ctypes as C

    def run(fn):
        indptr = np.zeros(len(raw_rows) + 1, dtype=np.int32)
        out, nnz, nvv = _lib.c_i32p(), C.c_int64(), C.c_int32()
        rc = fn(buf, off.ctypes.data_as(_lib.c_i64p), len(raw_rows), sep, 1, indptr.ctypes.data_as(_lib.c_i32p), C.byref(out),
                C.byref(nnz), C.byref(nvv))
        assert rc == 0, lib.bfk_last_error()
        idx = np.ctypeslib.as_array(out, shape=(max(nnz.value, 1),))[: nnz.value].copy()
        lib.bfk_free(out)
        return indptr, idx, nvv.value

    _same(run(lib.bfk_build_csr_device), run(lib.bfk_build_csr))
    # the oracle agrees with both (bytes in, no decoding anywhere)
    ob = orc.lib()
    o_ip = np.zeros(len(raw_rows) + 1, dtype=np.int32)
    o_out, o_nnz, o_nv = orc.c_i32p(), C.c_int64(), C.c_int32()
    assert ob.orc_build_csr(buf, off.ctypes.data_as(orc.c_i64p), len(raw_rows), sep, 1, o_ip.ctypes.data_as(orc.c_i32p),
                            C.byref(o_out), C.byref(o_nnz), C.byref(o_nv)) == 0
    o_ix = np.ctypeslib.as_array(o_out, shape=(max(o_nnz.value, 1),))[: o_nnz.value].copy()
    ob.orc_free(o_out)
    _same(run(lib.bfk_build_csr_device), (o_ip, o_ix, o_nv.value))


def test_device_csr_a_million_one_byte_rows():
    """rows that abut without any separator, one token per byte: the token list of a wave holds a token per byte"""
    rng = np.random.default_rng(4)
    rows = [chr(65 + int(x)) for x in rng.integers(0, 26, 1_000_000)]
    ip, ix, nv = _lib.build_csr_device(rows, " ")
    assert nv == 26 and np.array_equal(ip, np.arange(1_000_001, dtype=np.int32))
    first = {}
    want = np.fromiter((first.setdefault(r, len(first)) for r in rows), dtype=np.int32, count=len(rows))
    assert np.array_equal(ix, want)
