"""Soak of the text steps (development tooling, uses the oracle as the checker — like the tests): random batches of random shapes
through one-context steps and TextPipelines of random depth, with syncs at random points, slots-as-ids switched on and off,
batches outside what a device-driven step assumes (rows of > 128 tokens, all-empty inputs, runs of identical rows that make the
join give up, vocabularies that make the table grow) in between ordinary ones; every batch's labels against the oracle's.
usage (GPU box): python tools/soak_text.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from breakfast_amd import _lib  # noqa: E402
from breakfast_amd.distributed import TextPipeline  # noqa: E402
from breakfast_amd.synth import generate_family, generate_profiles  # noqa: E402
from oracle import ref_port as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
POOL = {"default": list(dict.fromkeys(generate_profiles(30000, seed=seed))),
        "indels": list(dict.fromkeys(generate_profiles(12000, seed=seed + 1, p_del=0.05, p_ins=0.01))),
        "long": list(dict.fromkeys(generate_family("long", 4000, seed=seed + 2))),
        "star": list(dict.fromkeys(generate_family("star", 12000, seed=seed + 3))),
        "aa": list(dict.fromkeys(generate_family("aa", 8000, seed=seed + 4)))}


def batch():
    kind = rng.choice(["default", "default", "default", "indels", "long", "star", "aa", "identical", "empty", "bigvocab", "tiny", "mixed"])
    if kind in POOL:
        rows = POOL[kind]
        n = int(rng.integers(1, len(rows)))
        a = int(rng.integers(0, len(rows) - n + 1))
        return kind, rows[a:a + n]
    if kind == "identical":   # a run of identical rows in the middle: the join's duplicate list overflows -> give-up -> redo
        rows = POOL["default"]
        n = int(rng.integers(400, 3000))
        out = rows[:n]
        k = int(rng.integers(1, n))
        return kind, out[:k] + [rows[k]] * int(rng.integers(280, 400)) + out[k:]
    if kind == "empty":
        return kind, [""] * int(rng.integers(1, 50))
    if kind == "bigvocab":    # every token distinct: the vocabulary table has to grow
        n = int(rng.integers(2000, 9000))
        return kind, [" ".join(f"W{r}_{j}" for j in range(int(rng.integers(5, 30)))) for r in range(n)]
    if kind == "tiny":
        return kind, [["A1C", "A1C G2T", "", "G2T", "Q9R A1C G2T", "X", "X X"][int(i)] for i in rng.integers(0, 7, size=int(rng.integers(1, 12)))]
    rows = []
    for k in ("default", "long", "aa"):
        src = POOL[k]
        a = int(rng.integers(0, len(src) - 300))
        rows += src[a:a + int(rng.integers(50, 300))]
    order = rng.permutation(len(rows))
    return kind, [rows[int(i)] for i in order]


def on_device(rows):
    buf, off = _lib.pack_rows(rows)
    d_text = torch.full((_lib.text_device_bytes(len(buf)),), 0x41, dtype=torch.uint8, device="cuda")
    if len(buf):
        d_text[: len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    return len(buf), d_text, torch.from_numpy(off).cuda()


def want_labels(rows, d):
    ip, ix, _ = orc.sparse_feature_matrix(rows, " ")
    if len(ix) == 0:
        return None
    return orc.cluster_csr(ip, ix, d, n_threads=16)["labels"]


t_end = time.time() + budget
n_batches = n_rounds = 0
kinds = {}
while time.time() < t_end:
    n_rounds += 1
    depth = int(rng.integers(1, 5))
    pipe = TextPipeline(0, depth)
    any_ids = bool(rng.integers(0, 2))
    for e in pipe.engines:
        e.ctx.set_token_ids(any_ids)
    pending = []
    for _ in range(int(rng.integers(3, 14))):
        kind, rows = batch()
        d = int(rng.choice([1, 1, 1, 1, 2, 3]))
        want = want_labels(rows, d)
        if want is None:   # (an all-empty input: the reference cannot build a matrix; the step reports an error at sync)
            continue
        T, d_text, d_off = on_device(rows)
        lab = torch.full((len(rows),), -7, dtype=torch.int32, device="cuda")
        pipe.step_text(d_text.data_ptr(), T, d_off.data_ptr(), len(rows), " ", d, lab, inputs_ready=True, want_event=False)
        pending.append((kind, d, len(rows), lab, want, d_text, d_off))
        kinds[kind] = kinds.get(kind, 0) + 1
        if rng.random() < 0.3:
            pipe.sync()
            for k_, d_, n_, lab_, want_, *_ in pending:
                assert np.array_equal(lab_.cpu().numpy(), want_), ("labels differ", k_, d_, n_, depth, any_ids, seed, n_rounds)
                n_batches += 1
            pending = []
    pipe.sync()
    for k_, d_, n_, lab_, want_, *_ in pending:
        assert np.array_equal(lab_.cpu().numpy(), want_), ("labels differ", k_, d_, n_, depth, any_ids, seed, n_rounds)
        n_batches += 1
    pipe.close()
    # one context, entry points mixed: text steps, a CSR built and clustered by hand, downloads in between, slots-as-ids and the
    # candidate generator switched while steps are open
    ctx = _lib.Context(0)
    pending = []
    for _ in range(int(rng.integers(3, 12))):
        kind, rows = batch()
        d = int(rng.choice([1, 1, 1, 2, 3]))
        want = want_labels(rows, d)
        if want is None:
            continue
        T, d_text, d_off = on_device(rows)
        lab = torch.full((len(rows),), -7, dtype=torch.int32, device="cuda")
        op = int(rng.integers(0, 4))
        if rng.random() < 0.3:
            ctx.set_token_ids(bool(rng.integers(0, 2)))
        if rng.random() < 0.3:
            ctx.set_candidate_path(str(rng.choice(["auto", "allpairs", "join" if d == 1 else "auto"])))
        if op <= 1:
            ctx.cluster_text_device(d_text.data_ptr(), T, d_off.data_ptr(), len(rows), " ", d, lab.data_ptr())
        elif op == 2:
            ctx.build_csr_device(d_text.data_ptr(), T, d_off.data_ptr(), len(rows), " ")
            ctx.cluster(d, lab.data_ptr())
        else:
            ctx.build_csr_device(d_text.data_ptr(), T, d_off.data_ptr(), len(rows), " ")
            ip, ix = ctx.download_csr()
            w = orc.sparse_feature_matrix(rows, " ")
            assert np.array_equal(ip, w[0]) and np.array_equal(ix, w[1]), ("CSR differs", kind, len(rows), seed, n_rounds)
            ctx.cluster(d, lab.data_ptr())
        pending.append((kind, d, len(rows), lab, want, d_text, d_off))
        kinds[kind] = kinds.get(kind, 0) + 1
        if rng.random() < 0.35:
            ctx.sync()
            for k_, d_, n_, lab_, want_, *_ in pending:
                assert np.array_equal(lab_.cpu().numpy(), want_), ("labels differ (one context)", k_, d_, n_, seed, n_rounds)
                n_batches += 1
            pending = []
    ctx.sync()
    for k_, d_, n_, lab_, want_, *_ in pending:
        assert np.array_equal(lab_.cpu().numpy(), want_), ("labels differ (one context)", k_, d_, n_, seed, n_rounds)
        n_batches += 1
    ctx.close()
    if n_rounds % 5 == 0:
        print(f"[soak] {n_rounds} pipelines, {n_batches} batches checked, {kinds}", flush=True)
print(f"[soak] done: {n_rounds} pipelines, {n_batches} batches, all labels equal the oracle's; {kinds}")
