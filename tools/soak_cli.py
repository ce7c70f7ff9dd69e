"""Soak of the CLI's native paths in ONE process (development tooling): random tables — synthetic profiles with duplicates, stray and
empty tokens, NA features, quoted / unquoted fields, CRLF or LF — under random feature types and filter options through
fastpath.run on the device stages and on the host stages (same stdout, same clusters.tsv), token separators of one and of several bytes, and chains of side-car cache runs
(write, grow, lose rows, another max-dist) on the device stages against the list path.  The default context, the preload thread
and the table machinery are reused from run to run: state that leaks from one run into the next shows here.
usage (GPU box): python tools/soak_cli.py [seconds] [seed]"""
import csv
import io
import os
import sys
import tempfile
import time
from contextlib import redirect_stdout
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from breakfast_amd import fastpath  # noqa: E402
from breakfast_amd.synth import generate_family, generate_profiles  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
tmp = Path(tempfile.mkdtemp(prefix="bfk_soak_cli_"))
POOL = {"dna": generate_profiles(20000, seed=seed, p_del=0.05, p_ins=0.02), "aa": generate_family("aa", 6000, seed=seed + 1)}
SEP2 = [" "]   # the token separator of the table in hand
STRAY = ["bogus", "S:N501Y", "xyz!", "a12c", "del:x:1", "C241", "", "  "]


def table(path, kind, n, quoting, crlf, sep2=" "):
    src = POOL[kind]
    a = int(rng.integers(0, len(src) - n))
    rows = list(src[a:a + n])
    for _ in range(int(rng.integers(0, 6))):          # stray / empty tokens, empty and NA features
        i = int(rng.integers(0, n))
        what = int(rng.integers(0, 4))
        if what == 0:
            rows[i] = rows[i] + " " + STRAY[int(rng.integers(0, len(STRAY)))]
        elif what == 1:
            rows[i] = STRAY[int(rng.integers(0, len(STRAY)))] + " " + rows[i]
        elif what == 2:
            rows[i] = ""
        else:
            rows[i] = "NA"
    rows += [rows[int(i)] for i in rng.integers(0, n, size=int(rng.integers(0, n // 4 + 1)))]   # duplicates
    if sep2 != " ":   # (NA stays a missing feature; a token separator of several bytes, or another byte)
        rows = [r if r == "NA" else sep2.join(r.split(" ")) for r in rows]
    with open(path, "w", newline="") as f:
        w = csv.writer(f, delimiter="\t", quoting=quoting, lineterminator="\r\n" if crlf else "\n")
        w.writerow(["accession", "note", "dna_profile"])
        for i, r in enumerate(rows):
            w.writerow([f"s{i}", 'he said "x"' if i % 97 == 0 else "n", r])
    return len(rows)


def run(inp, outdir, env, **kw):
    for k in ("BFK_DEVICE_PREP", "BFK_CACHE_REUSE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    buf = io.StringIO()
    err = None
    with redirect_stdout(buf):
        try:
            ok = fastpath.run(inp, "\t", "accession", "dna_profile", kw["var_type"], SEP2[0], kw["skip_ins"], kw["skip_del"], kw["trim_start"],
                              kw["trim_end"], 29903, kw["d"], kw["mcs"], outdir, kw.get("input_cache"), kw.get("output_cache"))
        except ValueError as e:   # (an all-empty matrix: the reference raises too)
            ok, err = "raised", str(e)
    out = buf.getvalue()
    tsv = (outdir / "clusters.tsv").read_bytes() if (outdir / "clusters.tsv").exists() else None
    return ok, err, out[out.index("Number of sequences"):] if "Number of sequences" in out else out, tsv


t_end = time.time() + budget
n_runs = n_chains = 0
while time.time() < t_end:
    kind = "dna" if rng.random() < 0.75 else "aa"
    var_type = {"dna": str(rng.choice(["covsonar_dna", "covsonar_dna", "raw"])), "aa": str(rng.choice(["covsonar_aa", "raw"]))}[kind]
    filt = var_type != "raw" and rng.random() < 0.8
    kw = dict(var_type=var_type, skip_ins=bool(filt and rng.integers(0, 2)), skip_del=bool(filt and rng.integers(0, 2)),
              trim_start=264 if filt and kind == "dna" else 0, trim_end=228 if filt and kind == "dna" else 0,
              d=int(rng.choice([1, 1, 1, 2, 3])), mcs=int(rng.choice([1, 2, 2, 3])))
    quoting = [csv.QUOTE_MINIMAL, csv.QUOTE_ALL][int(rng.integers(0, 2))]
    inp = tmp / f"in{n_runs}.tsv"
    SEP2[0] = " " if rng.random() < 0.6 else str(rng.choice([",", ", ", "; ", " | ", "||", "--", "  "]))   # (folded on the device when it has several bytes)
    n = table(inp, kind, int(rng.integers(20, 6000)), quoting, bool(rng.integers(0, 2)), SEP2[0])
    a = run(inp, tmp / f"d{n_runs}", {"BFK_DEVICE_PREP": "1"}, **kw)
    b = run(inp, tmp / f"h{n_runs}", {"BFK_DEVICE_PREP": "0"}, **kw)
    assert a == b, ("device stages differ from host stages", n_runs, kw, n, a[:3], b[:3])
    n_runs += 1
    if a[0] is True and rng.random() < 0.4:   # a chain of side-car runs on this table: write, grow / shrink, reuse
        n_chains += 1
        text = inp.read_text().splitlines(keepends=True)
        head, body = text[0], text[1:]
        cut = max(1, len(body) * 3 // 4)
        first = tmp / f"c{n_runs}_a.tsv"
        first.write_text(head + "".join(body[:cut]))
        grown = inp
        lost = tmp / f"c{n_runs}_l.tsv"
        lost.write_text(head + "".join(body[len(body) // 10:]))
        caches, files = {}, {}
        for flow, env in (("dev", {}), ("list", {"BFK_CACHE_REUSE": "1"})):
            c1, c2, c3 = (tmp / f"c{n_runs}_{flow}_{k}.bfkc" for k in (1, 2, 3))
            r1 = run(first, tmp / f"c{n_runs}_{flow}_o1", env, **kw, output_cache=c1)
            r2 = run(grown, tmp / f"c{n_runs}_{flow}_o2", env, **kw, input_cache=c1, output_cache=c2)
            r3 = run(lost, tmp / f"c{n_runs}_{flow}_o3", env, **kw, input_cache=c2, output_cache=c3)
            kw2 = dict(kw, d=kw["d"] % 3 + 1)
            r4 = run(grown, tmp / f"c{n_runs}_{flow}_o4", env, **kw2, input_cache=c3)
            caches[flow] = (r1, r2, r3, r4)
            files[flow] = [c.read_bytes() if c.exists() else None for c in (c1, c2, c3)]
        for k, (x, y) in enumerate(zip(caches["dev"], caches["list"])):
            assert x == y, ("side-car chain: device stages differ from the list path at run", k + 1, n_runs, kw, x[:3], y[:3])
        for k, (x, y) in enumerate(zip(files["dev"], files["list"])):
            assert x == y, ("side-car chain: the caches written differ at run", k + 1, n_runs, kw)
        assert caches["dev"][1][3] == a[3], "a grown input through an exact cache is the run without a cache"
    if n_runs % 10 == 0:
        print(f"[soak_cli] {n_runs} tables, {n_chains} cache chains", flush=True)
print(f"[soak_cli] done: {n_runs} tables on device and host stages, {n_chains} side-car chains on both cache paths: all equal")
