"""Wall clock of `python -m breakfast_amd` on a table written with csv.QUOTE_ALL against the same table unquoted, and against the
pandas mirror (BFK_NO_FASTPATH=1) — VERDICT r04 "missing" 3: a quote used to send the whole run to the mirror.
usage: python tools/quoted_wall.py [rows]"""
import csv
import hashlib
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rows = synth.generate_profiles(n, seed=20240601)
d = Path(tempfile.mkdtemp(prefix="bfk_quoted_"))
for name, q in (("plain", csv.QUOTE_MINIMAL), ("quoted", csv.QUOTE_ALL)):
    with open(d / f"{name}.tsv", "w", newline="") as f:
        w = csv.writer(f, delimiter="\t", quoting=q, lineterminator="\n")
        w.writerow(["accession", "dna_profile"])
        w.writerows((f"seq{i}", r) for i, r in enumerate(rows))
root = str(Path(__file__).resolve().parent.parent)
for name, env in (("plain", {}), ("quoted", {}), ("quoted", {"BFK_NO_FASTPATH": "1"})):
    best, sha = 1e9, None
    for k in range(3):
        out = d / f"out_{name}_{len(env)}_{k}"
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "breakfast_amd", "--input-file", str(d / f"{name}.tsv"), "--outdir", str(out), "--max-dist", "1"],
                       check=True, stdout=subprocess.DEVNULL, cwd=root, env={**os.environ, **env, "PYTHONPATH": root})
        best = min(best, time.perf_counter() - t0)
        sha = hashlib.sha256((out / "clusters.tsv").read_bytes()).hexdigest()[:16]
    print(f"{n} rows, {name:6s} {'pandas mirror' if env else 'native path  '}: {best:.3f} s (best of 3)  clusters.tsv {sha}", flush=True)
