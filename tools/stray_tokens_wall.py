"""the CLI's wall on a 1M-row file, clean against the same file with 100 stray tokens (tokens that match
no pattern of the feature type: the device prepare lists them; before round 5 it declined such a file and the host stages ran).
usage (GPU box): python tools/stray_tokens_wall.py [rows]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from breakfast_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
tmp = Path(tempfile.mkdtemp(prefix="bfk_stray_"))
clean = tmp / "clean.tsv"
synth.generate_tsv(clean, n)
lines = clean.read_text().splitlines()
rng = np.random.default_rng(1)
for k in rng.choice(np.arange(1, len(lines)), 100, replace=False):
    acc, prof = lines[k].split("\t")
    toks = prof.split(" ")
    toks.insert(int(rng.integers(0, len(toks) + 1)), ["S:N501Y", "stray", "n/a", "A12", "del:5"][int(rng.integers(5))])
    lines[k] = acc + "\t" + " ".join(toks)
stray = tmp / "stray.tsv"
stray.write_text("\n".join(lines) + "\n")


def run(inp, tag, env=None):
    ts, out_txt, sha = [], None, None
    for i in range(4):
        out = tmp / f"out_{tag}{i}"
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "breakfast_amd", "--input-file", str(inp), "--outdir", str(out), "--max-dist", "1"],
                           cwd=str(ROOT), capture_output=True, text=True, env={**os.environ, **(env or {})})
        ts.append(time.perf_counter() - t0)
        assert r.returncode == 0, r.stderr[-400:]
        out_txt = r.stdout
        sha = hashlib.sha256((out / "clusters.tsv").read_bytes()).hexdigest()[:16]
    return min(ts), sorted(ts)[len(ts) // 2], out_txt, sha


c = run(clean, "c")
s = run(stray, "s")
h = run(stray, "h", {"BFK_DEVICE_PREP": "0"})
inv = [ln for ln in s[2].splitlines() if ln.startswith("Skipping invalid")]
print(f"{n} rows: clean {c[0]:.3f} s (median {c[1]:.3f}); 100 stray tokens, device stages {s[0]:.3f} s (median {s[1]:.3f}), "
      f"{len(inv)} lines printed; the same file on the host stages {h[0]:.3f} s (median {h[1]:.3f}); "
      f"stdout equal device / host: {s[2] == h[2]}, clusters.tsv equal: {s[3] == h[3]}")
