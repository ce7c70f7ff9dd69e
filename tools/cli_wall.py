"""Wall-clock of the whole CLI (input file -> clusters.tsv) at a BASELINE.json size, native path and pandas path.
Run on a GPU box:  python tools/cli_wall.py [n_rows] [max_dist]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = sys.argv[2] if len(sys.argv) > 2 else "1"
tmp = Path(tempfile.mkdtemp())
inp = tmp / "in.tsv"
synth.generate_tsv(inp, n)
root = str(Path(__file__).resolve().parent.parent)
for name, env in (("native", {}), ("native", {}), ("pandas", {"BFK_NO_FASTPATH": "1"})):
    out = tmp / name
    t = time.time()
    subprocess.run([sys.executable, "-m", "breakfast_amd", "--input-file", str(inp), "--outdir", str(out), "--max-dist", d],
                   stdout=subprocess.DEVNULL, check=True, cwd=root, env=dict(os.environ, **env))
    dt = time.time() - t
    print(f"{name:7s} CLI wall {dt:7.3f} s   n={n} d={d}  sha256 {hashlib.sha256((out / 'clusters.tsv').read_bytes()).hexdigest()}",
          flush=True)
