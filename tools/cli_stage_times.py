"""Stage times of the CLI's native path (BFK_FRONT_TIMING=1 lines on stderr) + the process wall, device stages against host
stages.  Run on a GPU box:  python tools/cli_stage_times.py [n_rows] [max_dist]"""
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
for name, env in (("device", {}), ("device", {}), ("device", {}), ("host", {"BFK_DEVICE_PREP": "0"}), ("host", {"BFK_DEVICE_PREP": "0"})):
    t = time.time()
    r = subprocess.run([sys.executable, "-m", "breakfast_amd", "--input-file", str(inp), "--outdir", str(tmp / name), "--max-dist", d],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, cwd=root,
                       env=dict(os.environ, BFK_FRONT_TIMING="1", **env))
    dt = time.time() - t
    lines = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("[bfk")]
    print(f"== {name} stages: wall {dt:.3f} s")
    print("\n".join(lines), flush=True)
