#!/bin/bash
# the three-launch split of k_tok_hash: head units x sample stride -> ms of the hash phase (median of 7 builds); through gpurun
cd "$(dirname "$0")/.."
rows=${1:-100000}
python - "$rows" <<'PY'
import os, sys
sys.path.insert(0, ".")
from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles
n = int(sys.argv[1])
shape = os.environ.get("SHAPE", "tree")
if shape == "forest":  # ten independent trees one after the other: the tokens of tree k's root (carried by all its rows) first appear at row k * n / 10
    rows = [r for k in range(10) for r in generate_profiles(n // 10, seed=1000 + k)]
elif shape == "sorted":
    rows = sorted(generate_profiles(n))
elif shape in ("long", "star", "aa"):
    from breakfast_amd.synth import generate_family
    rows = generate_family(shape, n)
else:
    rows = generate_profiles(n)
rows = list(dict.fromkeys(rows))
print(shape, len(rows), "rows")
buf, off = _lib.pack_rows(rows)
ctx = _lib.Context(0)
ctx.set_profiling(True)
for head, sample in [tuple(int(x) for x in hs.split(":")) for hs in os.environ.get("SPLITS", "16:16 64:0 128:0 256:0 512:0 1024:0 256:16 64:64").split()]:
    if True:
        os.environ["BFK_TOK_HEAD_UNITS"] = str(head)
        os.environ["BFK_TOK_SAMPLE"] = str(sample)
        ph = []
        for _ in range(7):
            ctx.build_csr(buf, off, " ")
            ph.append(ctx.text_stats())
        print(f"head {head:3d} sample {sample:3d}:", {k: round(sorted(p[k] for p in ph)[3], 4) for k in ("ms_hash", "ms_total")}, flush=True)
PY
