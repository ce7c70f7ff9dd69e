#!/bin/bash
# The deferred phase of k_tok_hash, old against new, on several input shapes: BFK_TOK_DEBUG 0 = as built (free-looking slots claimed
# by compare-and-swap in the round trip of the loads, two batches per round), 64 = no claim, 128 = one batch per round, 192 = both
# off (round 4's phase).  All four give valid results.  -> ms of the hash phase (HIP events, median of 9 builds); through gpurun
# usage: tools/tok_claim_ab.sh [rows]      SHAPES="tree forest sorted aa long star" DBGS="0 64 128 192"
cd "$(dirname "$0")/.."
rows=${1:-100000}
for dbg in ${DBGS:-0 64 128 192}; do
  BFK_TOK_DEBUG=$dbg python - "$rows" "$dbg" <<'PY'
import os, sys
sys.path.insert(0, ".")
from breakfast_amd import _lib
from breakfast_amd.synth import generate_family, generate_profiles
n = int(sys.argv[1])
out = []
for shape in os.environ.get("SHAPES", "tree forest sorted aa long star").split():
    if shape == "forest":  # ten independent trees one after the other: a root's tokens first appear at row k * n / 10
        rows = [r for k in range(10) for r in generate_profiles(n // 10, seed=1000 + k)]
    elif shape == "sorted":
        rows = sorted(generate_profiles(n))
    elif shape in ("long", "star", "aa"):
        rows = generate_family(shape, n if shape != "long" else n // 2)
    else:
        rows = generate_profiles(n)
    rows = list(dict.fromkeys(rows))
    buf, off = _lib.pack_rows(rows)
    ctx = _lib.Context(0)
    ctx.set_profiling(True)
    ph = []
    for _ in range(9):
        nnz, nv = ctx.build_csr(buf, off, " ")
        ph.append(ctx.text_stats())
    out.append(f"{shape} {len(rows)} rows V={nv}: hash {sorted(p['ms_hash'] for p in ph)[4]:.4f} total {sorted(p['ms_total'] for p in ph)[4]:.4f}")
    ctx.close()
print(f"BFK_TOK_DEBUG={sys.argv[2]:>3}: " + " | ".join(out), flush=True)
PY
done
