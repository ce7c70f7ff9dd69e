#!/bin/bash
# k_tok_hash with parts switched off (BFK_TOK_DEBUG: timing only, results invalid) -> where its time goes
# usage (through gpurun): tools/tok_ab.sh [rows]
cd "$(dirname "$0")/.."
rows=${1:-100000}
for dbg in ${DBGS:-0 1 2 4 8}; do
  echo "== BFK_TOK_DEBUG=$dbg"
  BFK_TOK_DEBUG=$dbg python - "$rows" <<'PY'
import sys, time
sys.path.insert(0, ".")
from breakfast_amd import _lib
from breakfast_amd.synth import generate_profiles
rows = list(dict.fromkeys(generate_profiles(int(sys.argv[1]))))
buf, off = _lib.pack_rows(rows)
ctx = _lib.Context(0)
ctx.set_profiling(True)
ph = []
for _ in range(7):
    try:
        ctx.build_csr(buf, off, " ")
    except Exception as e:
        pass
    ph.append(ctx.text_stats())
print({k: round(sorted(p[k] for p in ph)[3], 4) for k in ("ms_scan", "ms_hash", "ms_ids", "ms_total")})
PY
done
