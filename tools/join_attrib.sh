#!/bin/bash
# timing attribution of the variant join: bench phases with parts of the kernels switched off (BFK_JOIN_DEBUG:
# 1 no settle, 2 no table probe, 4 no queueing of bitmap hits, 8 no table insert, 16 no clearing, 32 no unions,
# 128 no scattered bitmap loads, 256 no bitmap atomicOr; results are invalid, only the times mean something).  DBGS="0 7 135" selects runs.
cd "$(dirname "$0")/.."
for dbg in ${DBGS:-0 32 1 3 7 8 24}; do
  BFK_JOIN_DEBUG=$dbg python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/ja_$dbg.json 2>/dev/null
  python - <<PY
import json
b=json.loads(open("gpurun_out/ja_$dbg.json").read()); p=b["phases_ms"]
print("dbg=%-3s step %.4f  jhash %.4f join %.4f verify %.4f flatten %.4f" % ("$dbg", b["ms_per_step"], p["ms_prep"], p["ms_prefilter"], p["ms_verify"], p["ms_flatten"]))
PY
done
