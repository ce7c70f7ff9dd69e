#!/bin/bash
cd "$(dirname "$0")/.."
for cfg in "0 100000" "64 100000" "0 300000" "64 300000" "J0 300000" "0 600000" "J0 600000" "64 1000000"; do
  set -- $cfg
  if [ "$1" = "J0" ]; then export BFK_JOIN=0; export BFK_JOIN_DEBUG=0; else unset BFK_JOIN; export BFK_JOIN_DEBUG=$1; fi
  python bench.py --rows $2 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/jb.json 2>/dev/null
  python - <<PY
import json
b=json.loads(open("gpurun_out/jb.json").read()); p=b["phases_ms"]
print("cfg=%-12s step %.4f  prep %.4f pairs %.4f verify %.4f flatten %.4f" % ("$cfg", b["ms_per_step"], p["ms_prep"], p["ms_prefilter"], p["ms_verify"], p["ms_flatten"]))
PY
done
