#!/bin/bash
# join vs all-pairs (BFK_JOIN=0) at several sizes
cd "$(dirname "$0")/.."
for cfg in "1 100000" "0 100000" "1 600000" "0 600000" "1 1000000" "0 1000000" "1 2000000" "0 2000000"; do
  set -- $cfg
  BFK_JOIN=$1 python bench.py --rows $2 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/jb.json 2>/dev/null
  python - <<PY
import json
b=json.loads(open("gpurun_out/jb.json").read()); p=b["phases_ms"]
print("join=%s rows=%-8s step %.4f  prep %.4f pairs %.4f verify %.4f flatten %.4f" % ("$1", "$2", b["ms_per_step"], p["ms_prep"], p["ms_prefilter"], p["ms_verify"], p["ms_flatten"]))
PY
done
