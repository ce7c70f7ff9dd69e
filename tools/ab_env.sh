#!/bin/bash
# A/B of one environment knob over bench.py --quick lines (run on the GPU box):
#   tools/ab_env.sh VAR "v1 v2 .." "rows .." max_dist [extra bench args]     ("-" as a value: variable unset)
var=$1; vals=$2; rows=$3; d=$4; shift 4
for v in $vals; do
  for r in $rows; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    timeout -k 10 200 python bench.py --rows $r --max-dist $d --indels --quick --steps 10 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$var=$v rows=$r FAILED"; tail -2 gpurun_out/ab.err; continue; }
    python - "$var=$v" "$r" <<'PY'
import json, sys
b = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
c = b["counters"]
print(sys.argv[1], sys.argv[2], round(b["ms_per_step"], 3), {k: round(v, 3) for k, v in b["phases_ms"].items()},
      {k: c[k] for k in ("pairs_filtered", "n_candidates", "n_edges", "n_connected")})
PY
  done
done
