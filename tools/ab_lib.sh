#!/bin/bash
# A/B of library builds on one box: tools/ab_lib.sh "suffix .." "rows .." max_dist [bench args]   (breakfast_amd/libbfk<suffix>.so; "-" = the tree's)
sfx=$1; rows=$2; d=$3; shift 3
for v in $sfx; do
  if [ "$v" = "-" ]; then lib=$PWD/breakfast_amd/libbfk.so; else lib=$PWD/breakfast_amd/libbfk$v.so; fi
  for r in $rows; do
    BFK_LIB=$lib timeout -k 10 120 python bench.py --rows $r --max-dist $d --indels --quick --steps 10 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "lib$v rows=$r FAILED"; continue; }
    python - "lib$v" "$r" <<'PY'
import json, sys
b = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(b["ms_per_step"], 3), {k: round(v, 3) for k, v in b["phases_ms"].items()}, b["counters"]["n_candidates"])
PY
  done
done
