#!/bin/bash
# two-phase verify (BFK_VERIFY_PHASES = p: first 1/p of every queue shard, compress, the rest; BFK_VERIFY_PHASE2 = union
# of the second phase: 0 find + hook, 2 first hops then splicing).  usage: tools/phases_ab.sh ROWS DIST [bench args]
cd "$(dirname "$0")/.."
rows=$1; d=$2; shift 2
for cfg in ${CFGS:-"1_0" "8_2" "4_2" "2_2" "8_0" "4_0"}; do
  ph=${cfg%_*}; u=${cfg#*_}
  BFK_VERIFY_PHASES=$ph BFK_VERIFY_PHASE2=$u python bench.py --rows $rows --max-dist $d "$@" --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline > gpurun_out/ph.json 2>/dev/null
  python -c "
import json; b=json.loads(open('gpurun_out/ph.json').read()); print('phases=$ph union2=$u rows=$rows d=$d', round(b['ms_per_step'],4), round(b['phases_ms']['ms_verify'],4), b['result']['labels_crc'], b['counters']['n_edges'])"
done
