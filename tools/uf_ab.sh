#!/bin/bash
# union variants of k_verify (BFK_UF_LINK: 0 find + atomicMin hook, 1 splicing, 2 first hops checked, then splicing)
# usage: tools/uf_ab.sh ROWS DIST [extra bench args]
cd "$(dirname "$0")/.."
rows=$1; d=$2; shift 2
for v in 0 1 2; do
  BFK_UF_LINK=$v python bench.py --rows $rows --max-dist $d "$@" --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline > gpurun_out/uf.json 2>/dev/null
  python -c "
import json; b=json.loads(open('gpurun_out/uf.json').read()); print('link=$v rows=$rows d=$d', round(b['ms_per_step'],4), round(b['phases_ms']['ms_verify'],4), b['result']['labels_crc'], b['counters']['n_edges'])"
done
