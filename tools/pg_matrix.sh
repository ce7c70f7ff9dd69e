#!/bin/bash
# band kernels vs prefix groups over rows x max_dist (bench.py --quick lines -> gpurun_out/pg_matrix.txt); run on the GPU box
out=gpurun_out/pg_matrix.txt
: > $out
for rows in ${ROWS:-100000 300000 1000000}; do
  for d in ${DIST:-2 3 4 5}; do
    for path in allpairs prefix; do
      timeout -k 10 200 python bench.py --rows $rows --max-dist $d --indels --quick --path $path --steps 10 > gpurun_out/pgm.json 2> gpurun_out/pgm.err || { echo "$rows $d $path FAILED" >> $out; continue; }
      python - "$rows" "$d" "$path" >> $out <<PY
import json, sys
b = json.loads(open("gpurun_out/pgm.json").read().strip().splitlines()[-1])
print(*sys.argv[1:], round(b["ms_per_step"], 3), {k: round(v, 3) for k, v in b["phases_ms"].items()}, b["counters"]["n_candidates"], b["counters"]["n_connected"])
PY
    done
  done
done
cat $out
