#!/bin/bash
# quick GPU loop: parity tests, one bench line, kernel trace summary (run through gpurun)
set -o pipefail
cd "$(dirname "$0")/.."
python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/b1.json || exit 1
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_v -o v -- python3 $root/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > /dev/null 2>&1
cd $root
python - <<PY
import csv,glob,json
b=json.loads(open("gpurun_out/b1.json").read()); print(b["ms_per_step"], b["value"], b["counters"])
f=glob.glob("gpurun_out/prof_v/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]: print(r["Name"][:40], r["Calls"], r["AverageNs"])
PY
