#!/bin/bash
# kernel trace summary of `bench.py --quick` (per-kernel calls and mean duration); run through gpurun
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_q
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_q -o q -- python3 $root/bench.py --steps 100 --warmup 10 --quick "$@" > $root/gpurun_out/prof_q.json 2>/dev/null
python3 - $root <<'PY'
import csv, glob, sys, json
root = sys.argv[1]
f = glob.glob(root + "/gpurun_out/prof_q/**/*kernel_stats.csv", recursive=True)[0]
b = json.loads(open(root + "/gpurun_out/prof_q.json").read().strip().splitlines()[-1])
print("ms_per_step", round(b["ms_per_step"], 4), "resident", round(b["resident_csr"]["ms_per_step"], 4))
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"].split("(")[0][:44].ljust(46), r["Calls"].rjust(6), f"{float(r['AverageNs']) / 1000:8.2f} us", r["Percentage"])
PY
