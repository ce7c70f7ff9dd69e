#!/bin/bash
# rocprofv3 kernel stats of `bench.py --quick <args>` -> gpurun_out/prof_q_<tag>/kernel_stats.csv (+ the top kernels on stdout)
# usage (through gpurun): tools/prof_quick.sh <tag> [bench args...]
set -eo pipefail
tag=${1:?tag}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_q_$tag
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- \
    python3 "$root/bench.py" --quick "$@" > "$out/line.json" 2> "$out/err.log"
python3 - "$out" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
open(out + "/kernel_stats.csv", "w").write(open(st).read())
b = json.loads(open(out + "/line.json").read().strip().splitlines()[-1])
print("ms_per_step", round(b["ms_per_step"], 4), b["phases_ms"])
for r in list(csv.DictReader(open(st)))[:14]:
    print(f'{r["Name"].split("(")[0][:44]:46s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1000:9.2f} us')
PY
