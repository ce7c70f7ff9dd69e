#!/bin/bash
# rocprofv3 kernel stats of the device tokeniser (tools/text_bench.py) -> gpurun_out/prof_text_<rows>/
# usage (through gpurun): tools/prof_text.sh <rows>
set -eo pipefail
rows=${1:-100000}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_text_$rows
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- \
    python3 "$root/tools/text_bench.py" "$rows" > "$out/line.json"
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
open(out + "/kernel_stats.csv", "w").write(open(st).read())
for r in list(csv.DictReader(open(st)))[:14]:
    print(f'{r["Name"].split("(")[0][:40]:42s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1000:9.2f} us')
PY
