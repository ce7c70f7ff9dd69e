#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block is checked against (run through gpurun on a GPU box):
# Everything lands in gpurun_out/profiles/ (gpurun merges only gpurun_out/ back); copy the files into profiles/.
#   1. kernel trace + stats of `python3 bench.py`  -> profiles/<tag>_kernel_stats.csv
#   2. PMC passes (separate runs, never combined with traces): FETCH_SIZE / WRITE_SIZE, then the SQ counters
#      -> profiles/<tag>_pmc_per_launch.json (per-kernel means per launch; FETCH_SIZE/WRITE_SIZE in KiB as
#      rocprofv3 reports them - the guide's gfx950 correction is applied by the reader)
#   3. the bench line itself (no profiler attached)  -> profiles/<tag>_bench.json
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -eo pipefail
tag=${1:?tag}
shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
dst=$root/gpurun_out/profiles
mkdir -p "$out" "$dst"
cd /tmp
export TMPDIR=/tmp

python3 "$root/bench.py" "$@" > "$dst/${tag}_bench.json"
echo "[profile] bench line written"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- \
    python3 "$root/bench.py" --steps 100 --warmup 10 --no-cpu-baseline "$@" > "$dst/${tag}_bench_under_rocprof.json"
echo "[profile] kernel trace done"

i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i + 1))
    timeout -k 10 180 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -o p -- \
        python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null
    echo "[profile] pmc pass $i done"
done

python3 - "$out" "$dst/$tag" <<'EOF'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
open(dst + "_kernel_stats.csv", "w").write(open(st).read())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
json.dump(res, open(dst + "_pmc_per_launch.json", "w"), indent=1)
for r in list(csv.DictReader(open(st)))[:12]:
    print(f'{r["Name"].split("(")[0][:36]:38s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1000:8.2f} us')
EOF
