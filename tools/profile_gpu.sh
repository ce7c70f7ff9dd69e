#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block is checked against (run through gpurun on a GPU box):
# Everything lands in gpurun_out/profiles/ (gpurun merges only gpurun_out/ back); copy the files into profiles/.
#   1. the bench line itself (no profiler attached)  -> profiles/<tag>_bench.json
#   2. kernel trace + stats of `python3 bench.py --quick --contexts 1` (one context: every step behind the one before, so a kernel's
#      duration is its own and not that of two steps sharing the chip)  -> profiles/<tag>_kernel_stats.csv
#   3. PMC passes (separate runs, never combined with traces): FETCH_SIZE / WRITE_SIZE, then the SQ counters
#      -> profiles/<tag>_pmc_per_launch.json (per-kernel means per launch; FETCH_SIZE/WRITE_SIZE in KiB as
#      rocprofv3 reports them), stamped with the digest of the kernel sources and the workload key: bench.py
#      quotes `roofline.traffic` from it only when both match the running build
# usage: BFK_COMMIT=<sha> tools/profile_gpu.sh <tag> [--skip-bench] [bench args...]
set -eo pipefail
tag=${1:?tag}
shift || true
skip_bench=0
if [ "$1" = "--skip-bench" ]; then skip_bench=1; shift; fi
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
dst=$root/gpurun_out/profiles
mkdir -p "$out" "$dst"
cd /tmp
export TMPDIR=/tmp

if [ $skip_bench = 0 ]; then
    python3 "$root/bench.py" --detail "$dst/${tag}_bench_detail.json" "$@" > "$dst/${tag}_bench.json"
    echo "[profile] bench line written (+ the full record: ${tag}_bench_detail.json)"
fi

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- \
    python3 "$root/bench.py" --steps 200 --warmup 20 --quick --contexts 1 "$@" > "$dst/${tag}_bench_under_rocprof.json"
echo "[profile] kernel trace done"

i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout -k 10 180 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -o p -- \
        python3 "$root/bench.py" --steps 20 --warmup 5 --quick --contexts 1 "$@" > /dev/null
    echo "[profile] pmc pass $i done"
done

python3 - "$out" "$dst/$tag" "${BFK_COMMIT:-unknown}" <<'PYEOF'
import csv, glob, json, sys, collections
out, dst, commit = sys.argv[1], sys.argv[2], sys.argv[3]
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
open(dst + "_kernel_stats.csv", "w").write(open(st).read())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
cfg = json.loads(open(dst + "_bench_under_rocprof.json").read().strip().splitlines()[-1])["config"]
res["_meta"] = {"commit": commit, "source_digest": cfg["kernel_source_digest"], "workload": cfg["workload_key"],
                "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch as rocprofv3 reports them; SQ_* raw"}
json.dump(res, open(dst + "_pmc_per_launch.json", "w"), indent=1)
for r in list(csv.DictReader(open(st)))[:12]:
    print(f'{r["Name"].split("(")[0][:36]:38s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1000:8.2f} us')
PYEOF
