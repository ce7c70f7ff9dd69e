#!/bin/bash
# rocprofv3 evidence for the device tokeniser (bfk_text.hip), run through gpurun on a GPU box; results land in
# gpurun_out/profiles/ (copy them into profiles/):
#   <tag>_tok_line.json          tools/text_bench.py's line (no profiler attached)
#   <tag>_tok_kernel_stats.csv   kernel trace + stats of the same command
#   <tag>_tok_pmc_per_launch.json  FETCH_SIZE / WRITE_SIZE (KiB per launch, separate --pmc passes) + SQ counters per kernel,
#                                stamped with the source digest and the workload key tok_<rows>
# usage: tools/profile_text.sh <tag> [rows]
set -eo pipefail
tag=${1:?tag}
rows=${2:-100000}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_tok_$tag
dst=$root/gpurun_out/profiles
mkdir -p "$out" "$dst"
cd /tmp
export TMPDIR=/tmp
python3 "$root/tools/text_bench.py" "$rows" > "$dst/${tag}_tok_line.json"
echo "[profile] line written"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- \
    python3 "$root/tools/text_bench.py" "$rows" > /dev/null
echo "[profile] kernel trace done"
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout -k 10 240 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -o p -- \
        python3 "$root/tools/text_bench.py" "$rows" > /dev/null
    echo "[profile] pmc pass $i done"
done
python3 - "$out" "$dst/$tag" "$rows" "$root" <<'PYEOF'
import csv, glob, json, sys, collections
out, dst, rows, root = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
sys.path.insert(0, root)
import bench
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
open(dst + "_tok_kernel_stats.csv", "w").write(open(st).read())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "k_tok" in name or "k_voc" in name or "k_scan_single" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
res["_meta"] = {"source_digest": bench.kernel_source_digest(), "workload": f"tok_{rows}",
                "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch as rocprofv3 reports them (k_tok_hash: mean over its two "
                         "launches per build — the first 4 KiB alone, then the rest); SQ_* raw"}
json.dump(res, open(dst + "_tok_pmc_per_launch.json", "w"), indent=1)
for r in list(csv.DictReader(open(st)))[:16]:
    print(f'{r["Name"].split("(")[0][:36]:38s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1000:8.2f} us')
PYEOF
