#!/bin/bash
# SQ counters of k_tok_hashb (tools/tok_stamps.py under rocprofv3 --pmc, separate passes); run through gpurun
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_hashb
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "$@"; do
    i=$((i + 1))
    timeout -k 10 120 rocprofv3 --pmc $pmc --output-format csv -d "$out/p$i" -o p -- python3 "$root/tools/tok_stamps.py" > /dev/null 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("bfk::", "").replace("void ", "")
        if "k_tok_hash" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v) / len(v):14.0f}")
PY
