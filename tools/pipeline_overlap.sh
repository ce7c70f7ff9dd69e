#!/bin/bash
# kernel trace of a pipelined run (see pipeline_overlap.py); run through gpurun
root=$(cd "$(dirname "$0")/.." && pwd)
k=${1:-0}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_p
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_p -o p -- python3 $root/bench.py --steps 400 --warmup 20 --quick --contexts $k > $root/gpurun_out/prof_p.json 2>/dev/null
python3 $root/tools/pipeline_overlap.py $root
