#!/bin/bash
# per-launch timeline of the text step (see step_timeline.py); run through gpurun
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_q
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_q -o q -- python3 $root/bench.py --steps 100 --warmup 10 --quick --contexts 1 "$@" > $root/gpurun_out/prof_q.json 2>/dev/null
python3 $root/tools/step_timeline.py $root
