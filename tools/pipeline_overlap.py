"""How the steps of a pipelined run (bench.py --contexts k) share the chip, from a rocprofv3 kernel trace: per stream the steps it ran,
and over the busiest second of the trace how long 0, 1, 2, 3, ... kernels were in flight at once, the sum of kernel durations against
the wall time they covered (mean kernels in flight) and the steps completed per millisecond.  Run through gpurun:
    tools/pipeline_overlap.sh [contexts]"""
import csv
import glob
import sys
from collections import Counter, defaultdict

root = sys.argv[1]
f = glob.glob(root + "/gpurun_out/prof_p/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "bfk::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: bench.py --quick ends with 16 profiled text steps and the resident-CSR legs (no tokeniser); the 400 text steps in
# front of those 16 are the timed ones (pre-roll, the calibration instances and the warm-up lie further in front)
clears = [int(r["Start_Timestamp"]) for r in rows if "k_tok_clear" in r["Kernel_Name"]]
lo, hi = clears[max(0, len(clears) - 16 - 380)], clears[max(0, len(clears) - 16 - 20)]
rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
per_stream = defaultdict(list)
for r in rows:
    per_stream[r["Stream_Id"]].append(r)
print(f"{len(rows)} kernels on {len(per_stream)} streams over {(t1 - t0) / 1e6:.2f} ms")
for s, rs in sorted(per_stream.items(), key=lambda kv: -len(kv[1])):
    n_steps = sum(1 for r in rs if "k_flatten" in r["Kernel_Name"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"  stream {s:>3s} (queue {rs[0]['Queue_Id']}): {len(rs):6d} kernels, {n_steps:5d} steps, busy {busy / (t1 - t0):.2f} of the time")
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
level, last, hist = 0, t0, Counter()
for t, d in ev:
    hist[level] += t - last
    last = t
    level += d
tot = sum(hist.values())
print("kernels in flight -> share of the time: " + ", ".join(f"{k}: {v / tot:.3f}" for k, v in sorted(hist.items())))
ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
steps = sum(1 for r in rows if "k_flatten" in r["Kernel_Name"])
print(f"sum of kernel durations / wall = {ksum / (t1 - t0):.2f} kernels in flight on average; {steps} steps = {(t1 - t0) / 1e3 / max(steps, 1):.1f} us per step "
      f"(under the profiler), {ksum / 1e3 / max(steps, 1):.1f} us of kernel time per step")
