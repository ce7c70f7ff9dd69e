"""Timeline of ONE text step from a rocprofv3 kernel trace: every launch in order with its median duration and the median gap in
front of it, over all the steps the trace holds (a step = k_tok_clear .. k_flatten).  Run through gpurun:
    tools/step_timeline.sh [bench.py args]"""
import csv
import glob
import statistics
import sys

root = sys.argv[1]
f = glob.glob(root + "/gpurun_out/prof_q/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("bfk::", "")
    if name.startswith("k_tok_clear"):
        cur = []
    if cur is not None:
        cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        if name.startswith("k_flatten"):
            steps.append(cur)
            cur = None
sig = statistics.mode(tuple(n for n, _, _ in s) for s in steps)
steps = [s for s in steps if tuple(n for n, _, _ in s) == sig][5:]
print(f"{len(steps)} steps of {len(sig)} launches; median step (first start .. last end) "
      f"{statistics.median(s[-1][2] - s[0][1] for s in steps) / 1000:.1f} us")
tot_k = tot_g = 0.0
for i, name in enumerate(sig):
    dur = statistics.median(s[i][2] - s[i][1] for s in steps) / 1000
    gap = statistics.median(s[i][1] - s[i - 1][2] for s in steps) / 1000 if i else 0.0
    tot_k += dur
    tot_g += gap
    print(f"{i:3d} {name[:40].ljust(42)} {dur:8.2f} us   gap before {gap:6.2f}")
print(f"kernels {tot_k:.1f} us, gaps {tot_g:.1f} us")
