#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> the device timeline of the last part of a run: per kernel name its busy time, and how much of the span had 0 / 1 / 2+ kernels
running (union over all queues).  usage: trace_timeline.py <dir> [tail_ms]   (tail_ms: how much of the end of the trace to look at; default 200)"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
t_end = max(r[1] for r in rows); t0 = t_end - int(tail_ms * 1e6)
rows = [r for r in rows if r[0] >= t0 and r[2].startswith("k_")]
span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
busy = defaultdict(float); cnt = defaultdict(int)
for s, e, n in rows: busy[n] += (e - s) / 1e6; cnt[n] += 1
ev = sorted([(s, 1) for s, e, n in rows] + [(e, -1) for s, e, n in rows])
depth = 0; last = ev[0][0]; at = defaultdict(float)
for t, dlt in ev:
    at[min(depth, 3)] += (t - last) / 1e6; last = t; depth += dlt
print(f"span {span:.1f} ms, kernels {len(rows)}; time with 0 / 1 / 2 / 3+ kernels running: " + " / ".join(f"{at[k]:.1f}" for k in range(4)) + " ms")
for n in sorted(busy, key=lambda k: -busy[k]): print(f"  {n:28s} {cnt[n]:4d} launches  {busy[n]:8.2f} ms  avg {busy[n] / cnt[n]:.3f}")
