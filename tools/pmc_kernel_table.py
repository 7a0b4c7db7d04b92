#!/usr/bin/env python3
"""Per kernel (largest-grid launches only) the sum of every counter of one or more `rocprofv3 --pmc` passes, and each as a fraction of SQ_WAVE_CYCLES / SQ_BUSY_CYCLES
when those were collected in the same pass.  usage: pmc_kernel_table.py <dir> [<dir> ...]"""
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:]:
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    grid = defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
        grid[k] = max(grid[k], int(r["Grid_Size"]))
    tot = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
        if int(r["Grid_Size"]) != grid[k]:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    print(f"== {d}")
    for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", 0))):
        base = c.get("SQ_WAVE_CYCLES") or c.get("SQ_BUSY_CYCLES") or 1.0
        print(f"{k} (grid {grid[k]}, {len(disp[k])} launches)")
        for name, v in sorted(c.items()):
            print(f"    {name:32s} {v / len(disp[k]):16.0f}   {v / base:8.4f} of {'SQ_WAVE_CYCLES' if c.get('SQ_WAVE_CYCLES') else 'SQ_BUSY_CYCLES'}")
