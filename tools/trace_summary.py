#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per kernel: the launches at its LARGEST grid and their average duration.

    python tools/trace_summary.py <dir with *_kernel_trace.csv> out.csv

rocprofv3's own --stats averages every launch of a kernel, whatever its size: in a bench run the untimed setup launches the same kernels on
smaller slices (the proofs of the step's blobs are made 65,536 at a time), so the stats average of e.g. k_challenge_1w mixes 5 full-size launches
with 8 small ones.  This summary keeps the full-size launches only -- the figure bench.py's live HIP-event average must agree with."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    directory, dst = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no kernel_trace.csv under {directory}"
    per = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
            if not name.startswith("k_"):
                continue
            grid = int(r.get("Grid_Size") or int(r.get("Grid_Size_X", 0)) * max(1, int(r.get("Grid_Size_Y", 1))) * max(1, int(r.get("Grid_Size_Z", 1))))
            per[name].append((grid, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    with open(dst, "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["kernel", "largest_grid_work_items", "launches_at_largest_grid", "avg_ms", "min_ms", "max_ms", "launches_total"])
        for name in sorted(per, key=lambda n: -sum(d for g, d in per[n])):
            gmax = max(g for g, _ in per[name])
            ds = [d for g, d in per[name] if g == gmax]
            w.writerow([name, gmax, len(ds), round(sum(ds) / len(ds), 4), round(min(ds), 4), round(max(ds), 4), len(per[name])])


if __name__ == "__main__":
    main()
