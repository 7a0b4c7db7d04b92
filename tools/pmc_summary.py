#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes over the same bench.py command into profiles/rNN/pmc_traffic_*.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write out.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count
KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so it is doubled; the counters come from separate passes.
Every kernel is normalised by the blobs of ITS OWN largest launch, derived from the launch's Grid_Size (tools/launch_shapes.py): the
verify kernels of the bench see 524,288 blobs per launch, the commit / proof kernels of its untimed setup 65,536."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from launch_shapes import blobs_of_launch


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {directory}"
    rows = [r for f in files for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    by_kernel = defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
        by_kernel[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for name, v in by_kernel.items():
        gmax = max(g for g, _ in v)
        big = [c for g, c in v if g == gmax]
        out[name] = (sum(big) / len(big), len(big), gmax)
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1], sys.argv[2], sys.argv[-1]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, nf, grid = fetch.get(k, (0.0, 0, 0))
        w, _, gw = write.get(k, (0.0, 0, 0))
        blobs = blobs_of_launch(k, grid or gw)
        if not blobs:
            continue
        res[k] = {"fetch_bytes_per_blob_x2_corrected": round(2 * f * 1024 / blobs), "write_bytes_per_blob": round(w * 1024 / blobs), "launches": nf,
                  "grid_size": grid or gw, "blobs_per_launch": round(blobs)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py; per kernel the largest-grid launches only, normalised by "
                       "the blobs of that launch (Grid_Size -> blobs: tools/launch_shapes.py); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                       "1/2 for wide coalesced reads); KiB = 1024 B; kernels that run once per batch are divided by the blobs of their batches",
               "per_kernel": res}, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
