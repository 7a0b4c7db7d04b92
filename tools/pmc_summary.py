#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes over the same bench.py command into profiles/rNN/pmc_traffic_*.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write BLOBS_PER_LAUNCH out.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count
KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so it is doubled; the counters come from separate passes.
Only the launches that processed BLOBS_PER_LAUNCH blobs are summed (grid size filter per kernel: the largest grid seen)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


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
        out[name] = (sum(big) / len(big), len(big))
    return out


def main():
    fetch_dir, write_dir, blobs, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, nf = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        res[k] = {"fetch_bytes_per_blob_x2_corrected": round(2 * f * 1024 / blobs), "write_bytes_per_blob": round(w * 1024 / blobs), "launches": nf}
    json.dump({"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py, {blobs} blobs per launch (largest-grid launches "
                       "only); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 for wide coalesced reads); KiB = 1024 B; "
                       "kernels that run once per batch rather than per blob are still divided by the blob count",
               "blobs_per_launch": blobs, "per_kernel": res}, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
