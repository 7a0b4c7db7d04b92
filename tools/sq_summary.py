#!/usr/bin/env python3
"""Turn one rocprofv3 SQ counter pass over a bench.py command into profiles/rNN/sq[_tag]_vK.json.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \\
              --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py ...
    python tools/sq_summary.py gpurun_out/pmc_sq out.json

Per kernel (largest-grid launches only, normalised by the blobs of THAT launch -- Grid_Size -> blobs: tools/launch_shapes.py; the
verify kernels of the bench see 524,288 blobs per launch, the commit / proof kernels of its untimed setup 65,536): wave-instructions per blob
(SQ_INSTS_VALU is counted per wave-instruction), waves per launch, and the share of the wave-cycles spent waiting to issue
(SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) or parked in s_waitcnt / barriers (SQ_WAIT_ANY / SQ_WAVE_CYCLES).  bench.py's
roofline.alu combines the per-blob instruction counts with its live HIP-event kernel durations."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from launch_shapes import blobs_of_launch


def main():
    directory, dst = sys.argv[1], sys.argv[-1]
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {directory}"
    agg = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
            if name.startswith("k_"):
                agg[name][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    res = {}
    for name, counters in sorted(agg.items()):
        v = {}
        gmax = 0
        for cname, vals in counters.items():
            gmax = max(g for g, _ in vals)
            big = [c for g, c in vals if g == gmax]
            v[cname] = sum(big) / len(big)
            v["_launches"] = len(big)
        blobs = blobs_of_launch(name, gmax)
        if not blobs:
            continue
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        res[name] = {"valu_wave_insts_per_blob": round(v.get("SQ_INSTS_VALU", 0.0) / blobs, 2),
                     "salu_insts_per_blob": round(v.get("SQ_INSTS_SALU", 0.0) / blobs, 2),
                     "lds_insts_per_blob": round(v.get("SQ_INSTS_LDS", 0.0) / blobs, 2),
                     "waves_per_launch": round(v.get("SQ_WAVES", 0.0)),
                     "wait_inst_any_frac": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4) if wc else None,
                     "wait_any_frac": round(v.get("SQ_WAIT_ANY", 0.0) / wc, 4) if wc else None,
                     "launches": v.get("_launches", 0), "grid_size": gmax, "blobs_per_launch": round(blobs)}
    json.dump({"note": "rocprofv3 --pmc SQ_* over bench.py; per kernel the largest-grid launches only, normalised by the blobs of that launch; SQ_INSTS_VALU counts "
                       "wave-instructions (summed over the chip); SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md), only "
                       "their ratio is used",
               "per_kernel": res}, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
