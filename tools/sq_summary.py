#!/usr/bin/env python3
"""Turn one rocprofv3 SQ counter pass over a bench.py command into profiles/rNN/sq[_tag]_vK.json.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \\
              --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py ...
    python tools/sq_summary.py gpurun_out/pmc_sq BLOBS_PER_LAUNCH out.json

Per kernel (largest-grid launches only, i.e. the launches that processed BLOBS_PER_LAUNCH blobs): wave-instructions per blob
(SQ_INSTS_VALU is counted per wave-instruction), waves per launch, and the share of the wave-cycles spent waiting to issue
(SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) or parked in s_waitcnt / barriers (SQ_WAIT_ANY / SQ_WAVE_CYCLES).  bench.py's
roofline.alu combines the per-blob instruction counts with its live HIP-event kernel durations."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    directory, blobs, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
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
        for cname, vals in counters.items():
            gmax = max(g for g, _ in vals)
            big = [c for g, c in vals if g == gmax]
            v[cname] = sum(big) / len(big)
            v["_launches"] = len(big)
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        res[name] = {"valu_wave_insts_per_blob": round(v.get("SQ_INSTS_VALU", 0.0) / blobs, 2),
                     "salu_insts_per_blob": round(v.get("SQ_INSTS_SALU", 0.0) / blobs, 2),
                     "lds_insts_per_blob": round(v.get("SQ_INSTS_LDS", 0.0) / blobs, 2),
                     "waves_per_launch": round(v.get("SQ_WAVES", 0.0)),
                     "wait_inst_any_frac": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4) if wc else None,
                     "wait_any_frac": round(v.get("SQ_WAIT_ANY", 0.0) / wc, 4) if wc else None,
                     "launches": v.get("_launches", 0)}
    json.dump({"note": f"rocprofv3 --pmc SQ_* over bench.py, {blobs} blobs per launch (largest-grid launches only); SQ_INSTS_VALU counts "
                       "wave-instructions (summed over the chip); SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md), only "
                       "their ratio is used",
               "blobs_per_launch": blobs, "per_kernel": res}, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
