#!/usr/bin/env python3
"""Reads the kernel trace of tools/exp_wsum_rounds.py: the LAST k_lc_wsum / k_lc_hchain_quad / k_lc_buckets launch of every grid size, in ms.
usage: wsum_rounds_summary.py <kernel_trace.csv>"""
import csv, sys
last = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
    if name not in ("k_lc_wsum", "k_lc_hchain_quad", "k_lc_buckets"):
        continue
    grid = int(r.get("Grid_Size") or int(r.get("Grid_Size_X", 0)))
    wg = int(r.get("Workgroup_Size") or int(r.get("Workgroup_Size_X", 256)))
    last[(name, grid // wg)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for name in ("k_lc_wsum", "k_lc_hchain_quad", "k_lc_buckets"):
    print(name)
    for (nm, wgs), ms in sorted(last.items()):
        if nm == name and wgs > 256:
            extra = f" = {wgs / 512:.3f} rounds of 512, {ms / (wgs / 512):.3f} ms per round-equivalent" if name == "k_lc_wsum" else ""
            print(f"  {wgs:6d} workgroups: {ms:7.3f} ms{extra}")
