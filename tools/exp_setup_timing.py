#!/usr/bin/env python3
"""Times load_trusted_setup (with and without the wide-window MSM table) and single-call latencies on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
G1 = [g1[48 * i:48 * i + 48] for i in range(4096)]; G2 = [g2[96 * i:96 * i + 96] for i in range(65)]
for mode in ("bucket", "wide", "wide"):
    os.environ["KZG355_MSM"] = mode
    t0 = time.perf_counter(); s = kz.Kzg.load_trusted_setup(G1, G2); t1 = time.perf_counter()
    b = kz.Blob(random_blob(1))
    c = kz.Kzg.blob_to_kzg_commitment(b, s)
    ts = []
    for _ in range(5):
        t2 = time.perf_counter(); c = kz.Kzg.blob_to_kzg_commitment(b, s); ts.append(time.perf_counter() - t2)
    p = kz.Kzg.compute_blob_kzg_proof(b, c, s)
    tp = []
    for _ in range(5):
        t2 = time.perf_counter(); p = kz.Kzg.compute_blob_kzg_proof(b, c, s); tp.append(time.perf_counter() - t2)
    tv = []
    for _ in range(5):
        t2 = time.perf_counter(); ok = kz.Kzg.verify_blob_kzg_proof(b, c, p, s); tv.append(time.perf_counter() - t2)
    print(f"KZG355_MSM={mode}: load_trusted_setup {1e3 * (t1 - t0):.0f} ms; one blob (host buffers): commit {1e3 * min(ts):.2f} ms, "
          f"blob proof {1e3 * min(tp):.2f} ms, verify_blob_kzg_proof {1e3 * min(tv):.2f} ms ({ok}); free HBM {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB")
    s.free()
