#!/usr/bin/env python3
"""Per-kernel timings of the single-op entry points (verify_kzg_proof, verify_blob_kzg_proof, compute_blob_kzg_proof)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(ROOT, "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
b = kz.Blob(random_blob(1)); c = kz.Kzg.blob_to_kzg_commitment(b, s); p = kz.Kzg.compute_blob_kzg_proof(b, c, s)
z = kz.Bytes32(bytes(31) + b"\x05"); pr, y = kz.Kzg.compute_kzg_proof(b, z, s)
fams = ["validate_points", "challenge", "eval", "rpowers", "lincomb_shift", "lincomb_prep", "lincomb", "lincomb_horner", "pairing", "quotient", "msm_wide", "msm_finalize"]
ops = {"verify_kzg_proof": lambda: kz.Kzg.verify_kzg_proof(c, z, y, pr, s), "verify_blob_kzg_proof": lambda: kz.Kzg.verify_blob_kzg_proof(b, c, p, s),
       "compute_blob_kzg_proof": lambda: kz.Kzg.compute_blob_kzg_proof(b, c, s), "compute_kzg_proof": lambda: kz.Kzg.compute_kzg_proof(b, z, s)}
for name, fn in ops.items():
    for _ in range(3): fn()
    s.set_kernel_timing(True)
    t0 = time.perf_counter(); fn(); dt = (time.perf_counter() - t0) * 1e3
    s.set_kernel_timing(False)
    print(name, f"{dt:.2f} ms:", {f: round(s.last_kernel_ms(f), 3) for f in fams if s.last_kernel_ms(f) >= 0})
    kz.kzg.lib().kzg355_reset_kernel_stats(s.handle)
s.free()
