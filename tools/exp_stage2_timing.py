#!/usr/bin/env python3
"""Times the two stages of the sharded path on one GPU for the shapes an 8-GPU run gives each rank:
stage 1 on 2048 x 64 blobs, stage 2 on 2048 / world batches of 64 x world records (records of valid batches concatenate
to valid batches)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
G, n = int(os.environ.get("G", "2048")), 64
base = torch.frombuffer(bytearray(b"".join(random_blob(9000 + i) for i in range(n))), dtype=torch.uint8).to(dev)
blobs = base.repeat(G)
out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), n, s.handle) == 0
cs = out.raw
tc1 = torch.frombuffer(bytearray(cs), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc1.data_ptr(), n, s.handle) == 0
tc = tc1.repeat(G); tp = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev).repeat(G)
rec = torch.empty(G * n * 160, dtype=torch.uint8, device=dev)
stg = (C.c_int * G)()
def t(f, reps=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
ms1 = t(lambda: L.kzg355_verify_shard_records_device(rec.data_ptr(), stg, blobs.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, s.handle))
print(f"stage 1: {G} x {n} blobs: {ms1:.1f} ms")
for world in (1, 2, 4, 8):
    gm = G // world
    ok = (C.c_bool * gm)(); st2 = (C.c_int * gm)()
    ms2 = t(lambda: L.kzg355_verify_records_device(ok, st2, rec.data_ptr(), n * world, gm, s.handle))
    assert all(ok[i] for i in range(gm))
    if os.environ.get("KERNELS"):
        L.kzg355_reset_kernel_stats(s.handle); s.set_kernel_timing(True)
        L.kzg355_verify_records_device(ok, st2, rec.data_ptr(), n * world, gm, s.handle); s.set_kernel_timing(False)
        parts = []
        for fam in ("points_from_records", "rpowers", "lincomb_prep", "lincomb", "lincomb_horner", "pairing"):
            tot, cnt = C.c_double(), C.c_long()
            L.kzg355_kernel_ms_stats(s.handle, fam.encode(), C.byref(tot), C.byref(cnt))
            if cnt.value: parts.append(f"{fam} {tot.value / cnt.value:.2f}")
        print("   stage 2 kernels (ms):", ", ".join(parts))
    print(f"stage 2 as at world={world}: {gm} batches of {n * world}: {ms2:.1f} ms   -> est. step {ms1 + ms2:.1f} ms, {G * n * world / (ms1 + ms2) * 1e3 / 1e6:.2f} M blobs/s aggregate (without the all-gather)")
s.free()
