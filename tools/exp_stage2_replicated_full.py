#!/usr/bin/env python3
"""Stage 2 at the FULL shapes of an 8-rank step, on one card: what a rank verifies after the exchange when every rank holds 8192 x 64 blobs.
  replicated all-gather   8192 batches of 512 records (every rank verifies every batch)
  split forms             1024 batches of 512 records (all-to-all / split all-gather: 1 / 8 of the batches)
and the same at 2 and 4 ranks.  Records of valid 64-blob batches concatenate to a valid batch.  Every verdict must be true.
usage: exp_stage2_replicated_full.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
n = 64
base = torch.frombuffer(bytearray(b"".join(random_blob(9100 + i) for i in range(n))), dtype=torch.uint8).to(dev)
out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), n, s.handle) == 0
tc = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc.data_ptr(), n, s.handle) == 0
tp = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
rec1 = torch.empty(n * 160, dtype=torch.uint8, device=dev); st1 = (C.c_int * 1)()
assert L.kzg355_verify_shard_records_device(rec1.data_ptr(), st1, base.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, 1, s.handle) == 0
for world in (2, 4, 8):
    for form, groups in (("split forms", 8192 // world), ("replicated all-gather", 8192)):
        rec = rec1.repeat(world * groups)
        ok = (C.c_bool * groups)(); st2 = (C.c_int * groups)()
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc = L.kzg355_verify_records_device(ok, st2, rec.data_ptr(), n * world, groups, s.handle)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0, rc
        assert all(ok[i] for i in range(groups)) and not any(st2[i] for i in range(groups))
        free, total = torch.cuda.mem_get_info(dev)
        print(f"world {world}, {form}: {groups} batches of {n * world} records: {min(ts):.1f} ms, every verdict true; HBM in use {(total - free) / 1e9:.1f} GB", flush=True)
        del rec
s.free()
