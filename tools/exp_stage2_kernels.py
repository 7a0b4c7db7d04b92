#!/usr/bin/env python3
"""Stage 2 alone on G (default 8192) batches of 64 records made from ONE proved batch: the library's own HIP-event time per kernel family, best of REPS runs.
The harness for A/B builds of the stage-2 kernels (KZG355_LIBRARY selects the build).  usage: exp_stage2_kernels.py [G]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
G, n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 64
base = torch.frombuffer(bytearray(b"".join(random_blob(9000 + i) for i in range(n))), dtype=torch.uint8).to(dev)
out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), n, s.handle) == 0
tc = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc.data_ptr(), n, s.handle) == 0
tp = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
rec1 = torch.empty(n * 160, dtype=torch.uint8, device=dev); st1 = (C.c_int * 1)()
assert L.kzg355_verify_shard_records_device(rec1.data_ptr(), st1, base.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, 1, s.handle) == 0
rec = rec1.repeat(G)
# one batch in eight gets a wrong y: its verdict must be false, the others true
bad = torch.arange(0, G, 8, device=dev)
rec.view(G, n * 160)[bad, 80 + 31] ^= 1
ok = (C.c_bool * G)(); st2 = (C.c_int * G)()
best = {}
for rep in range(int(os.environ.get("REPS", "5")) + 1):
    L.kzg355_reset_kernel_stats(s.handle); s.set_kernel_timing(True)
    assert L.kzg355_verify_records_device(ok, st2, rec.data_ptr(), n, G, s.handle) == 0
    s.set_kernel_timing(False)
    if rep == 0:
        continue
    for fam in ("points_from_records", "rpowers", "lincomb_prep", "lincomb", "lincomb_horner", "pairing"):
        tot, cnt = C.c_double(), C.c_long()
        L.kzg355_kernel_ms_stats(s.handle, fam.encode(), C.byref(tot), C.byref(cnt))
        if cnt.value:
            best[fam] = min(best.get(fam, 1e9), tot.value / cnt.value)
if not os.environ.get("NOCHECK"):                                          # (diagnostic builds compute something else)
    assert all(bool(ok[i]) == (i % 8 != 0) for i in range(G)) and not any(st2[i] for i in range(G))
print(os.environ.get("TAG", ""), G, "batches:", ", ".join(f"{k} {v:.3f}" for k, v in best.items()), "ms; verdicts as expected", flush=True)
s.free()
