#!/usr/bin/env python3
"""Per-kernel HIP-event times of ONE 64-blob batch (device-resident inputs): where the ~7.6 ms of single-call latency go."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
n = int(os.environ.get("N", "64"))
base = torch.frombuffer(bytearray(b"".join(random_blob(9000 + i) for i in range(n))), dtype=torch.uint8).to(dev)
out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), n, s.handle) == 0
tc = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc.data_ptr(), n, s.handle) == 0
tp = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
ok = (C.c_bool * 1)(); sg = (C.c_int * 1)()
def call():
    assert L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, sg, base.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, 1, s.handle) == 0 and ok[0]
for _ in range(3): call()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"n = {n}: wall per call median {sorted(ts)[5]:.3f} ms, min {min(ts):.3f} ms")
L.kzg355_reset_kernel_stats(s.handle); s.set_kernel_timing(True)
for _ in range(5): call()
s.set_kernel_timing(False)
tot_all = 0.0
for fam in ("validate_points", "lincomb_shift", "challenge", "eval", "rpowers", "lincomb_prep", "lincomb", "lincomb_horner", "pairing"):
    tot, cnt = C.c_double(), C.c_long()
    L.kzg355_kernel_ms_stats(s.handle, fam.encode(), C.byref(tot), C.byref(cnt))
    if cnt.value:
        print(f"  {fam:16s} {tot.value / cnt.value:.3f} ms")
        if fam not in ("validate_points", "lincomb_shift"): tot_all += tot.value / cnt.value
print(f"  main-stream chain (challenge .. pairing): {tot_all:.3f} ms; validate / shift run on the side stream")
s.free()
