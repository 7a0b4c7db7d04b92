#!/usr/bin/env python3
"""HIP-event time of every kernel family of one device-resident verify launch set, verdicts NOT checked -- for diagnostic builds that change what a kernel
computes in order to see what a part of it costs (e.g. k_lc_buckets with every lane gathering the same few points).  usage: exp_kernel_times.py [batches] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ.setdefault("KZG355_SELFTEST", "0")
import torch
import kzg_rust_amd as kz
from synth import random_blob
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 64
g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.KzgSettings.load_trusted_setup_ex([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)], msm_bits=12, self_test=0)
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
base = torch.frombuffer(bytearray(b"".join(random_blob(9000 + i) for i in range(n))), dtype=torch.uint8).to(dev)
out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), n, s.handle) == 0
tc1 = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc1.data_ptr(), n, s.handle) == 0
blobs, tc, tp = base.repeat(G), tc1.repeat(G), torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev).repeat(G)
torch.cuda.synchronize()
ok = (C.c_bool * G)(); stg = (C.c_int * G)()
L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, blobs.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, s.handle)
L.kzg355_reset_kernel_stats(s.handle); s.set_kernel_timing(True)
for _ in range(reps):
    L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, blobs.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, s.handle)
s.set_kernel_timing(False)
print(f"{G} batches of {n}; verdicts true: {sum(1 for i in range(G) if ok[i])} of {G}")
for fam in ("validate_points", "challenge", "eval", "rpowers", "lincomb_prep", "lincomb", "lincomb_horner", "pairing"):
    tot, cnt = C.c_double(), C.c_long()
    L.kzg355_kernel_ms_stats(s.handle, fam.encode(), C.byref(tot), C.byref(cnt))
    if cnt.value:
        print(f"  {fam:18s} {tot.value / cnt.value:8.3f} ms")
s.free()
