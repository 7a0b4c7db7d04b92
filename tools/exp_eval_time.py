#!/usr/bin/env python3
"""Stage-1 kernels alone (challenge, evaluation) on N device-resident blobs, HIP-event time per kernel family: the harness for
k_eval experiments (no verdicts involved, so diagnostic builds that skip part of the evaluation can be timed).  N from argv."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
g = os.path.join(ROOT, "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
NB = 64
blobs = b"".join(random_blob(7000 + i) for i in range(NB))
base = torch.frombuffer(bytearray(blobs), dtype=torch.uint8).to(dev)
out = C.create_string_buffer(48 * NB); st = (C.c_int * NB)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), NB, s.handle) == 0
cs = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), cs.data_ptr(), NB, s.handle) == 0
ps = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
rep = n // NB
tb, tc, tp = base.repeat(rep), cs.repeat(rep), ps.repeat(rep)
rec = torch.zeros(160 * n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
stg = (C.c_int * (n // 64))()
L.kzg355_reset_kernel_stats(s.handle); s.set_kernel_timing(True)
for it in range(4):
    if it == 1: L.kzg355_reset_kernel_stats(s.handle)
    assert L.kzg355_verify_shard_records_device(rec.data_ptr(), stg, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), 64, n // 64, s.handle) == 0
s.set_kernel_timing(False)
res = {}
for fam in ("challenge", "eval", "validate_points"):
    tot, cnt = C.c_double(), C.c_long()
    L.kzg355_kernel_ms_stats(s.handle, fam.encode(), C.byref(tot), C.byref(cnt))
    if cnt.value: res[fam] = round(tot.value / cnt.value, 3)
print(os.environ.get("TAG", ""), n, res)
s.free()
