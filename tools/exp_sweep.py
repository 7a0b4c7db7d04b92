#!/usr/bin/env python3
"""Sweeps odd launch sizes through the device entry points (honest inputs must verify; commitments must be stable across
call shapes): a crash / hang / false verdict hunt, not a benchmark."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
NB = 1600
gen = torch.Generator(device=dev); gen.manual_seed(99)
tb = torch.randint(0, 256, (NB, 4096, 32), dtype=torch.uint8, device=dev, generator=gen); tb[:, :, 0] = 0; tb = tb.reshape(-1).contiguous()
out = C.create_string_buffer(48 * NB); st = (C.c_int * NB)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, tb.data_ptr(), NB, s.handle) == 0
cs = out.raw; tc = torch.frombuffer(bytearray(cs), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, tb.data_ptr(), tc.data_ptr(), NB, s.handle) == 0
ps = out.raw; tp = torch.frombuffer(bytearray(ps), dtype=torch.uint8).to(dev)
t0 = time.time()
for n in (1, 2, 15, 16, 17, 127, 128, 129, 1023, 1024, 1025, 1600):
    o2 = C.create_string_buffer(48 * n)
    assert L.kzg355_blob_to_kzg_commitment_many_device(o2, st, tb.data_ptr(), n, s.handle) == 0 and o2.raw == cs[:48 * n], n
    assert L.kzg355_compute_blob_kzg_proof_many_device(o2, st, tb.data_ptr(), tc.data_ptr(), n, s.handle) == 0 and o2.raw == ps[:48 * n], n
print("commit / proof shapes ok")
for npg, G in ((1, 1), (1, 70), (2, 33), (3, 65), (7, 200), (8, 64), (9, 63), (31, 51), (64, 1), (64, 2), (64, 3), (64, 25), (65, 24), (100, 16), (127, 12), (128, 12), (129, 12), (200, 8), (512, 3), (1600, 1)):
    ok = (C.c_bool * G)(); stg = (C.c_int * G)()
    rc = L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), npg, G, s.handle)
    assert rc == 0 and all(ok[i] for i in range(G)) and not any(stg[i] for i in range(G)), (npg, G, rc)
    # corrupt one proof of the last batch -> only that batch turns false
    bad = tp.clone(); j = (npg * G - 1) * 48; k = (npg * (G - 1)) * 48
    if npg > 1:
        tmp = bad[j:j + 48].clone(); bad[j:j + 48] = bad[k:k + 48]; bad[k:k + 48] = tmp
        rc = L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, tb.data_ptr(), tc.data_ptr(), bad.data_ptr(), npg, G, s.handle)
        assert rc == 0 and [ok[i] for i in range(G)] == [True] * (G - 1) + [False], (npg, G)
print(f"verify shapes ok ({time.time() - t0:.1f} s)")
s.free()
