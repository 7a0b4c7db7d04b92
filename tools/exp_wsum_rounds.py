#!/usr/bin/env python3
"""k_lc_wsum's grid quantisation, measured.  The kernel holds 57 KB of LDS per 256-thread workgroup: two workgroups per CU, 512 per round
of the card; 8192 batches are 52 * 8192 / 256 = 1664 workgroups = 3.25 rounds.  Stage 2 is run on record sets whose batch counts give 2, 2.25, 2.5, 3, 3.25, 3.5
and 4 rounds; under `rocprofv3 --kernel-trace` the trace holds one k_lc_wsum launch per count (told apart by grid size), and with KERNELS=1 the library's own
event timing of the window-sum + chain family is printed.
usage: rocprofv3 --kernel-trace -d gpurun_out/wsum -o wsum --output-format csv -- python3 tools/exp_wsum_rounds.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
n = 64
base = torch.frombuffer(bytearray(b"".join(random_blob(9000 + i) for i in range(n))), dtype=torch.uint8).to(dev)
out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), n, s.handle) == 0
tc = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc.data_ptr(), n, s.handle) == 0
tp = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
rec1 = torch.empty(n * 160, dtype=torch.uint8, device=dev); st1 = (C.c_int * 1)()
assert L.kzg355_verify_shard_records_device(rec1.data_ptr(), st1, base.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, 1, s.handle) == 0
COUNTS = [(5041, 2.0), (5671, 2.25), (6301, 2.5), (7561, 3.0), (8192, 3.25), (8822, 3.5), (10082, 4.0)]
rec = rec1.repeat(COUNTS[-1][0])
for G, rounds in COUNTS:
    ok = (C.c_bool * G)(); st2 = (C.c_int * G)()
    for rep in range(2):                                                     # (the second launch of each count is the one to read)
        assert L.kzg355_verify_records_device(ok, st2, rec.data_ptr(), n, G, s.handle) == 0
    assert all(ok[i] for i in range(G))
    wgs = (52 * G + 255) // 256
    line = f"{G} batches: k_lc_wsum grid {wgs} workgroups = {wgs / 512:.3f} rounds"
    if os.environ.get("KERNELS"):
        L.kzg355_reset_kernel_stats(s.handle); s.set_kernel_timing(True)
        L.kzg355_verify_records_device(ok, st2, rec.data_ptr(), n, G, s.handle); s.set_kernel_timing(False)
        tot, cnt = C.c_double(), C.c_long()
        L.kzg355_kernel_ms_stats(s.handle, b"lincomb_horner", C.byref(tot), C.byref(cnt))
        line += f"; window sums + chains {tot.value / max(cnt.value, 1):.3f} ms"
    print(line, flush=True)
s.free()
