#!/usr/bin/env python3
"""Crossover of the device-resident host-hash route (round 5): one synchronous kzg355_verify_blob_kzg_proof_batch_many_device call over G batches of 64 blobs already
in HBM, with the blobs copied back and hashed on the host threads (host_hash_device_max_blobs large) against the device's Fiat-Shamir kernels (set_host_hash(-1)).
Median ms per call over 12 calls after 3 warm-ups.  usage: exp_device_host_hash_crossover.py"""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.KzgSettings.load_trusted_setup_ex([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)], msm_bits=12, host_hash_device_max_blobs=1 << 20)
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
n = 64
base = torch.frombuffer(bytearray(b"".join(random_blob(9000 + i) for i in range(n))), dtype=torch.uint8).to(dev)
out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), n, s.handle) == 0
tc1 = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc1.data_ptr(), n, s.handle) == 0
tp1 = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
print(f"{'batches':>8s} {'blobs':>6s} {'host route ms':>14s} {'device hash ms':>15s}")
for G in (1, 2, 4, 8, 12, 16, 24, 32, 64):
    blobs, tc, tp = base.repeat(G), tc1.repeat(G), tp1.repeat(G)
    torch.cuda.synchronize()
    ok = (C.c_bool * G)(); stg = (C.c_int * G)()
    res = []
    for mode in (0, -1):
        s.set_host_hash(mode)
        ts = []
        for i in range(15):
            t0 = time.perf_counter()
            rc = L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, blobs.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, s.handle)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0 and all(ok[k] for k in range(G))
        res.append(statistics.median(ts[3:]))
    print(f"{G:8d} {G * n:6d} {res[0]:14.3f} {res[1]:15.3f}")
s.free()
