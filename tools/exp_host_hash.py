#!/usr/bin/env python3
"""Host-buffer verify calls with the Fiat-Shamir challenges hashed on the host (kzg_rust_amd/csrc/host_sha256.cpp) against the
device hash: wall time per call over a ladder of call sizes (the crossover that KZG355_HOST_HASH_MAX encodes), plus the per-kernel
HIP-event times of the reference-shaped call (one verify_blob_kzg_proof_batch of 64 blobs on host slices,
benches/kzg_benches.rs:113-120) both ways.  Output goes to profiles/r03/host_hash_crossover.txt."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import kzg_rust_amd as kz
from synth import random_blob

g = os.path.join(ROOT, "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
NB = int(os.environ.get("NB", "128"))          # distinct blobs; larger calls repeat them
blobs = b"".join(random_blob(9000 + i) for i in range(NB))
base = torch.frombuffer(bytearray(blobs), dtype=torch.uint8).to(dev)
out = C.create_string_buffer(48 * NB); st = (C.c_int * NB)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, base.data_ptr(), NB, s.handle) == 0
cs = out.raw
tc = torch.frombuffer(bytearray(cs), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc.data_ptr(), NB, s.handle) == 0
ps = out.raw
print(f"host threads: KZG355_HOST_THREADS={os.environ.get('KZG355_HOST_THREADS', 'default')}, cpus available {len(os.sched_getaffinity(0))}")


def timed(npg, groups, mode, reps=9):
    n = npg * groups
    rep = (n + NB - 1) // NB
    hb = np.frombuffer((blobs * rep)[:131072 * n], dtype=np.uint8).copy()
    hc, hp = (cs * rep)[:48 * n], (ps * rep)[:48 * n]
    ok = (C.c_bool * groups)(); sg = (C.c_int * groups)()
    s.set_host_hash(mode)
    ts = []
    for i in range(reps + 2):
        t0 = time.perf_counter()
        rc = L.kzg355_verify_blob_kzg_proof_batch_many(ok, sg, hb.ctypes.data_as(C.c_char_p), hc, hp, npg, groups, s.handle)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0 and all(ok[i] for i in range(groups)), (rc, list(ok))
    s.set_host_hash(0)
    ts = ts[2:]
    return statistics.median(ts), min(ts)


print("call shape            blobs   host hash (median / min ms)   device hash (median / min ms)")
for npg, groups in ((1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (32, 1), (64, 1), (128, 1), (64, 4), (64, 8), (64, 16), (64, 32), (64, 48), (64, 64), (64, 128)):
    h = timed(npg, groups, 1); d = timed(npg, groups, -1)
    print(f"{groups:3d} x {npg:3d}            {npg * groups:6d}   {h[0]:8.3f} / {h[1]:8.3f}            {d[0]:8.3f} / {d[1]:8.3f}")

fams = ("decompress_points", "validate_points", "lincomb_shift", "challenge", "challenge_from_digest", "eval", "rpowers", "lincomb_prep", "lincomb", "lincomb_horner", "pairing")
hb = np.frombuffer(blobs[:131072 * 64], dtype=np.uint8).copy()
ok1 = C.c_bool()
for mode, name in ((1, "host hash"), (-1, "device hash")):
    s.set_host_hash(mode)
    L.kzg355_reset_kernel_stats(s.handle); s.set_kernel_timing(True)
    for _ in range(5):
        assert L.kzg355_verify_blob_kzg_proof_batch(C.byref(ok1), hb.ctypes.data_as(C.c_char_p), 64, cs[:48 * 64], 64, ps[:48 * 64], 64, s.handle) == 0 and ok1.value
    s.set_kernel_timing(False)
    print(f"one verify_blob_kzg_proof_batch(n = 64) on host slices, {name}: HIP-event time per kernel family")
    for fam in fams:
        tot, cnt = C.c_double(), C.c_long()
        L.kzg355_kernel_ms_stats(s.handle, fam.encode(), C.byref(tot), C.byref(cnt))
        if cnt.value:
            print(f"  {fam:22s} {tot.value / cnt.value:.3f} ms")
s.set_host_hash(0)
s.free()
