#!/usr/bin/env python3
"""Does keeping two launch sets in flight (two host threads, each on its own workspace / stream) beat one at a time?"""
import ctypes as C, os, sys, time, threading
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
tc1 = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, base.data_ptr(), tc1.data_ptr(), n, s.handle) == 0
tp1 = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
def run(G, threads, steps):
    bufs = []
    for _ in range(threads):
        bufs.append((base.repeat(G), tc1.repeat(G), tp1.repeat(G), (C.c_bool * G)(), (C.c_int * G)()))
    torch.cuda.synchronize()
    def worker(k):
        b, c, p, ok, stg = bufs[k]
        for _ in range(steps):
            assert L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, b.data_ptr(), c.data_ptr(), p.data_ptr(), n, G, s.handle) == 0
            assert all(ok[i] for i in range(G))
    for k in range(threads): worker(k) if steps == 0 else None
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    return threads * steps * G * n / dt
CASES = ((2048, 1), (2048, 2), (2048, 3), (256, 1), (256, 2), (256, 4), (64, 1), (64, 2), (64, 4), (1, 1), (1, 2), (1, 4))
if len(sys.argv) > 1:
    CASES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for G, threads in CASES:
    run(G, threads, 1)
    print(f"G={G} x {threads} thread(s) in flight: {run(G, threads, 6) / 1e6:.3f} M blobs/s")
s.free()
