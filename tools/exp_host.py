#!/usr/bin/env python3
"""Host-buffer streaming rate of kzg355_verify_blob_kzg_proof_batch_many for several pipeline shapes (chunk size, chunks in
flight, copy threads; each read from the environment when a handle is created).  usage: exp_host.py [n_batches]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kzg_rust_amd as kz
G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
golden = os.path.join(ROOT, "tests", "golden")
g1 = open(os.path.join(golden, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(golden, "trusted_setup_g2.bin"), "rb").read()
g1l = [g1[48 * i:48 * i + 48] for i in range(4096)]; g2l = [g2[96 * i:96 * i + 96] for i in range(65)]
L = kz.kzg.lib()
dev = torch.device("cuda", 0)
n = 64 * G
gen = torch.Generator(device=dev); gen.manual_seed(1)
tb = torch.randint(0, 256, (n, 4096, 32), dtype=torch.uint8, device=dev, generator=gen); tb[:, :, 0] = 0
tb = tb.reshape(-1).contiguous()
s0 = kz.Kzg.load_trusted_setup(g1l, g2l)
out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, tb.data_ptr(), n, s0.handle) == 0
cs = out.raw; tc = torch.frombuffer(bytearray(cs), dtype=torch.uint8).to(dev)
assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, tb.data_ptr(), tc.data_ptr(), n, s0.handle) == 0
ps = out.raw
h = tb.cpu().numpy(); hp = h.ctypes.data_as(C.c_char_p)
del tb
s0.free()
ok = (C.c_bool * G)(); sg = (C.c_int * G)()
print(f"{n} blobs ({n * 131072 / 2**30:.1f} GiB) per call, cpus {len(os.sched_getaffinity(0))}", flush=True)
CONFIGS = [(512, 3, None), (256, 4, None), (1024, 3, None), (512, 3, 1)]
if len(sys.argv) > 2:
    CONFIGS = [tuple(int(v) if v != "auto" else None for v in c.split(",")) for c in sys.argv[2:]]
for chunk, inflight, threads in CONFIGS:
    os.environ["KZG355_CHUNK_MB"] = str(chunk); os.environ["KZG355_CHUNKS_IN_FLIGHT"] = str(inflight)
    if threads: os.environ["KZG355_COPY_THREADS"] = str(threads)
    else: os.environ.pop("KZG355_COPY_THREADS", None)
    os.environ["KZG355_MSM"] = "bucket"       # no 24 GB table per experiment handle
    if os.environ.get("STAGING"): os.environ["KZG355_STAGING"] = os.environ["STAGING"]
    s = kz.Kzg.load_trusted_setup(g1l, g2l)
    ts = []
    for i in range(4):
        t0 = time.perf_counter()
        rc = L.kzg355_verify_blob_kzg_proof_batch_many(ok, sg, hp, cs, ps, 64, G, s.handle)
        ts.append(time.perf_counter() - t0)
        assert rc == 0 and all(ok[j] for j in range(G))
    best = min(ts[1:])
    print(f"chunk {chunk:5d} MiB  in flight {inflight}  threads {threads or 'auto'}: first {ts[0]*1e3:7.1f} ms, best {best*1e3:7.1f} ms = {n/best/1e3:7.1f} k blobs/s = {n*131168/best/1e9:5.1f} GB/s", flush=True)
    s.free()
