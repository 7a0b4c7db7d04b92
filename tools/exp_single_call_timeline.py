#!/usr/bin/env python3
"""Timeline of ONE verify_blob_kzg_proof_batch(n = 64) call on host slices (benches/kzg_benches.rs:113-120) from a rocprofv3 kernel trace:
which kernel starts when, on which queue, and where the gaps are.

    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o tl --output-format csv -- python3 tools/exp_single_call_timeline.py run
    python3 tools/exp_single_call_timeline.py parse gpurun_out/tl > profiles/r04/single_call_timeline.txt

`run` makes 40 calls 3 ms apart (so that the calls are separated by idle time in the trace); `parse` prints the median call."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kzg_rust_amd as kz
    from synth import random_blob
    g = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
    s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    n = int(os.environ.get("N", "64"))
    blobs = [random_blob(9000 + i) for i in range(n)]
    cs = [kz.Kzg.blob_to_kzg_commitment(b, s) for b in blobs]
    ps = [kz.Kzg.compute_blob_kzg_proof(b, c, s) for b, c in zip(blobs, cs)]
    ts = []
    op = os.environ.get("OP", "verify_batch")          # or: commit, proof (compute_kzg_proof), blob_proof, verify_proof, verify_blob
    z = bytes(31) + b"\x05"
    y = kz.Kzg.compute_kzg_proof(blobs[0], z, s)[1] if op == "verify_proof" else None
    pz = kz.Kzg.compute_kzg_proof(blobs[0], z, s)[0] if op == "verify_proof" else None
    for _ in range(40):
        time.sleep(0.003)
        t0 = time.perf_counter()
        if op == "verify_batch": assert kz.Kzg.verify_blob_kzg_proof_batch(blobs, cs, ps, s)
        elif op == "commit": kz.Kzg.blob_to_kzg_commitment(blobs[0], s)
        elif op == "proof": kz.Kzg.compute_kzg_proof(blobs[0], z, s)
        elif op == "blob_proof": kz.Kzg.compute_blob_kzg_proof(blobs[0], cs[0], s)
        elif op == "verify_proof": assert kz.Kzg.verify_kzg_proof(cs[0], z, y, pz, s)
        elif op == "verify_blob": assert kz.Kzg.verify_blob_kzg_proof(blobs[0], cs[0], ps[0], s)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"wall per call (python wrapper included): median {sorted(ts)[20]:.3f} ms, min {min(ts):.3f} ms", file=sys.stderr)
    s.free()


def parse(directory):
    rows = []
    for f in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
    for f in glob.glob(os.path.join(directory, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?")) + " B", "dma"))
    rows.sort()
    # calls are separated by >= 1 ms of idle device time; a verify call ends with a pairing kernel
    calls, cur = [], []
    for r in rows:
        if cur and r[0] - max(x[1] for x in cur) > 1_000_000:
            calls.append(cur); cur = []
        cur.append(r)
    if cur:
        calls.append(cur)
    need = os.environ.get("NEED", "k_pairing,k_challenge_from_digest").split(",")      # kernels a call of the traced operation must contain
    calls = [c for c in calls if all(any(x[2].startswith(k) for x in c) for k in need)]
    assert calls, "no call with the kernels " + str(need) + " in the trace"
    calls = calls[len(calls) // 4:]                    # (the first calls of a run include one-time work)
    calls.sort(key=lambda c: max(x[1] for x in c) - c[0][0])
    c = calls[len(calls) // 2]
    t0 = c[0][0]
    print(f"{len(calls)} calls in the trace; the median one by device span ({(max(x[1] for x in c) - t0) / 1e3:.1f} us from its first device activity to the end of its pairing):")
    print(f"  {'start us':>9} {'end us':>9} {'dur us':>8}  queue  what")
    for st, en, name, q in c:
        print(f"  {(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  {q:>5}  {name}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2])
