#!/usr/bin/env python3
"""Shader clock and socket power DURING each kernel of a verify step (round 5).  One 8192-batch step is a fixed sequence of kernels, each filling the card for
milliseconds: validate_points, challenge, eval, rpowers, lincomb_prep, lincomb, lincomb_horner, pairing.  A thread samples the GPU's hwmon files (freq1_input,
power1_input) every millisecond with host timestamps while the main thread makes synchronous steps; the HIP-event durations of the kernels (kzg355_kernel_ms_stats)
cut every step into its kernels in launch order, and the samples are binned by kernel.  Why: DESIGN.md section 4a' quotes every kernel against the floor of its
instruction mix at ONE clock; if the card runs the kernels at different clocks, part of a kernel's distance from its floor is the clock, not its code.
usage: exp_clock_per_kernel.py [steps]"""
import ctypes as C, glob, os, statistics, sys, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
import kzg_rust_amd as kz
from synth import random_blob

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G, N, BLOB = 8192, 64, 131072
dev = torch.device("cuda", 0)
g = os.path.join(ROOT, "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib()
gen = torch.Generator(device=dev); gen.manual_seed(0x4844)
nb = G * N
tb = torch.randint(0, 256, (nb, BLOB // 32, 32), dtype=torch.uint8, device=dev, generator=gen); tb[:, :, 0] = 0; tb = tb.reshape(-1).contiguous()
out = C.create_string_buffer(48 * 65536); st = (C.c_int * nb)()
cs = bytearray(); ps = bytearray()
for lo in range(0, nb, 65536):
    assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, tb.data_ptr() + lo * BLOB, 65536, s.handle) == 0; cs += out.raw
tc = torch.frombuffer(cs, dtype=torch.uint8).to(dev)
for lo in range(0, nb, 65536):
    assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, tb.data_ptr() + lo * BLOB, tc.data_ptr() + lo * 48, 65536, s.handle) == 0; ps += out.raw
tp = torch.frombuffer(ps, dtype=torch.uint8).to(dev)
torch.cuda.synchronize()
pr = torch.cuda.get_device_properties(0)
hw = glob.glob(f"/sys/bus/pci/devices/{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0/hwmon/hwmon*")[0]
f_clk, f_pow = hw + "/freq1_input", hw + "/power1_input"
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        t = time.perf_counter()
        try:
            samples.append((t, int(open(f_clk).read()) / 1e6, int(open(f_pow).read()) / 1e6))
        except Exception:
            pass
        time.sleep(0.0005)


ok = (C.c_bool * G)(); stg = (C.c_int * G)()
for _ in range(2):
    assert L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), N, G, s.handle) == 0
L.kzg355_reset_kernel_stats(s.handle); s.set_kernel_timing(True)
th = threading.Thread(target=sampler, daemon=True); th.start()
windows = []
for _ in range(steps):
    t0 = time.perf_counter()
    assert L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), N, G, s.handle) == 0
    windows.append((t0, time.perf_counter()))
stop.set(); th.join()
s.set_kernel_timing(False)
order = ["validate_points", "challenge", "eval", "rpowers", "lincomb_prep", "lincomb", "lincomb_horner", "pairing"]
ms = {}
for fam in order:
    tot, cnt = C.c_double(), C.c_long()
    L.kzg355_kernel_ms_stats(s.handle, fam.encode(), C.byref(tot), C.byref(cnt))
    ms[fam] = tot.value / max(1, cnt.value)
total = sum(ms.values())
print(f"step {statistics.median(b - a for a, b in windows) * 1e3:.2f} ms (host), kernels {total:.2f} ms; {len(samples)} samples, {steps} steps")
bins = {fam: [] for fam in order}
for a, b in windows:
    # the kernels end when the call returns (minus the read-back of the verdicts: microseconds): lay them out backwards from the end of the call
    t_end = b
    edges = []
    for fam in reversed(order):
        edges.append((fam, t_end - ms[fam] / 1e3, t_end)); t_end -= ms[fam] / 1e3
    for t, clk, pw in samples:
        for fam, lo, hi in edges:
            if lo + 0.0005 <= t < hi - 0.0005:                   # half a millisecond of margin at both ends
                bins[fam].append((clk, pw))
print(f"{'kernel':18s} {'ms':>7s} {'samples':>8s} {'sclk MHz median (p10-p90)':>28s} {'power W median':>15s}")
for fam in order:
    v = bins[fam]
    if len(v) < 3:
        print(f"{fam:18s} {ms[fam]:7.2f} {len(v):8d}   (too short to sample)")
        continue
    c = sorted(x for x, _ in v); p = sorted(y for _, y in v)
    q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]
    print(f"{fam:18s} {ms[fam]:7.2f} {len(v):8d} {q(c, .5):12.0f} ({q(c, .1):.0f}-{q(c, .9):.0f}) {q(p, .5):15.0f}")
s.free()
