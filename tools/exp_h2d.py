#!/usr/bin/env python3
"""Raw host->device copy rates on this box: pinned and pageable sources, 64 MiB .. 2 GiB (what bounds the host-buffer entry points)."""
import time, torch
dev = torch.device("cuda", 0)
for mb in (8, 64, 512, 2048):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    hp = torch.empty(n, dtype=torch.uint8).pin_memory(); hp.random_(0, 255)
    hq = torch.empty(n, dtype=torch.uint8); hq.random_(0, 255)
    for name, h in (("pinned", hp), ("pageable", hq)):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{mb:5d} MiB {name:9s}: best {n / min(ts) / 1e9:6.1f} GB/s  median {n / sorted(ts)[2] / 1e9:6.1f} GB/s", flush=True)
    t0 = time.perf_counter(); hp.copy_(hq); dt = time.perf_counter() - t0
    print(f"{mb:5d} MiB host memcpy pageable->pinned (1 thread): {n / dt / 1e9:5.1f} GB/s", flush=True)
