#!/usr/bin/env python3
"""Raw host->device bandwidth of this box (pinned memory, one big copy and 794 KB copies): the
ceiling of the per-call host-input path."""
import time
import torch
for mb in (256, 16, 0.775):
    n = int(mb * 1024 * 1024)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(10, int(2e9 / n)) if mb < 100 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("H2D %8.3f MB x %d: %.1f GB/s, %.1f us per copy" % (mb, reps, n * reps / el / 1e9, el / reps * 1e6))
