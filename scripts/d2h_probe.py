"""Developer helper (GPU box): device -> pinned host bandwidth by chunk size and number of streams (plain torch copies)."""
import time
import torch
n = 320 << 20
x = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
def run(ns, chunk):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for i, o in enumerate(range(0, n, chunk)):
            with torch.cuda.stream(streams[i % ns]):
                h[o:o + chunk].copy_(x[o:o + chunk], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best
for ns in (1, 2, 4):
    for chunk in (8 << 20, 32 << 20, 320 << 20):
        dt = run(ns, chunk)
        print(f"D2H {ns} stream(s), chunks of {chunk >> 20:4d} MB: {1e3 * dt:6.2f} ms = {n / dt / 1e9:5.1f} GB/s", flush=True)
# and the other way
def run_up(ns, chunk):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for i, o in enumerate(range(0, n, chunk)):
            with torch.cuda.stream(streams[i % ns]):
                x[o:o + chunk].copy_(h[o:o + chunk], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best
for ns in (1, 2):
    dt = run_up(ns, 32 << 20)
    print(f"H2D {ns} stream(s), chunks of   32 MB: {1e3 * dt:6.2f} ms = {n / dt / 1e9:5.1f} GB/s", flush=True)
