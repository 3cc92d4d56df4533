"""Developer helper (GPU box): one emulated rank's step with the engine's diagnostics (retries, exact-path queries).
   python scripts/emu_rank_probe.py <world> <rank> [<rank> ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import batchelor_amd as bx  # noqa: E402
from bench import WORKLOADS, synth_batches  # noqa: E402

world = int(sys.argv[1])
ranks = [int(x) for x in sys.argv[2:]]
cfg, sizes, d, k, tree = WORKLOADS["config3"]
B = synth_batches(cfg, sizes, d)
eng = bx.MnnEngine(0)
eng.upload(B)
eng.run(k=k)
eng.emulate(1)
eng.run(k=k)
for r in ranks:
    eng.emulate(2, r, world)
    for _ in range(2):
        eng.run(k=k)
    eng.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.run(k=k)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 3
    p = eng.profile_detail()
    eng.set_profiling(False)
    print(f"rank {r} of {world}: {ms:.2f} ms per step; last run: {p}", flush=True)
eng.close()
