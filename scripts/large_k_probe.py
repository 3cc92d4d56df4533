"""Developer helper (GPU box): config 2 (2 x n cells x 50 PCs) at k beyond the candidate tiers' 36 -- what the FP64 scan costs.
   python scripts/large_k_probe.py [n] [k ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import batchelor_amd as bx  # noqa: E402
from tests.conftest import synth_batches  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ks = [int(x) for x in sys.argv[2:]] or [20, 36, 37, 50, 100]
B = synth_batches(2, [n, n], 50)
eng = bx.MnnEngine()
eng.upload(B)
for k in ks:
    eng.run(k=k)
    t = time.perf_counter()
    eng.run(k=k)
    dt = time.perf_counter() - t
    st = eng.merge_stats()[0]
    pd = eng.profile_detail()
    print(f"n={n} k={k}: {1e3 * dt:.1f} ms per step, {st['P']} pairs, kernel {pd['kernel']}, "
          f"{pd['exact_fallbacks']} queries to the FP64 paths, optimistic runs given up so far: {pd['optimistic_retries']}", flush=True)
eng.close()
