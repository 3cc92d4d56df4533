"""Developer helper (GPU box): the large-k search with and without seeded partitions -- time and the rows left to the FP64 paths.
   python scripts/lk_seed_probe.py [nx] [nq] [d] [k ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batchelor_amd import _lib, neighbors as nb  # noqa: E402
from tests.conftest import synth_batches  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 50
ks = [int(x) for x in sys.argv[4:]] or [100, 300, 1000]
X, Q = synth_batches(21, [nx, nq], d)
for k in ks:
    for seeded in (0, 1, 0, 1):
        _lib.dev_set("lk_seed", seeded)
        t = time.perf_counter()
        nb.query_knn(X, Q, k)
        dt = time.perf_counter() - t
        print(f"nx={nx} nq={nq} d={d} k={k} lk_seed={seeded}: {1e3 * dt:.1f} ms, {nb.last_knn_exact_fallbacks()} rows to the FP64 paths", flush=True)
_lib.dev_set("reset", 0)
