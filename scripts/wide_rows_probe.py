"""Developer helper (GPU box): what a search costs beyond 125 columns, where no candidate tier takes the rows and every query goes
through the FP64 scan (knn.hip: knn_exact_dist / knn_exact_select).   python scripts/wide_rows_probe.py [nx] [nq] [d ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batchelor_amd import neighbors as nb  # noqa: E402
from tests.conftest import synth_batches  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ds = [int(x) for x in sys.argv[3:]] or [100, 125, 126, 150, 200]
for d in ds:
    X, Q = synth_batches(5, [nx, nq], d)
    nb.query_knn(X, Q, 20)
    t = time.perf_counter()
    nb.query_knn(X, Q, 20)
    dt = time.perf_counter() - t
    print(f"nx={nx} nq={nq} d={d} k=20: {1e3 * dt:.1f} ms host to host, {nb.last_knn_exact_fallbacks()} queries through the FP64 paths", flush=True)
