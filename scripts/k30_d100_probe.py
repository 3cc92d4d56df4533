"""Developer helper (GPU box): k in (20, 50] at 100 PCs -- rows too long for the tiers with lists of 48 -- through the partitioned
search of knn.hip, a sample of the queries against the oracle.   python scripts/k30_d100_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batchelor_amd import neighbors as nb
from tests.conftest import synth_batches
from oracle import fastmnn_oracle as orc
X, Q = synth_batches(6, [100000, 20000], 100)
for k in (20, 30, 36, 50):
    nb.query_knn(X, Q, k)
    t = time.perf_counter(); idx, dist = nb.query_knn(X, Q, k); dt = time.perf_counter() - t
    rows = np.arange(0, 20000, 40)
    oi, od = orc.query_knn(X, Q[rows], k)
    print(f"d=100 k={k}: {1e3*dt:.1f} ms, exact fallbacks {nb.last_knn_exact_fallbacks()}, equal {np.array_equal(idx[rows], oi) and np.array_equal(dist[rows], od)}", flush=True)
