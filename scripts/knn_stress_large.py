"""Developer helper (GPU box): large kNN shapes (whole rounds of workgroups plus split tails, shared thresholds, long
sweeps) through the C ABI; a random subset of the queries against the oracle's brute force, bit for bit.
   python scripts/knn_stress_large.py <cases> <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import synth_batches  # noqa: E402
from batchelor_amd import _lib, neighbors as nb  # noqa: E402
from oracle import fastmnn_oracle as oracle  # noqa: E402

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    nx = int(rng.choice([40000, 100000, 250000, 600000]))
    nq = int(rng.choice([30000, 66000, 100000, 131072, 200000]))
    d = int(rng.choice([10, 30, 50, 64, 100]))
    k = int(rng.choice([5, 20, 30]))
    split = str(rng.choice(["", "", "3", "7"]))
    _lib.dev_set("split_c", int(split) if split else 0)  # testing hook of the library (bmx_dev_set)
    print("case", case, nx, nq, d, k, repr(split), flush=True)
    X, Q = synth_batches(5000 + seed * 100 + case, [nx, nq], d)
    idx, dist = nb.query_knn(X, Q, k)
    rows = np.sort(rng.choice(nq, 1500, replace=False))
    oi, od = oracle.query_knn(X, Q[rows], k)
    ok = np.array_equal(idx[rows], oi) and np.array_equal(dist[rows], od)
    print("   fallbacks", nb.last_knn_exact_fallbacks(), "ok" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
print("cases", cases, "mismatches", bad, flush=True)
