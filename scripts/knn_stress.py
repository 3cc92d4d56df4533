"""Developer helper (GPU box): random kNN shapes through the C ABI against the CPU oracle, bit for bit.
   python scripts/knn_stress.py <cases> <seed>"""
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
    nx = int(rng.choice([64, 70, 130, 500, 2100, 4095, 4100, 9000, 20000, 33000, 70000]))
    nq = int(rng.choice([1, 31, 257, 900, 2500, 6000]))
    d = int(rng.integers(2, 126)) if rng.random() < 0.85 else int(rng.integers(126, 220))  # (beyond 125: the FP64 scan)
    # (beyond 36: the partitioned search of knn.hip; beyond ~900: its big merge)
    k = int(min(rng.choice([1, 5, 20, 21, 36, 37, 50, 64, 65, 100, 150, 300, 700, 1200, 2500]), nx))
    dup = int(rng.choice([1, 1, 1, 2, 3]))  # every reference cell that many times: ties at the k-th place, decided by position
    force_c = str(rng.choice(["", "1", "2", "3", "5", "7"]))
    sample = str(rng.choice(["", "", "0", "1024", "4096"]))
    _lib.dev_set("force_c", int(force_c) if force_c else 0)  # testing hooks of the library (bmx_dev_set)
    _lib.dev_set("sample", int(sample) if sample else -1)
    print("case", case, nx, nq, d, k, dup, repr(force_c), repr(sample), flush=True)
    X, Q = synth_batches(1000 + seed * 1000 + case, [max(nx // dup, 1), nq], d)
    X = np.concatenate([X] * dup)
    k = min(k, X.shape[0])
    idx, dist = nb.query_knn(X, Q, k)
    oi, od = oracle.query_knn(X, Q, k)
    ok = np.array_equal(idx, oi) and np.array_equal(dist, od)
    if not ok:
        bad += 1
        print("MISMATCH", case, nx, nq, d, k, force_c, sample, flush=True)
print("cases", cases, "mismatches", bad, flush=True)
