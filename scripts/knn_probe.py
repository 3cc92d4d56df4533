"""Developer probe: one 100k x 100k kNN through the C ABI (run under rocprofv3 for kernel times)."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import synth_batches
from batchelor_amd import neighbors as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000  # queries
d = int(sys.argv[2]) if len(sys.argv) > 2 else 50
nref = int(sys.argv[3]) if len(sys.argv) > 3 else n
X, Q = synth_batches(2, [nref, n], d)
for it in range(3):
    t = time.time(); idx, dist = nb.query_knn(X, Q, 20); print("wall", time.time() - t, "fallbacks", nb.last_knn_exact_fallbacks(), flush=True)
