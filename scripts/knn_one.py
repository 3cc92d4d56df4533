import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import synth_batches
from batchelor_amd import neighbors as nb
from oracle import fastmnn_oracle as oracle
nx, nq, d, k = map(int, sys.argv[1:5])
X, Q = synth_batches(7, [nx, nq], d)
idx, dist = nb.query_knn(X, Q, k)
oi, od = oracle.query_knn(X, Q, k)
print("ok" if np.array_equal(idx, oi) and np.array_equal(dist, od) else "MISMATCH", nx, nq, d, k, flush=True)
