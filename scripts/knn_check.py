"""Developer check: query_knn against the CPU oracle's brute-force kNN (memory-light; keep nq * nr modest: the oracle
is O(nq * nr * d) on the host cores)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.conftest import synth_batches
from batchelor_amd import neighbors as nb
from oracle import fastmnn_oracle as orc
nref, nq, d, k = (int(x) for x in sys.argv[1:5])
if nref * nq * d > 2e12:
    raise SystemExit("refusing: too large for a host-side check")
X, Q = synth_batches(7, [nref, nq], d)
idx, dist = nb.query_knn(X, Q, k)
oi, od = orc.query_knn(X, Q, k)
bad = np.nonzero((idx != oi).any(axis=1) | (dist != od).any(axis=1))[0]
print("variant fallbacks", nb.last_knn_exact_fallbacks(), "bad queries", bad.size, bad[:10])
for q in bad[:3]:
    print(q, idx[q], oi[q])
