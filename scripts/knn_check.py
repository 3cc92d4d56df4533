"""Developer check: query_knn against a brute-force numpy kNN (small sizes)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.conftest import synth_batches
from batchelor_amd import neighbors as nb
nref, nq, d, k = (int(x) for x in sys.argv[1:5])
X, Q = synth_batches(7, [nref, nq], d)
idx, dist = nb.query_knn(X, Q, k)
d2 = ((Q[:, None, :] - X[None, :, :]) ** 2).sum(-1)
ref = np.argsort(d2, axis=1, kind="stable")[:, :k]
bad = np.nonzero((idx - 1 != ref).any(axis=1))[0]
print("variant fallbacks", nb.last_knn_exact_fallbacks(), "bad queries", bad.size, bad[:10])
for q in bad[:3]:
    print(q, idx[q], ref[q])
