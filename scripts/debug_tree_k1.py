"""Developer helper: where the k = 1 tree case of tests/test_gpu_degenerate_k.py diverges from the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import batchelor_amd as bx
from oracle import fastmnn_oracle as oracle
from tests.test_gpu_degenerate_k import twin_representatives

rng = np.random.default_rng(147)
centres = rng.standard_normal((40, 20)) * 3.0
B = []
for b in range(5):
    keep = np.sort(rng.choice(40, size=int(rng.integers(18, 36)), replace=False))
    B.append(centres[keep] + 0.05 * rng.standard_normal((keep.size, 20)) + 0.4 * b)
for k in (1, 3):
    for mo in ([[1, 3], [2, [4, 5]]], [[1, 3], [[4, 5], 2]], [1, 3, 2, 4, 5]):
        out = bx.reducedMNN(*B, k=k, merge_order=mo)
        ref = oracle.reduced_mnn(*B, k=k, merge_order=mo)
        err = np.abs(out.corrected - ref.corrected).max()
        print("k", k, mo, "err", err, "left", out.merge_info.left, ref.merge_info.left)
        rep = twin_representatives(ref.corrected)
        for m, ((ol, orr), (rl, rr)) in enumerate(zip(out.merge_info.pairs, ref.merge_info.pairs)):
            same = ol.size == rl.size and np.array_equal(ol, rl) and np.array_equal(orr, rr)
            print("   merge", m, "P", ol.size, rl.size, "exact" if same else "DIFF")
            if not same and ol.size == rl.size:
                bad = np.flatnonzero((ol != rl) | (orr != rr))
                print("     first diffs", [(int(ol[i]), int(orr[i]), int(rl[i]), int(rr[i])) for i in bad[:6]])
        sizes = np.cumsum([0] + [b.shape[0] for b in B])
        for b in range(5):
            e = np.abs(out.corrected[sizes[b]:sizes[b+1]] - ref.corrected[sizes[b]:sizes[b+1]]).max()
            print("   batch", b + 1, "max err", e)
        print("   batch.size", out.merge_info.batch_size, ref.merge_info.batch_size)
