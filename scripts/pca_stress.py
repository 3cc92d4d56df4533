"""Developer helper (GPU box): the device multiBatchPCA (subspace iteration on the FP64 matrix cores) on random
low-rank-plus-noise batches against the SVD oracle.   python scripts/pca_stress.py <cases> <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import batchelor_amd as bx  # noqa: E402
from oracle import pca_oracle as pca  # noqa: E402

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    G = int(rng.choice([64, 130, 333, 1000, 2100]))
    r = int(rng.integers(6, 21))
    d = int(rng.integers(2, r - 2))
    nb = int(rng.integers(2, 5))
    sizes = [int(rng.choice([100, 257, 900, 3000])) for _ in range(nb)]
    cos_norm = bool(rng.integers(0, 2))
    weights = None if rng.integers(0, 2) else [float(x) for x in rng.uniform(0.3, 3.0, nb)]
    print("case", case, G, r, d, sizes, cos_norm, weights is not None, flush=True)
    load = rng.standard_normal((G, r)) * np.linspace(3.0, 1.0, r)
    mats = [load @ rng.standard_normal((r, n)) + 0.3 * rng.standard_normal((G, n)) + rng.uniform(-0.4, 0.4) for n in sizes]
    if cos_norm:
        mats = [m + 3.0 for m in mats]
    try:
        ref_in = [pca.cosine_norm(m) for m in mats] if cos_norm else mats
        ref, meta = pca.multi_batch_pca(ref_in, d=d, weights=weights, get_variance=True)
        mine = bx.multiBatchPCA(*mats, d=d, weights=weights, cos_norm=cos_norm)
        sgn = np.sign((mine["rotation"] * meta["rotation"]).sum(axis=0))
        np.testing.assert_allclose(mine["centers"], meta["centers"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(mine["rotation"] * sgn[None, :], meta["rotation"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(mine["d"] ** 2 / nb, meta["var.explained"], rtol=1e-8)
        for got, want in zip(mine["pcs"], ref):
            np.testing.assert_allclose(got * sgn[None, :], want, rtol=1e-5, atol=1e-7)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print("MISMATCH", repr(exc)[:400], flush=True)
print("cases", cases, "mismatches", bad, flush=True)
