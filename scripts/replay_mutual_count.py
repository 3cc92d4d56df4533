"""Analysis helper (CPU only, not a test, not shipped): VERDICT r3 #3, experiment (i).

Question: can the second search of findMutualNN (every LISTED left cell's k2 nearest right cells, 25 of the 93 ms of a
config-3 step) be replaced by COUNTING?  A pair (l, r) that the first search found (r lists l at distance d(l, r)) is mutual
iff fewer than k2 right cells are closer to l than r.  A listed left cell l drops out altogether when even its NEAREST
lister has k2 right cells inside its distance -- decidable by a threshold-only sweep (no survivor path in the candidate
kernel) if the count can be taken from fp16 products, i.e. with the pass's error margin on the threshold.

Replays the bench generator (bench.synth_batches, config-3 shape at reduced size) through the CPU oracle's merge loop,
captures the two matrices every merge hands to findMutualNN, and reports per merge
  * the share of listed left cells that are in no mutual pair ("inactive"),
  * how many of them a threshold-only count with the fp16 margin (d^2 < m^2 - 2 eps) settles,
  * how many a count over a SAMPLE of the right cells settles (early exit),
  * the share of candidate pairs 1 / 2 / 4 thresholds per left cell decide.
   python scripts/replay_mutual_count.py [cells per batch] [batches]
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batches  # noqa: E402
from oracle import fastmnn_oracle as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
K = 20
EPS2 = 2 * 0.05  # twice the fp16 pass's error bound at 50 PCs (squared-distance units)

captured = []
_orig = orc.restricted_mnn


def _spy(left_data, left_restrict, right_data, right_restrict, k, prop_k=None, nthreads=0):
    captured.append((left_data.copy(), right_data.copy()))
    return _orig(left_data, left_restrict, right_data, right_restrict, k, prop_k, nthreads)


orc.restricted_mnn = _spy
B = synth_batches(3, [n] * nb, 50)
orc.reduced_mnn(*B)

print(f"# {nb} batches x {n} cells x 50 PCs, k = {K}; eps margin on d^2: {EPS2}")
for m, (L, R) in enumerate(captured):
    tL, tR = cKDTree(L), cKDTree(R)
    dRL, iRL = tL.query(R, K, workers=-1)          # S1: every right cell's k nearest left cells
    listed = np.unique(iRL)
    nsel = listed.size
    # per listed left cell: its listers' distances
    order = np.argsort(iRL.ravel(), kind="stable")
    lflat, dflat = iRL.ravel()[order], dRL.ravel()[order]
    starts = np.searchsorted(lflat, listed)
    ends = np.append(starts[1:], lflat.size)
    dmin = np.minimum.reduceat(dflat, starts)
    dmax = np.maximum.reduceat(dflat, starts)
    nlist = ends - starts
    Lq = L[listed]
    # exact: right cells strictly closer than the nearest lister (the lister itself sits AT dmin)
    c_min = tR.query_ball_point(Lq, dmin * (1 - 1e-12), return_length=True, workers=-1)
    inactive = c_min >= K
    # with the fp16 margin: only right cells DEFINITELY inside count
    rdef = np.sqrt(np.maximum(dmin ** 2 - EPS2, 0.0))
    c_def = tR.query_ball_point(Lq, rdef, return_length=True, workers=-1)
    settled = c_def >= K
    # sample of the right cells (1/4, 1/8): a hypergeometric draw of the definitely-inside count
    rng = np.random.default_rng(m)
    s4 = rng.binomial(c_def, 0.25) >= K
    s8 = rng.binomial(c_def, 0.125) >= K
    # S2 for reference: the true k-th right distance of each listed left cell
    dLR, _ = tR.query(Lq, K, workers=-1)
    dk = dLR[:, -1]
    mutual = dflat <= np.repeat(dk, nlist)
    P = int(mutual.sum())
    active = np.add.reduceat(mutual.astype(np.int64), starts) > 0
    assert np.array_equal(active, ~inactive) or abs(int(active.sum()) - int((~inactive).sum())) < 5
    # pairs decided by t thresholds per left cell: thresholds at quantiles of its lister distances (with margin both ways)
    dec = {}
    for t in (1, 2, 4):
        decided = np.zeros(dflat.size, dtype=bool)
        for qi in range(t):
            q = (qi + 1) / (t + 1) if t > 1 else 0.0
            thr = np.array([np.quantile(dflat[s:e], q) for s, e in zip(starts[:2000], ends[:2000])])
            sub = slice(0, ends[1999] if starts.size >= 2000 else dflat.size)
            cin = tR.query_ball_point(Lq[:thr.size], np.sqrt(np.maximum(thr ** 2 - EPS2, 0)), return_length=True, workers=-1)
            cout = tR.query_ball_point(Lq[:thr.size], np.sqrt(thr ** 2 + EPS2), return_length=True, workers=-1)
            thr_r, cin_r, cout_r = (np.repeat(a, nlist[:thr.size]) for a in (thr, cin, cout))
            dsub = dflat[sub]
            decided[sub] |= (dsub >= thr_r) & (cin_r >= K)        # at least k right cells definitely closer: not mutual
            decided[sub] |= (dsub <= thr_r) & (cout_r < K)        # fewer than k right cells possibly closer: mutual
        dec[t] = decided[sub].mean()
    print(f"merge {m + 1}: nL={L.shape[0]} nR={R.shape[0]} listed={nsel} ({nsel / L.shape[0]:.2f} of left) "
          f"pairs={P} active={int(active.sum())} ({active.mean():.3f} of listed)")
    print(f"   inactive listed cells: {inactive.mean():.3f} of listed; settled by a margin count over ALL right cells: "
          f"{settled.mean():.3f} of listed ({settled.sum() / max(1, inactive.sum()):.3f} of the inactive); over a 1/4 "
          f"sample {s4.mean():.3f}, a 1/8 sample {s8.mean():.3f}")
    q = np.quantile(c_min[inactive], [0.1, 0.25, 0.5, 0.75, 0.9]) if inactive.any() else []
    print(f"   right cells inside the nearest lister's distance, inactive cells (10/25/50/75/90 %): {np.round(q, 0)}; "
          f"listers per listed cell median {np.median(nlist):.0f}, seed/nearest distance ratio median {np.median(dmax / dmin):.2f}")
    print(f"   candidate pairs decided by 1 / 2 / 4 thresholds per left cell (first 2000 listed cells): "
          f"{dec[1]:.3f} / {dec[2]:.3f} / {dec[4]:.3f}")
