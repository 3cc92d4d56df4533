"""k <= 2 with more than two batches -- the shape clusterMNN() feeds the hot path (R/clusterMNN.R:147:
`reducedMNN(pca, k=1, merge.order=, auto.merge=)` on cluster centroids, any number of batches).

With k = 1 the tricube bandwidth of an MNN-involved cell is its distance to itself (0 -> 1e-8, R/utils_tricube.R:9), its
weight is 1 and a cell with a single pair lands EXACTLY on its left partner, up to the last bit of `r + (l - r)`
(R/fastMNN.R:606-607).  The next merge's reference then holds two cells an ulp apart, and which of the two a new cell
lists first is decided by the last bit of column means and dot products whose summation order the reference leaves to R's
BLAS (`%*%` in R/fastMNN.R:630, long-double `colMeans`) -- so neither the reference on another machine nor any
restatement is bit-reproducible there.  What every correct implementation must agree on, and what is asserted here on
the seeds the round-2 stress runs flagged (gpurun_out/estress.log cases 21/22/49/50, es11.log 24/69, es12.log 59):
  * the same merges in the same order, the same number of pairs per merge;
  * corrected coordinates within 1e-12 (relative to the column's scale) -- seven digits tighter than north_star's 1e-5;
  * the pair arrays equal as multisets once the members of every group of cells closer than 1e-9 (relative) to each
    other in the result -- the ulp twins -- are identified with the group's first cell (pairs are emitted by left cell
    id, so which twin was picked also decides where a pair sorts: order cannot be part of the claim here).
One shape is left out on purpose: a merge whose BOTH sides already hold twins (a tree like list(list(1,3), list(2,
list(4,5))): the right subtree was merged with k = 1 before it meets the left one).  Two tied left cells and two tied right
cells can come out as one mutual pair or as two depending on the last bits, so even the NUMBER of pairs -- and through the
mean over MNN cells (R/fastMNN.R:481) the batch vector, at the 1e-2 level -- is decided by rounding noise there, in the
reference as much as here (scripts/debug_tree_k1.py shows it: 37 against 38 pairs).  With a fresh batch on the right
(every default / progressive / auto order, and trees whose right child is a leaf) the claims above hold.
For k >= 3 and for two batches the plain bit-exact assertion of test_gpu_engine.py applies (and is what runs there)."""
import numpy as np
import pytest

from tests.conftest import synth_batches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bx():
    import batchelor_amd
    return batchelor_amd


def twin_representatives(corrected, rel=1e-9):
    """rep[i] = lowest row index of the group of rows within `rel` (times the data's scale) of row i (transitively)."""
    from scipy.spatial import cKDTree
    x = np.asarray(corrected, dtype=np.float64)
    scale = float(np.abs(x).max()) or 1.0
    rep = np.arange(x.shape[0])
    pairs = cKDTree(x).query_pairs(rel * scale, output_type="ndarray")

    def find(i):
        while rep[i] != i:
            rep[i] = rep[rep[i]]
            i = rep[i]
        return i

    for a, b in pairs:
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            rep[max(ra, rb)] = min(ra, rb)
    return np.array([find(i) for i in range(x.shape[0])])


def assert_same_up_to_twins(out, ref):
    assert out.corrected.shape == ref.corrected.shape
    scale = np.abs(ref.corrected).max(axis=0)
    err = (np.abs(out.corrected - ref.corrected).max(axis=0) / scale).max()
    assert err < 1e-12, err
    assert out.merge_info.left == ref.merge_info.left and out.merge_info.right == ref.merge_info.right
    rep = twin_representatives(ref.corrected)
    ntwins = int((rep != np.arange(rep.size)).sum())
    exact = True
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert ol.size == rl.size
        exact = exact and np.array_equal(ol, rl) and np.array_equal(orr, rr)
        # which twin a cell pairs with also decides where the pair sorts (pairs come out by left id): compare the
        # identified pairs as multisets
        mine = np.stack([rep[ol - 1], rep[orr - 1]], axis=1)
        want = np.stack([rep[rl - 1], rep[rr - 1]], axis=1)
        assert np.array_equal(mine[np.lexsort(mine.T[::-1])], want[np.lexsort(want.T[::-1])])
    np.testing.assert_allclose(out.merge_info.batch_size, ref.merge_info.batch_size, rtol=1e-9, equal_nan=True)
    assert np.array_equal(out.merge_info.skipped, ref.merge_info.skipped)
    np.testing.assert_allclose(out.merge_info.lost_var, ref.merge_info.lost_var, rtol=1e-7, atol=1e-12)
    return err, ntwins, exact


# (generator seed, case) of scripts/engine_stress.py draws that differed from the oracle in round 2, re-stated as the
# inputs they produced: batch sizes, dimensions, k, data seed
FLAGGED = [
    ([900, 150, 400, 150], 2, 1, 2000 + 1 * 1000 + 21),
    ([400, 900, 400, 400], 5, 1, 2000 + 1 * 1000 + 22),
    ([4500, 150, 900, 2000, 4500], 5, 1, 2000 + 1 * 1000 + 49),
    ([900, 150, 900, 2000], 5, 1, 2000 + 1 * 1000 + 50),
    ([60, 2000, 150], 80, 2, 2000 + 11 * 1000 + 24),
    ([400, 400, 60], 10, 2, 2000 + 11 * 1000 + 69),
    ([150, 900, 400, 150, 2000], 30, 2, 2000 + 12 * 1000 + 59),
]


@pytest.mark.parametrize("sizes,d,k,seed", FLAGGED)
def test_flagged_degenerate_draws(oracle, bx, sizes, d, k, seed):
    B = synth_batches(seed, sizes, d)
    out = bx.reducedMNN(*B, k=k)
    ref = oracle.reduced_mnn(*B, k=k)
    err, ntwins, _ = assert_same_up_to_twins(out, ref)
    assert ntwins > 0  # the degeneracy the test is about is really there


@pytest.mark.parametrize("kw", [{}, {"auto_merge": True}, {"merge_order": [[1, 3], [[4, 5], 2]]}])
def test_cluster_mnn_shape(oracle, bx, kw):
    # R/clusterMNN.R:147: a few dozen centroids per batch, k = 1, five batches, tree / auto-merge / default order
    rng = np.random.default_rng(147)
    centres = rng.standard_normal((40, 20)) * 3.0
    B = []
    for b in range(5):
        keep = np.sort(rng.choice(40, size=int(rng.integers(18, 36)), replace=False))
        B.append(centres[keep] + 0.05 * rng.standard_normal((keep.size, 20)) + 0.4 * b)
    out = bx.reducedMNN(*B, k=1, **kw)
    ref = oracle.reduced_mnn(*B, k=1, **kw)
    assert_same_up_to_twins(out, ref)


def test_seeded_draws_k1_k2(oracle, bx):
    # the same generator scripts/engine_stress.py runs for k <= 2 with more than two batches
    rng = np.random.default_rng(20250316)
    for case in range(10):
        nb = int(rng.integers(3, 6))
        sizes = [int(rng.choice([60, 150, 400, 900, 2000])) for _ in range(nb)]
        d = int(rng.choice([2, 5, 10, 30, 50, 80]))
        k = int(rng.choice([1, 2]))
        B = synth_batches(3000 + case, sizes, d)
        assert_same_up_to_twins(bx.reducedMNN(*B, k=k), oracle.reduced_mnn(*B, k=k))
