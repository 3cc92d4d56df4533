"""GPU checks at BASELINE.json's full size (config 2: 2 x 100 000 cells x 50 PCs, k = 20), through properties that do
not need the CPU oracle to repeat the whole job:

* exact kNN: a random sample of the query rows is compared with the oracle's brute-force answer against the FULL
  reference (indices and distances bitwise), and every row's distances are sorted with ties in index order;
* MNN pairs: the engine's pairs (found with the second search restricted to the left cells that occur in some right
  cell's list) are exactly the mutual intersection of the two FULL neighbour lists, in the reference's order
  (src/find_mutual_nns.cpp:23-36: left ascending, then the left cell's neighbour rank);
* the engine is deterministic, leaves the first batch's geometry untouched except for the common shift along the
  batch vector, and its tricube-smoothed correction moves every cell of the second batch by a finite amount.
"""
import numpy as np
import pytest

from tests.conftest import synth_batches

pytestmark = pytest.mark.gpu

N, D, K = 100_000, 50, 20


@pytest.fixture(scope="module")
def data():
    return synth_batches(2, [N, N], D)


@pytest.fixture(scope="module")
def lists(data):
    from batchelor_amd import neighbors as nb
    L, R = data
    idx_lr, dist_lr = nb.query_knn(R, L, K)   # for each left cell: its K nearest right cells
    fb1 = nb.last_knn_exact_fallbacks()
    idx_rl, dist_rl = nb.query_knn(L, R, K)   # for each right cell: its K nearest left cells
    fb2 = nb.last_knn_exact_fallbacks()
    return idx_lr, dist_lr, idx_rl, dist_rl, fb1 + fb2


def test_full_size_knn_sample_matches_oracle(oracle, data, lists):
    L, R = data
    idx_lr, dist_lr, idx_rl, dist_rl, fallbacks = lists
    assert fallbacks <= 200                     # the certificate holds for all but a handful of 200 000 queries
    rng = np.random.default_rng(5)
    rows = np.sort(rng.choice(N, 1500, replace=False))
    oi, od = oracle.query_knn(R, L[rows], K)
    assert np.array_equal(idx_lr[rows], oi) and np.array_equal(dist_lr[rows], od)
    oi, od = oracle.query_knn(L, R[rows], K)
    assert np.array_equal(idx_rl[rows], oi) and np.array_equal(dist_rl[rows], od)
    for idx, dist in ((idx_lr, dist_lr), (idx_rl, dist_rl)):
        assert idx.min() >= 1 and idx.max() <= N
        assert np.all(np.diff(dist, axis=1) >= 0)                      # ascending distances ...
        tie = np.diff(dist, axis=1) == 0
        assert np.all(np.diff(idx, axis=1)[tie] > 0)                   # ... ties by lowest index
        assert np.all(np.sort(idx, axis=1)[:, 1:] != np.sort(idx, axis=1)[:, :-1])   # no duplicates in a row


def mutual_pairs(idx_lr, idx_rl):
    """The reference's pair order from two full 1-based neighbour lists."""
    n_l, k2 = idx_lr.shape
    left = np.repeat(np.arange(1, n_l + 1, dtype=np.int64), k2)
    right = idx_lr.reshape(-1).astype(np.int64)
    back = np.repeat(np.arange(1, idx_rl.shape[0] + 1, dtype=np.int64), idx_rl.shape[1]) * (n_l + 1) \
        + idx_rl.reshape(-1)                                           # key (right cell, its left neighbour)
    keep = np.isin(right * (n_l + 1) + left, back)
    return left[keep].astype(np.int32), right[keep].astype(np.int32)


def test_full_size_pairs_are_the_mutual_intersection(data, lists):
    import batchelor_amd as bx
    idx_lr, _, idx_rl, _, _ = lists
    first, second = mutual_pairs(idx_lr, idx_rl)
    out = bx.reducedMNN(*data, k=K)
    got_l, got_r = out.merge_info.pairs[0]
    assert got_l.size == first.size
    assert np.array_equal(got_l, first)
    assert np.array_equal(got_r - N, second)    # the engine numbers cells over the concatenated batches


def test_full_size_engine_properties(data):
    import batchelor_amd as bx
    L, R = data
    a = bx.reducedMNN(L, R, k=K)
    b = bx.reducedMNN(L, R, k=K)
    assert np.array_equal(a.corrected, b.corrected)                    # deterministic, bit for bit
    assert np.array_equal(a.merge_info.pairs[0][0], b.merge_info.pairs[0][0])
    assert np.all(np.isfinite(a.corrected))
    # the reference batch is only shifted along the batch vector (R/fastMNN.R:626-640): all pairwise differences of
    # its cells orthogonal to that vector are untouched -- check through the rank of the displacement field
    disp = a.corrected[:N] - L
    s = np.linalg.svd(disp[:2000] - disp[:2000].mean(axis=0), compute_uv=False)
    assert s[1] <= 1e-9 * max(s[0], 1e-300)                            # rank one
    # the second batch moved towards the first: the batch offset along dimension 2 (shift = 1 in synth_batches) shrinks
    before = abs(R[:, 1].mean() - L[:, 1].mean())
    after = abs(a.corrected[N:, 1].mean() - a.corrected[:N, 1].mean())
    assert after < 0.5 * before
    lost = np.asarray(a.merge_info.lost_var)
    assert lost.shape == (1, 2) and np.all(lost >= -1e-12) and np.all(lost < 0.5)


@pytest.mark.parametrize("k", [50, 100])
def test_full_size_knn_beyond_the_tiers_lists(oracle, data, k):
    """k > 36 at BASELINE config 2's size (`prop.k` = 0.001 of 100 000 cells is k = 100, R/MNN_tree.R:140-146): the partitioned
    search of knn.hip (large_k_search) on the matrix cores instead of the FP64 scan -- 600 sampled queries against the
    oracle's brute force over the FULL reference, indices and distances bitwise."""
    from batchelor_amd import neighbors as nb
    L, R = data
    idx, dist = nb.query_knn(R, L, k)
    assert nb.last_knn_exact_fallbacks() <= 2000
    rows = np.sort(np.random.default_rng(50 + k).choice(N, 600, replace=False))
    oi, od = oracle.query_knn(R, L[rows], k)
    assert np.array_equal(idx[rows], oi) and np.array_equal(dist[rows], od)
    assert np.all(np.diff(dist, axis=1) >= 0)


def test_full_size_knn_k_5000(oracle, data):
    """k = 5 000 = prop.k 0.05 of BASELINE config 2's 100 000 cells (R/MNN_tree.R:140-146), all 100 000 queries: the partitioned
    search with the big merge (knn.hip: lk_merge_big) instead of seconds of FP64 scan; 200 sampled rows against the oracle's brute
    force over the full reference, indices and distances bitwise.  The time is printed (< 1 s was the mark)."""
    import time
    import torch
    from batchelor_amd import neighbors as nb
    L, R = data
    nb.query_knn(R, L[:2000], 5000)  # (workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, dist = nb.query_knn(R, L, 5000)
    dt = time.perf_counter() - t0
    print(f"query_knn k = 5000, {N} x {N} cells, host to host ({idx.nbytes + dist.nbytes >> 20} MB of results): {dt:.2f} s; "
          f"{nb.last_knn_exact_fallbacks()} queries through the FP64 scan")
    rows = np.sort(np.random.default_rng(5000).choice(N, 200, replace=False))
    oi, od = oracle.query_knn(R, L[rows], 5000)
    assert np.array_equal(idx[rows], oi) and np.array_equal(dist[rows], od)
    assert nb.last_knn_exact_fallbacks() <= 2000


def test_full_size_prop_k_run(data):
    # reducedMNN(prop.k = 0.0005) on 2 x 100 000 cells: k = max(20, round(50)) = 50 for every search of the merge
    import batchelor_amd as bx
    a = bx.reducedMNN(*data, k=K, prop_k=0.0005)
    b = bx.reducedMNN(*data, k=50)
    assert np.array_equal(a.corrected, b.corrected)
    assert np.array_equal(a.merge_info.pairs[0][0], b.merge_info.pairs[0][0])
    assert a.merge_info.pairs[0][0].size > 1_000_000
