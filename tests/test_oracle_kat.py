"""Pins the CPU oracle (oracle/) against the known-answer tests and executable specifications that the reference's
own test-suite holds for the fastMNN / reducedMNN hot path.  Citations are relative to /root/reference.

No GPU and no product code here: this file is the evidence that the oracle restates the reference."""
import math

import numpy as np
import pytest

from tests.conftest import synth_batches


# ---------------------------------------------------------------- tests/testthat/test-reduced-mnn.R:80-105
def _grid():
    core = np.column_stack([np.repeat(np.arange(1, 11), 10), np.tile(np.arange(1, 11), 10)]).astype(np.float64)
    b1 = core.copy()
    b1[:, 0] += 20
    b2 = core.copy()
    b2[:, 1] += 20
    return core, b1, b2


def test_grid_kat_two_batches(oracle):
    core, b1, _ = _grid()
    out = oracle.reduced_mnn(core, b1, k=1)
    np.testing.assert_allclose(out.corrected[:, 0], np.full(200, 5.5), rtol=0, atol=1e-12)
    np.testing.assert_allclose(out.corrected[:, 1], np.concatenate([core[:, 1], b1[:, 1]]), rtol=0, atol=1e-12)


def test_grid_kat_three_batches(oracle):
    core, b1, b2 = _grid()
    out = oracle.reduced_mnn(core, b1, b2, k=1)
    np.testing.assert_allclose(out.corrected[:, 0], np.full(300, 5.5), rtol=0, atol=1e-12)
    np.testing.assert_allclose(out.corrected[:, 1], np.full(300, 5.5), rtol=0, atol=1e-12)


def test_grid_kat_hierarchical(oracle):
    core, b1, b2 = _grid()
    outY = oracle.reduced_mnn(core + 10, b2 + 10, k=1)
    np.testing.assert_allclose(outY.corrected[:, 0], np.concatenate([core[:, 0], b2[:, 0]]) + 10, atol=1e-12)
    np.testing.assert_allclose(outY.corrected[:, 1], np.full(200, 15.5), atol=1e-12)
    outZ = oracle.reduced_mnn(core, b1, core + 10, b2 + 10, merge_order=[[1, 2], [3, 4]], k=1)
    np.testing.assert_allclose(outZ.corrected[:, 0], np.full(400, 5.5), atol=1e-12)
    np.testing.assert_allclose(outZ.corrected[:, 1], np.full(400, 5.5), atol=1e-12)


# ---------------------------------------------------------------- tests/testthat/test-tree.R:4-104
def test_binarize_tree_kats(oracle):
    bt = oracle.binarize_tree
    assert bt([1, 2, 3]) == [[1, 2], 3]
    assert bt([1, 2, 3, 4, 5]) == [[[[1, 2], 3], 4], 5]
    assert bt([[1, 2, 3]]) == [[1, 2], 3]
    assert bt([[1], [2]]) == [1, 2]
    assert bt([[1, 2, 3], [4, 5, 6]]) == [[[1, 2], 3], [[4, 5], 6]]
    assert bt([[np.arange(1, 4)], [np.arange(4, 7)]]) == [[[1, 2], 3], [[4, 5], 6]]
    ref = [[[1, 2], [3, 4]], [[5, 6], [7, 8]]]
    assert bt(ref) == ref
    with pytest.raises(ValueError, match="node with no children"):
        bt([[], [1, 2, 3], [4, 5, 6]])


def test_create_tree_predefined_kats(oracle):
    B = [np.full((1, 1), float(i)) for i in (1, 2, 3, 4)]
    out = oracle.create_tree_predefined(B[:3], [np.array([10]), np.array([20]), np.array([30])], [1, 2, 3])
    assert out[0][0].data is B[0] and out[0][0].restrict[0] == 10
    assert out[0][1].data is B[1] and out[0][1].restrict[0] == 20
    assert out[1].data is B[2] and out[1].restrict[0] == 30
    out = oracle.create_tree_predefined(B[:3], [np.array([10]), np.array([20]), np.array([30])], [3, 2, 1])
    assert out[0][0].data is B[2] and out[0][1].data is B[1] and out[1].data is B[0]
    out = oracle.create_tree_predefined(B, None, [[1, 4], [3, 2]])
    assert out[0][0].data is B[0] and out[0][1].data is B[3] and out[1][0].data is B[2] and out[1][1].data is B[1]
    assert out[0][0].restrict is None
    # character leaves
    a = oracle.resolve_merge_tree(3, [1, 2, 3])
    b = oracle.resolve_merge_tree(3, ["A", "B", "C"], names=["A", "B", "C"])
    assert a == b
    assert oracle.resolve_merge_tree(4, [["a", "d"], ["c", "b"]], names=list("abcd")) == [[1, 4], [3, 2]]
    for bad, names in (([1, 2, 3], None), ([1, 1], None), (["A", "C"], ["A", "B"])):
        with pytest.raises(ValueError, match="invalid leaf nodes"):
            oracle.resolve_merge_tree(2, bad, names)


def test_next_merge_order(oracle):
    # tests/testthat/test-fast-mnn.R:365-366: list(list(4,3),list(1,2)) merges (1,2) first
    B = [np.zeros((2, 1)) for _ in range(4)]
    tree = oracle.create_tree_predefined(B, None, [[4, 3], [1, 2]])
    l, r, path = oracle.get_next_merge(tree)
    assert l.index == [1] and r.index == [2] and path == (1,)


# ---------------------------------------------------------------- tests/testthat/test-utils.R:117-152
def test_restore_original_order_kat(oracle):
    out = oracle.restore_original_order([2, 1, 3], [10, 20, 30])
    expect = np.concatenate([np.arange(21, 31), np.arange(1, 21), np.arange(31, 61)])
    assert np.array_equal(out, expect)
    rng = np.random.default_rng(1000010)
    original = [rng.random(n) for n in (35, 13, 23, 2, 42)]
    s = [5, 3, 1, 4, 2]
    shuffled = [original[i - 1] for i in s]
    out = oracle.restore_original_order(s, [len(x) for x in original])
    assert np.array_equal(np.concatenate(shuffled)[out - 1], np.concatenate(original))
    with pytest.raises(ValueError, match="not equal"):
        oracle.restore_original_order([], [1])
    assert oracle.restore_original_order([], []).size == 0


def test_reindex_pairings_property(oracle):
    rng = np.random.default_rng(1000011)
    S = rng.permutation(40) + 1
    pairings = [(rng.integers(1, 11, 20), np.arange(11, 31)), (np.arange(30, 0, -1), rng.integers(33, 41, 30))]
    out = oracle.reindex_pairings(pairings, S)
    for (ol, orr), (pl, pr) in zip(out, pairings):
        assert np.array_equal(S[ol - 1], pl) and np.array_equal(S[orr - 1], pr)


# ---------------------------------------------------------------- tests/testthat/test-fast-mnn.R:7-32
def test_average_correction_spec(oracle):
    rng = np.random.default_rng(1200001)
    t1 = rng.standard_normal((100, 10))
    t2 = rng.standard_normal((200, 10))
    mnn1 = rng.integers(1, 101, 250)
    mnn2 = rng.integers(1, 101, 250)
    correct = t1[mnn1 - 1] - t2[mnn2 - 1]
    groups = sorted(set(mnn2.tolist()))
    ref = np.vstack([correct[mnn2 == g].mean(axis=0) for g in groups])
    avg, second = oracle.average_correction(t1, mnn1, t2, mnn2)
    np.testing.assert_allclose(avg, ref, rtol=1e-12)
    assert second.tolist() == groups
    avg, second = oracle.average_correction(t1, np.zeros(0, int), t2, np.zeros(0, int))
    assert avg.shape == (0, 10) and second.size == 0


# ---------------------------------------------------------------- tests/testthat/test-fast-mnn.R:35-51
def test_center_along_batch_vector_spec(oracle):
    rng = np.random.default_rng(1200002)
    test = rng.standard_normal((100, 10))
    batch = rng.standard_normal(10)
    centered = oracle.center_along_batch_vector(test, batch)
    assert np.std(centered @ batch, ddof=1) < 1e-8
    test2 = np.vstack([test, test[:10]])
    keep = np.arange(1, 101)
    cur = oracle.center_along_batch_vector(test2, batch, restrict=keep)
    assert np.array_equal(centered, cur[:100])


# ---------------------------------------------------------------- tests/testthat/test-fast-mnn.R:54-92
@pytest.mark.parametrize("k,ndist", [(20, 3), (11, 3), (11, 1)])
def test_tricube_weighted_correction_spec(oracle, k, ndist):
    rng = np.random.default_rng(1200003)
    test = rng.standard_normal((100, 10))
    correction = rng.standard_normal((50, 10))
    involved = rng.permutation(100)[:50] + 1

    cur_uniq = test[involved - 1]
    safe_k = min(k, 50)
    # independent exact kNN
    d2 = ((test[:, None, :] - cur_uniq[None, :, :]) ** 2).sum(-1)
    order = np.argsort(d2, axis=1, kind="stable")[:, :safe_k]
    dist = np.sqrt(np.take_along_axis(d2, order, axis=1))
    middle = int(math.ceil(safe_k / 2))
    ref = test.copy()
    for x in range(100):
        ad = dist[x]
        mid = np.sort(ad)[middle - 1]
        with np.errstate(invalid="ignore", divide="ignore"):
            w = (1 - np.minimum(1, ad / (mid * ndist)) ** 3) ** 3
            w = w / w.sum()
        ref[x] = test[x] + (correction[order[x]] * w[:, None]).sum(axis=0)
    out = oracle.tricube_weighted_correction(test, correction, involved, k=k, ndist=ndist)
    np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-12, equal_nan=True)


# ---------------------------------------------------------------- tests/testthat/test-utils.R:82-115
def test_compute_tricube_average_kats(oracle):
    rng = np.random.default_rng(1000009)
    A = rng.random((50, 20))
    idx, dist = oracle.query_knn(A, A, 11)
    idx, dist = idx[:, 1:], dist[:, 1:]  # findKNN excludes self
    out = oracle.compute_tricube_average(A, idx, dist)
    assert out.shape == A.shape
    out = oracle.compute_tricube_average(A, idx[:, :1], dist[:, :1])
    assert np.array_equal(out, A[idx[:, 0] - 1])
    out = oracle.compute_tricube_average(A, idx, np.ones_like(dist))
    np.testing.assert_allclose(out, np.vstack([A[idx[i] - 1].mean(axis=0) for i in range(50)]), rtol=1e-12)
    uni = np.tile(np.arange(1, 51)[:, None], (1, idx.shape[1]))
    np.testing.assert_allclose(oracle.compute_tricube_average(A, uni, dist), A, rtol=1e-12)
    out = oracle.compute_tricube_average(A, idx[:, :0], dist[:, :0])
    assert out.shape == A.shape and np.all(out == 0)
    assert oracle.compute_tricube_average(A[:0], idx[:0], dist[:0]).shape == (0, 20)


# ---------------------------------------------------------------- exact kNN contract, independent cross-checks
def test_knn_matches_independent_searches(oracle):
    from scipy.spatial import cKDTree
    X, Q = synth_batches(1, [3000, 1700], 50)
    idx, dist = oracle.query_knn(X, Q, 20)
    dd, ii = cKDTree(X).query(Q, k=20)
    assert np.array_equal(idx - 1, ii)
    np.testing.assert_allclose(dist, dd, rtol=1e-12)
    # bitwise: distance = sqrt of the left-to-right sum over dims
    q, r = Q[5], X[idx[5, 3] - 1]
    s = 0.0
    for c in range(50):
        t = q[c] - r[c]
        s += t * t
    assert dist[5, 3] == math.sqrt(s)


def test_knn_ties_lowest_index_and_small_inputs(oracle):
    X = np.array([[0.0, 0], [1, 0], [1, 0], [0, 1], [1, 0]])
    idx, dist = oracle.query_knn(X, np.array([[1.0, 0.0], [0, 0]]), 3)
    assert idx.tolist() == [[2, 3, 5], [1, 2, 3]]
    idx, _ = oracle.query_knn(X, X[:2], 10)  # k > n is clamped like safe.k
    assert idx.shape == (2, 5)
    for nt in (1, 3):
        a, _ = oracle.query_knn(X, X, 2, nthreads=nt)
        assert a[:, 0].tolist() == [1, 2, 2, 4, 2]


# ---------------------------------------------------------------- src/find_mutual_nns.cpp:8-41
def test_find_mutual_nns_order(oracle):
    left = np.array([[1, 2], [2, 3], [3, 1]])
    right = np.array([[1, 3], [1, 2], [3, 2]])
    f, s = oracle.find_mutual_nns(left, right)
    assert list(zip(f.tolist(), s.tolist())) == [(1, 1), (1, 2), (2, 2), (2, 3), (3, 3), (3, 1)]
    rng = np.random.default_rng(7)
    nL, nR, k1, k2 = 40, 30, 5, 7
    L = np.vstack([rng.permutation(nR)[:k2] + 1 for _ in range(nL)])
    R = np.vstack([rng.permutation(nL)[:k1] + 1 for _ in range(nR)])
    f, s = oracle.find_mutual_nns(L, R)
    exp = [(l + 1, int(r)) for l in range(nL) for r in L[l] if (l + 1) in R[r - 1]]
    assert list(zip(f.tolist(), s.tolist())) == exp


# ---------------------------------------------------------------- tests/testthat/test-mnn-correct.R:28-92
def _smooth_ref(data1, data2, mnn1, mnn2, s2):
    d2 = ((data2[:, None, :] - data2[None, :, :]) ** 2).sum(-1)
    w = np.exp(-d2 / s2)
    uniq = np.unique(mnn2)
    dens = w[:, uniq - 1].sum(axis=1)
    N = np.bincount(mnn2, minlength=data2.shape[0] + 1)[1:]
    with np.errstate(divide="ignore", invalid="ignore"):
        kernel = (w / (N * dens)[:, None]).T[:, mnn2 - 1]
    kernel = kernel / kernel.sum(axis=1)[:, None]
    return kernel @ (data1[mnn1 - 1] - data2[mnn2 - 1])


def _compute_correction_vectors(oracle, data1, data2, mnn1, mnn2, s2):
    """R/mnnCorrect.R:451-460."""
    avg, ids = oracle.average_correction(data1, mnn1, data2, mnn2)
    return oracle.smooth_gaussian_kernel(avg.T, ids - 1, data2.T, s2).T


@pytest.mark.parametrize("case", ["vanilla", "repeats", "many", "bandwidth"])
def test_smooth_gaussian_kernel_spec(oracle, case):
    rng = np.random.default_rng(10003)
    data1 = rng.standard_normal((400, 25)) * 0.1
    data2 = rng.standard_normal((1000, 25)) * 0.1
    mnn1, mnn2, s2 = np.arange(1, 11), np.arange(30, 20, -1), 0.1
    if case == "repeats":
        mnn1, mnn2 = np.concatenate([[11, 12, 13], mnn1]), np.concatenate([[30, 30, 30], mnn2])
    elif case == "many":
        mnn1, mnn2 = np.arange(1, 201), np.arange(500, 300, -1)
    elif case == "bandwidth":
        s2 = 0.5
    out = _compute_correction_vectors(oracle, data1, data2, mnn1, mnn2, s2)
    np.testing.assert_allclose(out, _smooth_ref(data1, data2, mnn1, mnn2, s2), rtol=1e-8, atol=1e-14)


def test_smooth_gaussian_kernel_errors(oracle):
    with pytest.raises(RuntimeError, match="'index' must have length"):
        oracle.smooth_gaussian_kernel(np.zeros((3, 4)), np.zeros(3, int), np.zeros((3, 5)), 1.0)


# ---------------------------------------------------------------- tests/testthat/test-mnn-correct.R:94-174
def _asv_ref(data1, data2, cell_vect, sigma):
    d1, d2 = data1.T, data2.T
    out = np.zeros(cell_vect.shape[0])
    for cell in range(out.size):
        v = cell_vect[cell]
        l2 = math.sqrt(float(np.sum(v ** 2)))
        v = v / l2
        c2, c1 = d2 @ v, d1 @ v
        diff2 = d2[cell][None, :] - d2
        diff2 = diff2 - np.outer(diff2 @ v, v)
        w2 = np.exp(-(diff2 ** 2).sum(axis=1) / sigma)
        diff1 = d2[cell][None, :] - d1
        diff1 = diff1 - np.outer(diff1 @ v, v)
        w1 = np.exp(-(diff1 ** 2).sum(axis=1) / sigma)
        rank2 = np.empty(c2.size, dtype=int)
        rank2[np.argsort(c2, kind="stable")] = np.arange(c2.size)
        prob2 = w2[rank2 <= rank2[cell]].sum() / w2.sum()
        ord1 = np.argsort(c1, kind="stable")
        ecdf1 = np.cumsum(w1[ord1]) / w1.sum()
        hit = np.flatnonzero(ecdf1 >= prob2)  # empty only when prob2 == 1 and the cumsum rounds below 1
        quan1 = c1[ord1[hit.min() if hit.size else -1]]
        out[cell] = (quan1 - c2[cell]) / l2
    return out


def _asv_reference_shape(seed=100032):
    # the reference's own test shape (test-mnn-correct.R:96-98): data1 25 x 400, data2 25 x 1000 (rnorm, sd 0.1),
    # corvect 1000 x 25 (runif) -- drawn with our RNG, R's stream is not reproducible here
    rng = np.random.default_rng(seed)
    data1 = rng.standard_normal((25, 400)) * 0.1
    data2 = rng.standard_normal((25, 1000)) * 0.1
    corvect = rng.random((1000, 25))
    return data1, data2, corvect


@pytest.mark.parametrize("sigma", [1.0, 0.1])
def test_adjust_shift_variance_spec(oracle, sigma):
    # test-mnn-correct.R:94-151 at ITS shape: the oracle equals the reference's REF on EVERY cell -- what the reference's
    # expect_equal(ref, test) (tolerance 1.5e-8) asserts (:143-149)
    data1, data2, corvect = _asv_reference_shape()
    ref = _asv_ref(data1, data2, corvect, sigma)
    out = oracle.adjust_shift_variance(data1, data2, corvect, sigma, np.arange(400), np.arange(1000))
    np.testing.assert_allclose(out, ref, rtol=1.5e-8, atol=1e-12)
    assert np.abs(out / ref - 1.0).mean() < 1e-13  # measured 6e-15 .. 2e-14


@pytest.mark.parametrize("sigma", [1.0, 0.1])
def test_adjust_shift_variance_spec_swapped_shape(oracle, sigma):
    # the same with the larger batch as the reference batch (25 x 1000 against 25 x 400)
    data2, data1, _ = _asv_reference_shape(100034)
    corvect = np.random.default_rng(5).random((400, 25))
    ref = _asv_ref(data1, data2, corvect, sigma)
    out = oracle.adjust_shift_variance(data1, data2, corvect, sigma, np.arange(1000), np.arange(400))
    np.testing.assert_allclose(out, ref, rtol=1.5e-8, atol=1e-12)


# ---------------------------------------------------------------- R::logspace_add, src/adjust_shift_variance.cpp:99,107,130,150
def _logspace_many(oracle, lx, ly, libm):
    import ctypes
    f64p = ctypes.POINTER(ctypes.c_double)
    out = np.empty_like(lx)
    oracle.lib().orc_logspace_add_many(lx.ctypes.data_as(f64p), ly.ctypes.data_as(f64p), ctypes.c_int64(lx.size), int(libm),
                                       out.ctypes.data_as(f64p))
    return out


LOGSPACE_DRAWS = {
    # name: (draw, measured share of pairs on which the portable sum and glibc's differ at all)
    "U(-50,0)": (lambda r, n: r.uniform(-50, 0, n), 0.0026),
    "3*N(0,1)": (lambda r, n: 3.0 * r.standard_normal(n), 0.081),
    "U(-5,0)": (lambda r, n: r.uniform(-5, 0, n), 0.155),
}


@pytest.mark.parametrize("name", list(LOGSPACE_DRAWS))
def test_logspace_add_portable_against_libm(oracle, name, capsys):
    """The oracle (and the HIP side, csrc/portable_math.hpp) sum with a bit-reproducible exp / log1p; Rmath's logspace_add
    calls the platform's.  How far apart the two are, on a million random pairs per distribution: never more than 3 units in
    the last place of max(|result|, |larger operand|, 0.5), and they differ at all on 0.3 % .. 16 % of the pairs depending on
    how close the operands are (the closer, the more of log1p's argument range is exercised)."""
    draw, measured = LOGSPACE_DRAWS[name]
    rng = np.random.default_rng(20250601)
    lx, ly = draw(rng, 10 ** 6), draw(rng, 10 ** 6)
    a, b = _logspace_many(oracle, lx, ly, False), _logspace_many(oracle, lx, ly, True)
    scale = np.spacing(np.maximum(np.maximum(np.abs(a), np.abs(np.maximum(lx, ly))), 0.5))
    ulps = np.abs(a - b) / scale
    share = float((a != b).mean())
    with capsys.disabled():
        print(f"\n[logspace_add {name}] portable != libm on {share:.4f} of 1e6 pairs, max {ulps.max():.1f} ulp")
    assert ulps.max() <= 4.0, ulps.max()
    assert share <= 2.5 * measured + 0.01, share
    # and both are the true sum to within a few ulp: against extended precision
    ref = np.logaddexp(lx.astype(np.longdouble), ly.astype(np.longdouble))
    assert (np.abs(a.astype(np.longdouble) - ref) / scale).max() <= 4.0


@pytest.mark.parametrize("shape", ["reference", "100 dimensions"])
@pytest.mark.parametrize("sigma", [1.0, 0.1, 0.01])
def test_adjust_shift_variance_math_library_moves_no_cell(oracle, shape, sigma, capsys):
    """adjust_shift_variance summed with the platform's exp / log1p (orc_set_logspace_libm(1): what an R linked against this
    libc computes) against the oracle's portable arithmetic, on the reference's own test shape (test-mnn-correct.R:96-98) and
    on a 100-dimension shape where the walk is decided by the last bit: the share of cells that land on
    another quantile.  Measured: none on either shape at any bandwidth -- the chains' decisions are made by which addends are
    absorbed, which a <= 3 ulp difference in log1p(exp(.)) of the absorbed addend does not change."""
    if shape == "reference":
        data1, data2, corvect = _asv_reference_shape()
    else:
        rng = np.random.default_rng(7)
        data1, data2, corvect = rng.standard_normal((100, 600)), rng.standard_normal((100, 800)), rng.random((800, 100))
    r1, r2 = np.arange(data1.shape[1]), np.arange(data2.shape[1])
    out = oracle.adjust_shift_variance(data1, data2, corvect, sigma, r1, r2)
    oracle.lib().orc_set_logspace_libm(1)
    try:
        out_libm = oracle.adjust_shift_variance(data1, data2, corvect, sigma, r1, r2)
    finally:
        oracle.lib().orc_set_logspace_libm(0)
    moved = float((out != out_libm).mean())
    with capsys.disabled():
        print(f"\n[adjust_shift_variance {shape}, sigma {sigma}] cells moved by the math library: {moved:.4f}")
    assert moved <= 0.01, moved


def test_adjust_shift_variance_restrict_identity(oracle):
    # tests/testthat/test-mnn-correct.R:160-173 (expect_identical)
    rng = np.random.default_rng(100033)
    data1 = rng.standard_normal((25, 60)) * 0.1
    data2 = rng.standard_normal((25, 80)) * 0.1
    corvect = rng.random((80, 25))
    i1 = np.arange(9, 20)
    i2 = np.arange(19, 8, -1)
    A1 = np.hstack([data1, data1[:, i1]])
    A2 = np.hstack([data2, data2[:, i2]])
    t1 = oracle.adjust_shift_variance(data1, data2, corvect, 1.0, np.arange(60), np.arange(80))
    t2 = oracle.adjust_shift_variance(A1, A2, np.vstack([corvect, corvect[i2]]), 1.0, np.arange(60), np.arange(80))
    assert np.array_equal(t1, t2[:80])
    assert np.array_equal(t1[i2], t2[80:])


def test_adjust_shift_variance_errors(oracle):
    z = np.zeros
    with pytest.raises(RuntimeError, match="number of genes do not match"):
        oracle.adjust_shift_variance(z((3, 4)), z((2, 5)), z((5, 3)), 1.0, [0], [0])
    with pytest.raises(RuntimeError, match="number of cells do not match"):
        oracle.adjust_shift_variance(z((3, 4)), z((3, 5)), z((4, 3)), 1.0, [0], [0])
    with pytest.raises(RuntimeError, match="subset indices out of range"):
        oracle.adjust_shift_variance(z((3, 4)), z((3, 5)), z((5, 3)), 1.0, [4], [0])


# ---------------------------------------------------------------- engine-level properties of the reference tests
def _offset_batches(seed, sizes, d=10):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((n, d)) + i for i, n in enumerate(sizes)]


def test_prop_k_identities(oracle):
    # tests/testthat/test-reduced-mnn.R:39-58
    B1, B2 = _offset_batches(12000052, [100, 100], d=100)
    ref = oracle.reduced_mnn(B1, B2)
    out = oracle.reduced_mnn(B1, B2, k=10, prop_k=20 / 100)
    assert np.array_equal(ref.corrected, out.corrected)
    out = oracle.reduced_mnn(B1, B2, prop_k=0)
    assert np.array_equal(ref.corrected, out.corrected)
    B2a = _offset_batches(5, [100, 200], d=100)[1]
    ref = oracle.reduced_mnn(B1, B2a)
    out = oracle.reduced_mnn(B1, B2a, prop_k=20 / 100)
    assert not np.array_equal(ref.corrected, out.corrected)


def test_merge_order_metamorphic(oracle):
    # tests/testthat/test-fast-mnn.R:267-310: re-ordered inputs + matching merge.order give identical output rows
    B = _offset_batches(1200007, [120, 150, 90])
    ref = oracle.reduced_mnn(B[0], B[1], B[2])
    out = oracle.reduced_mnn(B[2], B[1], B[0], merge_order=[3, 2, 1])
    back = np.concatenate([np.arange(240, 360), np.arange(90, 240), np.arange(0, 90)])
    np.testing.assert_array_equal(ref.corrected, out.corrected[back])
    assert out.batch.tolist() == [1] * 90 + [2] * 150 + [3] * 120
    assert out.merge_info.left[0] == [3] and out.merge_info.right[0] == [2]
    # pairs refer to output rows of the right batches
    for (l, r), lset, rset in zip(out.merge_info.pairs, out.merge_info.left, out.merge_info.right):
        assert l.size > 0 and np.all(np.isin(out.batch[l - 1], lset)) and np.all(np.isin(out.batch[r - 1], rset))


def test_restriction_identity(oracle):
    # tests/testthat/test-reduced-mnn.R:107-133 (expect_identical on restricted vs duplicated cells)
    B1, B2, B3 = _offset_batches(12000053, [100, 200, 50])
    ref = oracle.reduced_mnn(B1, B2, B3)
    i1, i2, i3 = np.arange(99, 48, -1), np.arange(0, 20), np.arange(49, 50)
    C1, C2, C3 = np.vstack([B1, B1[i1]]), np.vstack([B2, B2[i2]]), np.vstack([B3, B3[i3]])
    keep = [np.arange(1, 101), np.arange(1, 201), np.arange(1, 51)]
    out = oracle.reduced_mnn(C1, C2, C3, restrict=keep)
    for b, (n, ii) in enumerate(((100, i1), (200, i2), (50, i3)), start=1):
        r = ref.corrected[ref.batch == b]
        o = out.corrected[out.batch == b]
        assert np.array_equal(r, o[:n])
        assert np.array_equal(r[ii], o[n:])


def test_batch_size_skip_and_lost_var(oracle):
    # tests/testthat/test-fast-mnn.R:199-224, 409-457
    B1, B2 = _offset_batches(1200010, [300, 400], d=20)
    out = oracle.reduced_mnn(B1, B2)
    assert out.merge_info.batch_size[0] > 0.5 and not out.merge_info.skipped[0]
    assert np.all(out.merge_info.lost_var > 0)
    rng = np.random.default_rng(3)
    A1, A2 = rng.standard_normal((300, 20)), rng.standard_normal((400, 20))
    out = oracle.reduced_mnn(A1, A2)
    assert out.merge_info.batch_size[0] < 0.1
    out = oracle.reduced_mnn(A1, A2, min_batch_skip=0.1)
    assert out.merge_info.skipped[0] and np.all(out.merge_info.lost_var == 0)
    assert np.array_equal(out.corrected, np.vstack([A1, A2]))
    out = oracle.reduced_mnn(A1, A2, min_batch_skip=None)
    assert np.isnan(out.merge_info.batch_size[0]) and not out.merge_info.skipped[0]


def test_auto_merge_equals_explicit_order(oracle):
    # tests/testthat/test-fast-mnn.R:312-335: auto.merge == merge.order implied by the MNN counts
    B = _offset_batches(1200012, [150, 220, 130, 90])
    auto = oracle.reduced_mnn(*B, auto_merge=True)
    order = [auto.merge_info.left[0][0], auto.merge_info.right[0][0]]
    for m in range(1, 3):
        new = [x for x in auto.merge_info.left[m] + auto.merge_info.right[m] if x not in order]
        order += new
    # the first chosen pair has the most MNN pairs of all pairs
    counts = {}
    for i in range(4):
        for j in range(i):
            f, _ = oracle.restricted_mnn(B[i], None, B[j], None, 20)
            counts[(i + 1, j + 1)] = f.size
    best = max(counts.values())
    assert counts[(auto.merge_info.left[0][0], auto.merge_info.right[0][0])] == best
    assert sorted(order) == [1, 2, 3, 4]
    assert auto.corrected.shape == (590, 10) and auto.batch.tolist() == sum(([b + 1] * n for b, n in enumerate([150, 220, 130, 90])), [])


def test_single_object_with_batch_factor(oracle):
    # tests/testthat/test-reduced-mnn.R:60-78
    B = _offset_batches(120000521, [200, 400, 300], d=20)
    com = np.vstack(B)
    batches = np.repeat([1, 2, 3], [200, 400, 300])
    rng = np.random.default_rng(0)
    shuffle = rng.permutation(900)
    out = oracle.reduced_mnn(com[shuffle], batch=batches[shuffle])
    ref = oracle.reduced_mnn(*B)
    np.testing.assert_allclose(ref.corrected[shuffle], out.corrected, rtol=1e-9, atol=1e-12)  # expect_equal upstream
    assert np.array_equal(ref.batch[shuffle], out.batch)


def test_cpu_baselines_are_exact(oracle):
    """bench.py's two CPU baselines (oracle/cpu_baselines.py: KMKNN-style pruned search, BLAS brute force) give the
    brute-force oracle's neighbours: what is timed there is the same exact search."""
    from oracle import cpu_baselines as cb
    X, Q = synth_batches(2, [3000, 200], 20)
    oi, od = oracle.query_knn(X, Q, 12)
    ia, da, st = cb.kmknn_knn(X, Q, 12)
    assert np.array_equal(ia, oi) and np.array_equal(da, od)
    assert 0.0 < st["visited"] <= 1.0
    ib, db, _ = cb.blas_knn(X, Q, 12)
    assert np.array_equal(ib, oi)
    np.testing.assert_allclose(db, od, rtol=1e-12)


def test_cpu_baseline_b_filtered_brute_force_is_an_exact_knn():
    """bench.py's CPU baseline B (oracle/cpu_baselines.py: DGEMM tiles + a fused running-threshold filter + exact
    re-evaluation of the kept) against the oracle's brute force: same neighbours in the same order, same distances to
    rounding -- the number quoted as `cpu_baseline` is the time of a search that really is the reference's search."""
    from oracle import cpu_baselines as cb
    from oracle import fastmnn_oracle as orc
    rng = np.random.default_rng(11)
    X = rng.standard_normal((9000, 30)) / np.sqrt(1.0 + np.arange(30) / 5.0)
    Q = rng.standard_normal((700, 30)) / np.sqrt(1.0 + np.arange(30) / 5.0) + 0.2
    for chunk in (2048, 0):          # the filtered form (several chunks) and the one-partition form
        idx, dist, info = cb.blas_knn(X, Q, 20, block=128, workers=2, chunk=chunk)
        oi, od = orc.query_knn(X, Q, 20)
        assert np.array_equal(idx, oi)
        np.testing.assert_allclose(dist, od, rtol=1e-12)
        assert info["workers"] == 2


@pytest.mark.parametrize("nx,nq,d,k", [(9000, 700, 30, 20), (5003, 257, 50, 20), (300, 131, 7, 36), (64, 10, 3, 5)])
def test_cpu_baseline_t_tiled_brute_force_is_an_exact_knn(oracle, nx, nq, d, k):
    """bench.py's CPU baseline T (oracle/tiled_knn_baseline.c: packed operands, a 4 x 8 AVX2 + FMA micro-kernel with the
    threshold filter on the tile in registers, all cores) against the oracle's brute force: the same neighbours in the same
    order, distances bitwise (the kept candidates are re-evaluated in the oracle's order of operations).  Ragged sizes: rows
    that do not fill an octet, queries that do not fill a group of four."""
    from oracle import cpu_baselines as cb
    rng = np.random.default_rng(12)
    X = rng.standard_normal((nx, d)) / np.sqrt(1.0 + np.arange(d) / 5.0)
    Q = rng.standard_normal((nq, d)) / np.sqrt(1.0 + np.arange(d) / 5.0) + 0.2
    idx, dist = cb.tiled_knn(X, Q, k, nthreads=3)
    oi, od = oracle.query_knn(X, Q, k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
