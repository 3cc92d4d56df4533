"""GPU parity of the steps upstream of the engine (device cosineNorm, device multiBatchPCA by subspace iteration on the
FP64 matrix cores, fused projection; the host Gram PCA kept as an alternative) and of the fastMNN() front-end against
the oracle (oracle/pca_oracle.py, which uses a direct SVD)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bx():
    import batchelor_amd
    return batchelor_amd


@pytest.fixture(scope="module")
def pca():
    from oracle import pca_oracle
    return pca_oracle


def align_sign(mine, ref):
    return mine * np.sign((mine * ref).sum(axis=0))[None, :]


def test_cosine_norm_device(bx, pca):
    # tests/testthat/test-cos-norm.R:5-47
    rng = np.random.default_rng(10001)
    X = rng.standard_normal((100, 100))
    np.testing.assert_allclose(bx.cosineNorm(X), pca.cosine_norm(X), rtol=1e-13)
    X = rng.poisson(5, (20, 1000)).astype(float)
    out = bx.cosineNorm(X, mode="all")
    ref = pca.cosine_norm(X, mode="all")
    np.testing.assert_allclose(out["matrix"], ref["matrix"], rtol=1e-13)
    np.testing.assert_allclose(out["l2norm"], ref["l2norm"], rtol=1e-13)
    np.testing.assert_allclose(bx.cosineNorm(X, mode="l2norm"), ref["l2norm"], rtol=1e-13)
    Z = np.zeros((100, 20))
    out = bx.cosineNorm(Z, mode="all")
    assert np.array_equal(out["matrix"], Z) and np.array_equal(out["l2norm"], np.zeros(20))


def test_host_pca_and_device_projection_match_svd_oracle(bx, pca):
    rng = np.random.default_rng(1200002)
    t1, t2, t3 = rng.standard_normal((200, 500)), rng.standard_normal((200, 900)) + 0.3, rng.standard_normal((200, 300))
    ref, meta = pca.multi_batch_pca([t1, t2, t3], d=20, get_variance=True)
    mine = bx.multiBatchPCA_host(t1, t2, t3, d=20)
    rot = align_sign(mine["rotation"], meta["rotation"])
    np.testing.assert_allclose(rot, meta["rotation"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(mine["centers"], meta["centers"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(mine["d"] ** 2 / 3, meta["var.explained"], rtol=1e-9)
    for t, r in zip((t1, t2, t3), ref):
        got = bx.project(t, rot, mine["centers"], cos_norm=False)
        np.testing.assert_allclose(got, r, rtol=1e-7, atol=1e-9)
    # fused cosine normalisation == normalise first, then project
    got = bx.project(t2, rot, mine["centers"], cos_norm=True)
    np.testing.assert_allclose(got, (pca.cosine_norm(t2) - mine["centers"][:, None]).T @ rot, rtol=1e-9, atol=1e-11)
    with pytest.raises(ValueError, match="not the same"):
        bx.multiBatchPCA_host(t1, t2[:0])


@pytest.mark.parametrize("weights,cos_norm", [(None, False), ([1.0, 3.0, 0.5], False), (None, True)])
def test_device_multi_batch_pca_matches_svd_oracle(bx, pca, weights, cos_norm):
    # low-rank signal + noise (what PCA is run on), three unequal batches with offsets: rotation up to sign, centres,
    # singular values and the projected cells against the direct SVD (R/multiBatchPCA.R:211-322)
    rng = np.random.default_rng(1200007)
    G, r = 333, 12
    load = rng.standard_normal((G, r)) * np.linspace(3.0, 1.0, r)
    mats = [load @ rng.standard_normal((r, n)) + 0.3 * rng.standard_normal((G, n)) + off
            for n, off in ((500, 0.0), (911, 0.4), (300, -0.2))]
    if cos_norm:
        mats = [m + 3.0 for m in mats]          # away from the origin, as normalised expression is
    ref_in = [pca.cosine_norm(m) for m in mats] if cos_norm else mats
    ref, meta = pca.multi_batch_pca(ref_in, d=10, weights=weights, get_variance=True)
    mine = bx.multiBatchPCA(*mats, d=10, weights=weights, cos_norm=cos_norm, iters=40)
    rot = align_sign(mine["rotation"], meta["rotation"])
    sgn = np.sign((mine["rotation"] * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(mine["centers"], meta["centers"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(rot, meta["rotation"], rtol=1e-6, atol=1e-8)
    w = np.ones(3) if weights is None else np.asarray(weights)
    np.testing.assert_allclose(mine["d"] ** 2 / 3, meta["var.explained"], rtol=1e-9)
    for got, want in zip(mine["pcs"], ref):
        np.testing.assert_allclose(got * sgn[None, :], want, rtol=1e-6, atol=1e-8)
    assert w.size == 3
    with pytest.raises(ValueError, match="not the same"):
        bx.multiBatchPCA(mats[0], mats[1][:0])
    with pytest.raises(RuntimeError, match="d <= 56"):
        bx.multiBatchPCA(*mats, d=60)


def test_device_pca_without_a_spectral_gap_converges_with_more_iterations(bx, pca):
    # pure noise: the d-th and the 65th eigenvalue are close, subspace iteration needs many more sweeps
    rng = np.random.default_rng(1200002)
    t1, t2 = rng.standard_normal((200, 900)), rng.standard_normal((200, 700)) + 0.3
    ref, meta = pca.multi_batch_pca([t1, t2], d=8)
    mine = bx.multiBatchPCA(t1, t2, d=8, iters=400)
    rot = align_sign(mine["rotation"], meta["rotation"])
    np.testing.assert_allclose(rot, meta["rotation"], rtol=1e-5, atol=1e-7)


def test_fast_mnn_front_end(bx, pca):
    # tests/testthat/test-fast-mnn.R:124-150 shape, at a size where the path is exercised (genes x cells)
    rng = np.random.default_rng(1200004)
    B1 = rng.standard_normal((300, 2000))
    B2 = rng.standard_normal((300, 2500)) + 1
    B3 = rng.standard_normal((300, 1500)) + 2
    out = bx.fastMNN(B1, B2, B3, d=30, pca="host")
    ref, meta = pca.fast_mnn(B1, B2, B3, d=30)
    sgn = np.sign((out.rotation * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(out.rotation * sgn[None, :], meta["rotation"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(out.corrected * sgn[None, :], ref.corrected, rtol=1e-5, atol=1e-8)
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert np.array_equal(ol, rl) and np.array_equal(orr, rr)
    assert list(out.batch) == list(ref.batch)
    with pytest.raises(ValueError, match="at least two batches"):
        bx.fastMNN(B1)


def test_fast_mnn_front_end_device_pca(bx, pca):
    # the whole front-end on the device: cosine normalisation, multiBatchPCA, projection, merge engine -- on data with
    # structure (5 shared populations + batch offsets), against the oracle's SVD path; pairs bit-exact
    rng = np.random.default_rng(1200011)
    G, r = 400, 8
    load = rng.standard_normal((G, r)) * 2.0
    cent = rng.standard_normal((r, 5)) * 2.0

    def batch(n, off):
        z = cent[:, rng.integers(0, 5, n)] + rng.standard_normal((r, n))
        return np.abs(load @ z + 0.5 * rng.standard_normal((G, n)) + 6.0 + off)
    B = [batch(1500, 0.0), batch(1800, 0.8), batch(1200, -0.5)]
    out = bx.fastMNN(*B, d=8, pca_iters=60)
    ref, meta = pca.fast_mnn(*B, d=8)
    sgn = np.sign((out.rotation * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(out.rotation * sgn[None, :], meta["rotation"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(out.corrected * sgn[None, :], ref.corrected, rtol=1e-5, atol=1e-8)
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert np.array_equal(ol, rl) and np.array_equal(orr, rr)
