"""GPU parity of the steps upstream of the engine (device cosineNorm + fused projection, host Gram PCA) and of the
fastMNN() front-end against the oracle (oracle/pca_oracle.py, which uses a direct SVD)."""
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
    mine = bx.multiBatchPCA(t1, t2, t3, d=20)
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
        bx.multiBatchPCA(t1, t2[:0])


def test_fast_mnn_front_end(bx, pca):
    # tests/testthat/test-fast-mnn.R:124-150 shape, at a size where the path is exercised (genes x cells)
    rng = np.random.default_rng(1200004)
    B1 = rng.standard_normal((300, 2000))
    B2 = rng.standard_normal((300, 2500)) + 1
    B3 = rng.standard_normal((300, 1500)) + 2
    out = bx.fastMNN(B1, B2, B3, d=30)
    ref, meta = pca.fast_mnn(B1, B2, B3, d=30)
    sgn = np.sign((out.rotation * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(out.rotation * sgn[None, :], meta["rotation"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(out.corrected * sgn[None, :], ref.corrected, rtol=1e-5, atol=1e-8)
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert np.array_equal(ol, rl) and np.array_equal(orr, rr)
    assert list(out.batch) == list(ref.batch)
    with pytest.raises(ValueError, match="at least two batches"):
        bx.fastMNN(B1)
