"""Pins oracle/pca_oracle.py (cosineNorm, multiBatchPCA) against the reference's own tests, re-expressed with our RNG.
Citations relative to /root/reference."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def pca():
    from oracle import pca_oracle
    return pca_oracle


def besides_sign(left, right):
    """tests/testthat/test-multi-pca.R:6-10."""
    ratio = left[0] / right[0]
    return right * np.sign(ratio)[None, :]


# ---------------------------------------------------------------- tests/testthat/test-cos-norm.R:5-47
def test_cosine_norm_spec(pca):
    rng = np.random.default_rng(10001)
    X = rng.standard_normal((100, 100))
    cellnorm = np.sqrt((X ** 2).sum(axis=0))
    np.testing.assert_allclose(pca.cosine_norm(X), X / cellnorm[None, :], rtol=1e-14)
    X = rng.poisson(5, (20, 1000)).astype(float)
    cellnorm = np.sqrt((X ** 2).sum(axis=0))
    out = pca.cosine_norm(X, mode="all")
    np.testing.assert_allclose(out["matrix"], X / cellnorm[None, :], rtol=1e-14)
    np.testing.assert_allclose(out["l2norm"], cellnorm, rtol=1e-14)
    np.testing.assert_allclose(pca.cosine_norm(X, mode="l2norm"), cellnorm, rtol=1e-14)
    Z = np.zeros((100, 20))
    out = pca.cosine_norm(Z, mode="all")
    assert np.array_equal(out["matrix"], Z) and np.array_equal(out["l2norm"], np.zeros(20))


# ---------------------------------------------------------------- tests/testthat/test-multi-pca.R:13-56
def test_multi_batch_pca_invariances(pca):
    rng = np.random.default_rng(1200001)
    t1 = rng.standard_normal((10, 100))
    t2 = rng.standard_normal((10, 200))
    ref, _ = pca.multi_batch_pca([t1, t2], d=5)
    out, _ = pca.multi_batch_pca([t1, np.hstack([t2, t2])], d=5)
    np.testing.assert_allclose(ref[0], besides_sign(ref[0], out[0]), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(np.vstack([ref[1], ref[1]]), besides_sign(np.vstack([ref[1], ref[1]]), out[1]),
                               rtol=1e-8, atol=1e-10)
    assert ref[0].shape == (100, 5) and ref[1].shape == (200, 5)
    # equal sizes: equivalent to a PCA of the cbind (prcomp: centre by the column means, right singular basis)
    t3 = rng.standard_normal((10, 100))
    out, _ = pca.multi_batch_pca([t1, t3], d=4)
    both = np.hstack([t1, t3]).T
    both = both - both.mean(axis=0)
    _, _, vt = np.linalg.svd(both, full_matrices=False)
    pr = both @ vt[:4].T
    mine = np.vstack(out)
    np.testing.assert_allclose(mine, besides_sign(mine, pr), rtol=1e-8, atol=1e-10)
    # full rank preserves all pairwise distances
    out, _ = pca.multi_batch_pca([t1, t2], d=10)
    every = np.vstack([t1.T, t2.T])
    got = np.vstack(out)
    dref = np.linalg.norm(every[:, None] - every[None], axis=-1)
    dgot = np.linalg.norm(got[:, None] - got[None], axis=-1)
    np.testing.assert_allclose(dgot, dref, rtol=1e-9, atol=1e-10)
    with pytest.raises(ValueError, match="not the same"):
        pca.multi_batch_pca([t1, t2[:0]])


def test_multi_batch_pca_projection_identity_and_variance(pca):
    # tests/testthat/test-multi-pca.R:97-105, :237-265
    rng = np.random.default_rng(1200002)
    t1, t2, t3 = rng.standard_normal((20, 50)), rng.standard_normal((20, 100)), rng.standard_normal((20, 150))
    out, meta = pca.multi_batch_pca([t1, t2, t3], d=20, get_variance=True)
    for t, o in zip((t1, t2, t3), out):
        np.testing.assert_allclose(o, (t - meta["centers"][:, None]).T @ meta["rotation"], rtol=1e-12, atol=1e-12)
    assert abs(meta["var.explained"].sum() / meta["var.total"] - 1) < 1e-10
    centre = (t1.mean(1) + t2.mean(1) + t3.mean(1)) / 3
    manual = sum(((t - centre[:, None]) ** 2).mean(axis=1).sum() for t in (t1, t2, t3)) / 3
    assert abs(manual / meta["var.total"] - 1) < 1e-10
    alt, ameta = pca.multi_batch_pca([t1, t2, t3], d=10, get_variance=True)
    np.testing.assert_allclose(meta["var.explained"][:10], ameta["var.explained"], rtol=1e-10)
    for i in range(10):
        stuff = [x[:, i] for x in alt]
        assert abs(sum(s.mean() for s in stuff)) < 1e-10
        assert abs(np.mean([np.mean(s ** 2) for s in stuff]) / meta["var.explained"][i] - 1) < 1e-9


def test_weights(pca):
    # R/multiBatchPCA.R:299-352
    assert pca.construct_weight_vector([10, 20, 30], None).tolist() == [1, 1, 1]
    assert pca.construct_weight_vector([10, 20, 30], False).tolist() == [10, 20, 30]
    assert pca.construct_weight_vector([10, 20, 30], [1, 2, 3]).tolist() == [1, 2, 3]
    np.testing.assert_allclose(pca.construct_weight_vector([10, 20, 30], [[1, 2], 3]), [0.25, 0.25, 0.5])
    with pytest.raises(ValueError, match="invalid integer indices"):
        pca.construct_weight_vector([10, 20, 30], [[1, 2], 4])


def test_fast_mnn_front_end_runs(pca):
    # tests/testthat/test-fast-mnn.R:124-150 shape: offsets so that the correction is not skipped
    rng = np.random.default_rng(1200004)
    B1 = rng.standard_normal((100, 300))
    B2 = rng.standard_normal((100, 400)) + 1
    out, meta = pca.fast_mnn(B1, B2, d=20)
    assert out.corrected.shape == (700, 20) and meta["rotation"].shape == (100, 20)
    assert out.merge_info.pairs[0][0].size > 0
