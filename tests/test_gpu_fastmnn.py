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
    mine = bx.multiBatchPCA(*mats, d=10, weights=weights, cos_norm=cos_norm)   # default: until the residual is small
    assert mine["path"] == "device" and mine["residual"] <= 1e-9 and 1 <= mine["iters_used"] <= 200
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


def test_device_pca_wide_block_d_above_56(bx, pca):
    # fastMNN(d = 100) (BASELINE.json configs[4]): d > 56 takes the 128-vector block
    rng = np.random.default_rng(1200008)
    G, r = 400, 90
    load = rng.standard_normal((G, r)) * np.linspace(4.0, 1.0, r)
    mats = [load @ rng.standard_normal((r, n)) + 0.2 * rng.standard_normal((G, n)) + off
            for n, off in ((700, 0.0), (500, 0.3))]
    ref, meta = pca.multi_batch_pca(mats, d=80, get_variance=True)
    mine = bx.multiBatchPCA(*mats, d=80)
    assert mine["path"] == "device"
    sgn = np.sign((mine["rotation"] * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(mine["d"] ** 2 / 2, meta["var.explained"], rtol=1e-9)
    np.testing.assert_allclose(mine["rotation"] * sgn[None, :], meta["rotation"], rtol=1e-5, atol=1e-7)
    for got, want in zip(mine["pcs"], ref):
        np.testing.assert_allclose(got * sgn[None, :], want, rtol=1e-5, atol=1e-7)


def test_small_or_low_rank_inputs_take_the_host_path(bx, pca):
    # what the 64-vector device block cannot take: 60 genes; d = 130; data of exact rank 10
    rng = np.random.default_rng(1200009)
    few = [rng.standard_normal((60, 300)), rng.standard_normal((60, 200)) + 0.5]
    ref, meta = pca.multi_batch_pca(few, d=5)
    mine = bx.multiBatchPCA(*few, d=5)
    assert mine["path"].startswith("host")
    sgn = np.sign((mine["rotation"] * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(mine["rotation"] * sgn[None, :], meta["rotation"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(mine["pcs"][1] * sgn[None, :], ref[1], rtol=1e-6, atol=1e-8)
    low = [rng.standard_normal((300, 10)) @ rng.standard_normal((10, n)) for n in (400, 500)]
    ref, meta = pca.multi_batch_pca(low, d=6)
    mine = bx.multiBatchPCA(*low, d=6)        # (host if the Cholesky QR notices the rank, device otherwise: both fine)
    sgn = np.sign((mine["rotation"] * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(mine["rotation"] * sgn[None, :], meta["rotation"], rtol=1e-6, atol=1e-8)
    wide = [rng.standard_normal((300, 400)), rng.standard_normal((300, 500))]
    assert bx.multiBatchPCA(*wide, d=130, return_pcs=False)["path"].startswith("host")
    # round 1's host signature still works
    l2 = [bx.cosineNorm(m, mode="l2norm") for m in few]
    old = bx.multiBatchPCA(*few, d=5, l2=l2, block=128)
    assert old["path"].startswith("host") and old["rotation"].shape == (60, 5)


def test_device_pca_without_a_spectral_gap_converges_by_itself(bx, pca):
    # pure noise: the d-th and the 65th eigenvalue are close.  Default arguments: the iteration runs until the Ritz
    # residual says it is done (round 2 stopped after a fixed 15 sweeps and was off by 0.1 here)
    rng = np.random.default_rng(1200002)
    t1, t2 = rng.standard_normal((200, 900)), rng.standard_normal((200, 700)) + 0.3
    ref, meta = pca.multi_batch_pca([t1, t2], d=8)
    mine = bx.multiBatchPCA(t1, t2, d=8)
    assert mine["path"] == "device" and mine["residual"] <= 1e-9
    rot = align_sign(mine["rotation"], meta["rotation"])
    np.testing.assert_allclose(rot, meta["rotation"], rtol=1e-5, atol=1e-7)
    # a budget that is too small is an error, not a silently wrong subspace
    with pytest.raises(RuntimeError, match="did not reach the tolerance"):
        bx.multiBatchPCA(t1, t2, d=8, max_iters=3)
    # the fixed-count form of round 2 is still there for whoever asks for it
    fixed = bx.multiBatchPCA(t1, t2, d=8, iters=5, return_pcs=False)
    assert fixed["iters_used"] == 5


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
    with pytest.raises(ValueError, match="'batch' must be specified"):   # one object, no batch= (R/checkInputs.R:128)
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
    out = bx.fastMNN(*B, d=8)
    ref, meta = pca.fast_mnn(*B, d=8)
    sgn = np.sign((out.rotation * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(out.rotation * sgn[None, :], meta["rotation"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(out.corrected * sgn[None, :], ref.corrected, rtol=1e-5, atol=1e-8)
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert np.array_equal(ol, rl) and np.array_equal(orr, rr)


def test_fast_mnn_single_object_with_batch_labels(bx, pca):
    # .fast_mnn_single (R/fastMNN.R:364-388): one genes x cells matrix + batch=, cells of the batches interleaved;
    # results and pairs in the caller's cell order (tests/testthat/test-fast-mnn.R:152-177)
    rng = np.random.default_rng(1200012)
    G, r = 300, 6
    load = rng.standard_normal((G, r)) * 2.0
    n = [900, 700, 800]
    mats = [np.abs(load @ rng.standard_normal((r, m)) + 0.4 * rng.standard_normal((G, m)) + 5.0 + 0.6 * i)
            for i, m in enumerate(n)]
    x = np.hstack(mats)
    labels = np.repeat(["b2", "b0", "b1"], n)
    shuffle = rng.permutation(x.shape[1])
    x, labels = x[:, shuffle], labels[shuffle]
    out = bx.fastMNN(x, batch=labels, d=6)
    ref, meta = pca.fast_mnn_single(x, labels, d=6)
    sgn = np.sign((out.rotation * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(out.corrected * sgn[None, :], ref.corrected, rtol=1e-5, atol=1e-8)
    assert list(out.batch) == list(labels)            # names(batches)[batch] (R/fastMNN.R:424)
    assert list(np.asarray(sorted(set(labels)))[ref.batch - 1]) == list(labels)
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert np.array_equal(ol, rl) and np.array_equal(orr, rr)
    with pytest.raises(ValueError, match="'batch' must be specified"):
        bx.fastMNN(x)


@pytest.mark.parametrize("where", ["device", "host"])
def test_config4_shape_20000_genes_50_pcs_4_batches(bx, pca, where):
    # BASELINE.json configs[3] at test scale: 20 000 genes -> cosineNorm -> multiBatchPCA(d = 50) -> reducedMNN over
    # 4 batches (3 000 cells each: 1.9 GB of input), PCA on the device and on the host (R/fastMNN.R:339-358,
    # R/multiBatchPCA.R:211-258), against the oracle's dense decomposition; MNN pairs bit-exact
    rng = np.random.default_rng(20250314 + 4000)
    G, d, nb, n = 20000, 50, 4, 3000
    load = np.abs(rng.standard_normal((G, d))) * (1.0 / np.sqrt(1.0 + np.arange(d) / 5.0))
    B = [load @ rng.standard_normal((d, n)) + 0.5 * (rng.random((G, n)) - 0.5) * 3.4641 + 4.0 + 0.3 * b for b in range(nb)]
    out = bx.fastMNN(*B, d=d, pca=where)
    ref, meta = pca.fast_mnn(*B, d=d, pca_method="gram")
    sgn = np.sign((out.rotation * meta["rotation"]).sum(axis=0))
    np.testing.assert_allclose(out.rotation * sgn[None, :], meta["rotation"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out.corrected * sgn[None, :], ref.corrected, rtol=1e-5, atol=1e-8)
    assert [p[0].size for p in out.merge_info.pairs] == [p[0].size for p in ref.merge_info.pairs]
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert np.array_equal(ol, rl) and np.array_equal(orr, rr)
    np.testing.assert_allclose(out.merge_info.lost_var, ref.merge_info.lost_var, rtol=1e-6, atol=1e-10)
