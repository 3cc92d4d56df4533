"""GPU parity of the MI355X merge engine against the CPU oracle, through the C ABI.

Bar (BASELINE.json north_star): MNN pair indices bit-exact, corrected coordinates within 1e-5 relative
(they come out ~1e-13 here; the tolerance below is the stated one)."""
import numpy as np
import pytest

from tests.conftest import synth_batches

pytestmark = pytest.mark.gpu
RTOL = 1e-5   # north_star tolerance for corrected coordinates


@pytest.fixture(scope="module")
def bx():
    import batchelor_amd
    return batchelor_amd


def assert_same_result(out, ref, rtol=RTOL):
    assert out.corrected.shape == ref.corrected.shape
    scale = np.abs(ref.corrected).max(axis=0)
    err = np.abs(out.corrected - ref.corrected).max(axis=0) / scale
    assert err.max() < rtol, err.max()
    np.testing.assert_allclose(out.corrected, ref.corrected, rtol=rtol, atol=1e-12)
    assert list(out.batch) == list(ref.batch)
    assert out.merge_info.left == ref.merge_info.left and out.merge_info.right == ref.merge_info.right
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert np.array_equal(ol, rl) and np.array_equal(orr, rr)      # bit-exact, order included
    np.testing.assert_allclose(out.merge_info.batch_size, ref.merge_info.batch_size, rtol=1e-9, equal_nan=True)
    assert np.array_equal(out.merge_info.skipped, ref.merge_info.skipped)
    np.testing.assert_allclose(out.merge_info.lost_var, ref.merge_info.lost_var, rtol=1e-7, atol=1e-12)
    return err.max()


def test_config1_two_batches(oracle, bx):
    # BASELINE.json configs[0]: 2 x 2000 x 50
    B = synth_batches(1, [2000, 2000], 50)
    out = bx.reducedMNN(*B)
    ref = oracle.reduced_mnn(*B)
    err = assert_same_result(out, ref)
    assert err < 1e-10
    assert 0.0 < out.merge_info.batch_size[0] < 1.0 and not out.merge_info.skipped[0]


def test_golden_fixture_config1(bx):
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "config1_reduced_mnn.npz")
    g = np.load(path)
    B = synth_batches(1, [2000, 2000], 50)
    out = bx.reducedMNN(*B)
    np.testing.assert_allclose(out.corrected[g["rows"]], g["corrected_rows"], rtol=RTOL, atol=1e-12)
    assert np.array_equal(out.merge_info.pairs[0][0], g["pairs_left"])
    assert np.array_equal(out.merge_info.pairs[0][1], g["pairs_right"])
    np.testing.assert_allclose(out.merge_info.lost_var, g["lost_var"], rtol=1e-7)
    np.testing.assert_allclose(out.merge_info.batch_size, g["batch_size"], rtol=1e-9)


@pytest.mark.parametrize("sizes,d,kw", [
    ([1500, 2500, 1000], 50, {}),
    ([700, 600, 900, 500], 20, {"merge_order": [[1, 4], [3, 2]]}),
    ([700, 600, 900, 500], 20, {"merge_order": [4, 2, 1, 3]}),
    ([900, 1100], 100, {"k": 10}),
    ([400, 300], 10, {"k": 30}),
    ([400, 300], 10, {"k": 50}),           # beyond the MFMA path's k: exact scan
    ([800, 1000], 30, {"prop_k": 0.05}),
    ([300, 500], 10, {"min_batch_skip": None}),
])
def test_engine_matches_oracle(oracle, bx, sizes, d, kw):
    B = synth_batches(7, sizes, d)
    out = bx.reducedMNN(*B, **kw)
    ref = oracle.reduced_mnn(*B, **kw)
    assert_same_result(out, ref)


def test_grid_kats_on_gpu(bx):
    # tests/testthat/test-reduced-mnn.R:80-105 -- exact toy answers, massive ties (exact-path territory)
    core = np.column_stack([np.repeat(np.arange(1, 11), 10), np.tile(np.arange(1, 11), 10)]).astype(np.float64)
    b1, b2 = core.copy(), core.copy()
    b1[:, 0] += 20
    b2[:, 1] += 20
    out1 = bx.reducedMNN(core, b1, k=1)
    np.testing.assert_allclose(out1.corrected[:, 0], 5.5, atol=1e-12)
    np.testing.assert_allclose(out1.corrected[:, 1], np.concatenate([core[:, 1], b1[:, 1]]), atol=1e-12)
    out2 = bx.reducedMNN(core, b1, b2, k=1)
    np.testing.assert_allclose(out2.corrected, 5.5, atol=1e-12)
    outY = bx.reducedMNN(core + 10, b2 + 10, k=1)
    np.testing.assert_allclose(outY.corrected[:, 0], np.concatenate([core[:, 0], b2[:, 0]]) + 10, atol=1e-12)
    np.testing.assert_allclose(outY.corrected[:, 1], 15.5, atol=1e-12)
    outZ = bx.reducedMNN(core, b1, core + 10, b2 + 10, merge_order=[[1, 2], [3, 4]], k=1)
    np.testing.assert_allclose(outZ.corrected, 5.5, atol=1e-12)


def test_skip_and_no_offset(oracle, bx):
    rng = np.random.default_rng(3)
    A1, A2 = rng.standard_normal((300, 20)), rng.standard_normal((400, 20))
    out = bx.reducedMNN(A1, A2, min_batch_skip=0.1)
    assert out.merge_info.skipped[0] and np.all(out.merge_info.lost_var == 0)
    assert np.array_equal(out.corrected, np.vstack([A1, A2]))   # untouched, bit for bit
    ref = oracle.reduced_mnn(A1, A2, min_batch_skip=0.1)
    assert_same_result(out, ref)


def test_restriction(oracle, bx):
    # tests/testthat/test-reduced-mnn.R:107-133
    rng = np.random.default_rng(12000053)
    B1, B2, B3 = (rng.standard_normal((n, 10)) + i for i, n in enumerate([100, 200, 50]))
    i1, i2, i3 = np.arange(99, 48, -1), np.arange(0, 20), np.arange(20, 45)
    C1, C2, C3 = np.vstack([B1, B1[i1]]), np.vstack([B2, B2[i2]]), np.vstack([B3, B3[i3]])
    keep = [np.arange(1, 101), np.arange(1, 201), np.arange(1, 51)]
    out = bx.reducedMNN(C1, C2, C3, restrict=keep)
    ref = oracle.reduced_mnn(C1, C2, C3, restrict=keep)
    assert_same_result(out, ref)
    plain = bx.reducedMNN(B1, B2, B3)
    for b, (n, ii) in enumerate(((100, i1), (200, i2), (50, i3)), start=1):
        r = plain.corrected[plain.batch == b]
        o = out.corrected[out.batch == b]
        np.testing.assert_allclose(o[:n], r, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(o[n:], r[ii], rtol=1e-12, atol=1e-13)


def test_single_object_and_names(oracle, bx):
    B = synth_batches(9, [300, 500, 400], 20)
    com = np.vstack(B)
    labels = np.repeat(["c", "a", "b"], [300, 500, 400])
    shuffle = np.random.default_rng(0).permutation(1200)
    out = bx.reducedMNN(com[shuffle], batch=labels[shuffle])
    ref = oracle.reduced_mnn(com[shuffle], batch=labels[shuffle])
    np.testing.assert_allclose(out.corrected, ref.corrected, rtol=RTOL, atol=1e-12)
    for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs):
        assert np.array_equal(ol, rl) and np.array_equal(orr, rr)
    named = bx.reducedMNN(*B, names=["X", "Y", "Z"], merge_order=["Z", "X", "Y"])
    assert list(named.batch[:3]) == ["X"] * 3 and named.merge_info.left[0] == ["Z"]
    with pytest.raises(ValueError, match="names of batches should be unique"):
        bx.reducedMNN(*B, names=["X", "X", "Z"])
    with pytest.raises(ValueError, match="invalid leaf nodes"):
        bx.reducedMNN(*B, merge_order=[1, 2, 2])


def test_auto_merge(oracle, bx):
    B = synth_batches(11, [600, 900, 500, 400], 20)
    out = bx.reducedMNN(*B, auto_merge=True)
    ref = oracle.reduced_mnn(*B, auto_merge=True)
    assert_same_result(out, ref)


def test_engine_reuse_and_profile(bx):
    B = synth_batches(2, [3000, 3000], 50)
    eng = bx.MnnEngine()
    eng.upload(B)
    eng.set_profiling(True)
    eng.run()
    a = eng.download()
    eng.run()
    b = eng.download()
    assert np.array_equal(a.corrected, b.corrected)          # deterministic, bit for bit
    assert np.array_equal(a.merge_info.pairs[0][0], b.merge_info.pairs[0][0])
    prof = eng.profile()
    assert prof["topk_launches"] == 3 and prof["topk_ms"] > 0
    eng.close()


def test_engine_random_configurations(oracle, bx):
    # a seeded draw over batch counts, sizes, dimensions, k / prop.k, merge orders, auto-merge and restrictions
    # (scripts/engine_stress.py runs the same generator for as long as one likes)
    rng = np.random.default_rng(20250315)
    for case in range(8):
        nb = int(rng.integers(2, 6))
        sizes = [int(rng.choice([60, 150, 400, 900, 2000])) for _ in range(nb)]
        d = int(rng.choice([2, 5, 20, 50, 64, 100]))
        mode = case % 5
        kw = {}
        if mode == 0:
            kw["k"] = int(rng.choice([2, 5, 10, 25, 30]))
        elif mode == 1:
            kw["prop_k"] = float(rng.choice([0.01, 0.05, 0.1]))
        elif mode == 2:
            kw["merge_order"] = [int(x) for x in rng.permutation(nb) + 1]
        elif mode == 3:
            kw["auto_merge"] = True
        else:
            kw["restrict"] = [np.sort(rng.choice(n, size=max(30, n // 2), replace=False)) + 1 for n in sizes]
        B = synth_batches(400 + case, sizes, d)
        assert_same_result(bx.reducedMNN(*B, **kw), oracle.reduced_mnn(*B, **kw))


def test_optimistic_run_starts_over_when_a_search_cannot_be_completed_on_the_device(oracle, bx):
    """A run first leaves the count of a search's uncertified queries on the device and sweeps them there (no host round
    trip); more than 256 of them, or a list that overflows with near-ties, raises a device flag and the run is repeated with
    host-checked searches (Engine::run).  Clusters of 150 near-duplicates put thousands of references inside the fp16 pass's
    error margin of every query's k-th neighbour: the first tier certifies next to nothing -- the retry must happen and the
    result must be the oracle's."""
    rng = np.random.default_rng(4242)
    centres = rng.standard_normal((30, 50)) * 2.0
    B = [np.repeat(centres, 150, axis=0) + 1e-4 * rng.standard_normal((4500, 50)) + 0.3 * b for b in range(2)]
    eng = bx.MnnEngine(0)
    try:
        eng.upload(B)
        eng.set_profiling(True)
        eng.run()
        out = eng.download()
        assert eng.profile_detail()["optimistic_retries"] >= 1
    finally:
        eng.close()
    ref = oracle.reduced_mnn(*B)
    assert_same_result(out, ref)


def test_large_k_with_restriction_and_three_batches(oracle, bx):
    # k = 50 (beyond the candidate tiers' lists: the partitioned search of knn.hip) through the whole engine, with a restrict
    # vector on two of three batches (the searches then run over row LISTS) and prop.k on top
    B = synth_batches(41, [5000, 4200, 3600], 50)
    keep = [np.arange(1, 4001), None, np.arange(300, 3500)]
    assert_same_result(bx.reducedMNN(*B, k=50, restrict=keep), oracle.reduced_mnn(*B, k=50, restrict=keep))
    assert_same_result(bx.reducedMNN(*B, k=20, prop_k=0.02, restrict=keep), oracle.reduced_mnn(*B, k=20, prop_k=0.02, restrict=keep))


@pytest.mark.parametrize("outliers,retries", [(True, 0), (False, 1)])
def test_large_k_queries_the_merge_cannot_certify(oracle, bx, outliers, retries):
    """k = 100: the searches are partitioned (knn.hip, large_k_search: reference rows g with the same g % 7).  Six cells of the
    first batch each have 40 exact copies in the second, all in its first partition: the merge of the partitions' lists cannot
    certify a query that has them among its 100 nearest.  With the six far from everything else only they themselves do: in the
    engine's optimistic run (no host read-back inside a search) those queries take the FP64 scan on the device -- one such query
    used to send the whole run back to its start.  With the six in the crowd hundreds of queries do, more than the device-side
    path takes: the run starts over with host-checked searches -- and what is queued behind the search that gave up must
    survive its spoilt lists (every index in range).  Either way the result is the oracle's, pairs bit for bit."""
    B1, B2 = synth_batches(31, [3000, 30000], 20)
    if outliers:
        B1[:6] = 12.0 * np.eye(20)[:6]
    for j in range(6):
        B2[(np.arange(40) + 40 * j) * 7] = B1[j]
    eng = bx.MnnEngine()
    try:
        eng.upload([B1, B2])
        eng.run(k=100)
        pd = eng.profile_detail()
        assert pd["optimistic_retries"] == retries
        assert pd["exact_fallbacks"] >= 6
    finally:
        eng.close()
    assert_same_result(bx.reducedMNN(B1, B2, k=100), oracle.reduced_mnn(B1, B2, k=100))


@pytest.mark.parametrize("d,kw", [(140, {}), (131, {"k": 70}), (200, {"prop_k": 0.1})])
def test_rows_beyond_125_columns(oracle, bx, d, kw):
    """No candidate tier takes rows of more than 125 columns: every search of the run is the FP64 scan (knn.hip: knn_exact_dist in
    LDS tiles, knn_exact_select in two sweeps / by bisection) -- same result as the oracle's, pairs bit for bit."""
    B = synth_batches(41, [1500, 2000, 900], d)
    assert_same_result(bx.reducedMNN(*B, **kw), oracle.reduced_mnn(*B, **kw))
