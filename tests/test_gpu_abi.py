"""GPU tests of the C ABI exactly as INTEGRATION.md's .Call shim drives it: the one-shot bmx_fast_mnn() with R's
memory layout (column-major doubles, int32, 1-based ids), then bmx_engine_pairs() on the returned handle; plus
edge shapes of the boundary: `restrict` as an arbitrary R subsetting vector, and the exact FP64 path with more
queries than a launch grid's y dimension holds."""
import ctypes

import numpy as np
import pytest

from tests.conftest import synth_batches

pytestmark = pytest.mark.gpu


def test_bmx_fast_mnn_one_shot_as_the_shim_calls_it(oracle):
    from batchelor_amd import _lib
    from batchelor_amd.merge_tree import encode_postorder, resolve_merge_order
    from batchelor_amd.reduced_mnn import BmxParams
    L = _lib.lib()
    B = synth_batches(8, [900, 700, 800], 25)
    nb, d = len(B), 25
    mats = [np.asfortranarray(b) for b in B]                       # REAL(x): column-major
    data = (ctypes.c_void_p * nb)(*[m.ctypes.data for m in mats])
    nrows = np.asarray([m.shape[0] for m in mats], dtype=np.int32)
    n_restrict = np.full(nb, -1, dtype=np.int32)                   # restrict = NULL
    tree = encode_postorder(resolve_merge_order(nb, [3, 1, 2]))    # merge.order = c(3, 1, 2)
    params = BmxParams(ctypes.sizeof(BmxParams), 20, float("nan"), 3.0, 0.0, 0)   # var_adj / sigma left zero-initialised
    N = int(nrows.sum())
    corrected = np.zeros((N, d), dtype=np.float64, order="F")
    batch = np.zeros(N, dtype=np.int32)
    ml = np.zeros((nb - 1, nb), dtype=np.int32)
    mr = np.zeros((nb - 1, nb), dtype=np.int32)
    bs = np.zeros(nb - 1)
    sk = np.zeros(nb - 1, dtype=np.int32)
    lv = np.zeros((nb - 1, nb), dtype=np.float64, order="F")
    h = ctypes.c_void_p()
    rc = L.bmx_fast_mnn(nb, d, data, _lib.i32p(nrows), None, _lib.i32p(n_restrict), ctypes.byref(params),
                        _lib.i32p(tree), int(tree.size), _lib.f64p(corrected), _lib.i32p(batch), _lib.i32p(ml),
                        _lib.i32p(mr), _lib.f64p(bs), _lib.i32p(sk), _lib.f64p(lv), ctypes.byref(h))
    _lib.check(rc)
    try:
        ref = oracle.reduced_mnn(*B, merge_order=[3, 1, 2])
        np.testing.assert_allclose(corrected, ref.corrected, rtol=1e-5, atol=1e-12)
        assert batch.tolist() == list(ref.batch)
        assert [[int(x) for x in row if x] for row in ml] == ref.merge_info.left
        assert [[int(x) for x in row if x] for row in mr] == ref.merge_info.right
        np.testing.assert_allclose(bs, ref.merge_info.batch_size, rtol=1e-9)
        assert sk.tolist() == [0, 0]
        np.testing.assert_allclose(lv, ref.merge_info.lost_var, rtol=1e-7, atol=1e-12)
        for m in range(nb - 1):
            pl, pr, n = _lib.c_i32p(), _lib.c_i32p(), ctypes.c_int64(0)
            _lib.check(L.bmx_engine_pairs(h, m, ctypes.byref(pl), ctypes.byref(pr), ctypes.byref(n)))
            gl, gr = _lib.take_i32(pl, n.value), _lib.take_i32(pr, n.value)
            assert np.array_equal(gl, ref.merge_info.pairs[m][0]) and np.array_equal(gr, ref.merge_info.pairs[m][1])
        # every merge's lists in one call into the caller's arrays (what the shim does for merge.info's `pairs`)
        nm = nb - 1
        cap = np.asarray([ref.merge_info.pairs[m][0].size for m in range(nm)], dtype=np.int64)
        mine = [(np.full(c + 3, -7, dtype=np.int32), np.full(c + 3, -7, dtype=np.int32)) for c in cap]
        lp = (ctypes.c_void_p * nm)(*[a.ctypes.data for a, _ in mine])
        rp = (ctypes.c_void_p * nm)(*[b.ctypes.data for _, b in mine])
        i64p = ctypes.POINTER(ctypes.c_int64)
        _lib.check(L.bmx_engine_pairs_all_into(h, nm, lp, rp, (cap + 3).ctypes.data_as(i64p)))
        for m in range(nm):
            for got, want in zip(mine[m], ref.merge_info.pairs[m]):
                assert np.array_equal(got[:cap[m]], want) and (got[cap[m]:] == -7).all()
        short = cap.copy()
        short[nm - 1] -= 1
        assert L.bmx_engine_pairs_all_into(h, nm, lp, rp, short.ctypes.data_as(i64p)) != 0
        assert "too short" in L.bmx_last_error().decode()
        assert L.bmx_engine_pairs_all_into(h, nm + 1, lp, rp, cap.ctypes.data_as(i64p)) != 0
        assert "number of merges" in L.bmx_last_error().decode()
        # error path: the reference's message, no engine handed back
        bad = np.asarray([1, 2, 2, 0, 0], dtype=np.int32)
        h2 = ctypes.c_void_p()
        rc = L.bmx_fast_mnn(nb, d, data, _lib.i32p(nrows), None, _lib.i32p(n_restrict), ctypes.byref(params),
                            _lib.i32p(bad), 5, None, None, None, None, None, None, None, ctypes.byref(h2))
        assert rc != 0 and not h2.value
        assert "invalid leaf nodes" in L.bmx_last_error().decode()
    finally:
        L.bmx_engine_destroy(h)


def test_restrict_is_an_r_subsetting_vector_in_any_order(oracle):
    # checkRestrictions() turns `restrict` into integer positions in the caller's order (R/checkInputs.R:96-120,
    # R/utils_subset.R): an unsorted vector is legal and changes the order of the pairs.
    import batchelor_amd as bx
    from tests.test_gpu_engine import assert_same_result
    rng = np.random.default_rng(77)
    B = synth_batches(9, [500, 400, 450], 15)
    keep = [rng.permutation(500)[:300] + 1, None, rng.permutation(450)[:200] + 1]
    out = bx.reducedMNN(*B, restrict=keep)
    ref = oracle.reduced_mnn(*B, restrict=keep)
    assert_same_result(out, ref)


@pytest.mark.parametrize("where", ["left", "right", "both"])
def test_restrict_may_name_a_cell_more_than_once(oracle, where):
    """Any R subsetting vector is a legal `restrict` (R/checkInputs.R:96-120), so a cell may be named twice: the searches and
    the centring mean then see it as two points (R/MNN_tree.R:113-127, R/fastMNN.R:633-637), its pairs come out once per
    point, and .average_correction's rowsum (R/fastMNN.R:571-579) still groups them by CELL.  Three batches, so that the
    merged node of the first merge carries the repeated positions into the second."""
    import batchelor_amd as bx
    from tests.test_gpu_engine import assert_same_result
    rng = np.random.default_rng(78)
    B = synth_batches(9, [500, 400, 450], 15)
    base = [rng.permutation(500)[:300] + 1, rng.permutation(400)[:250] + 1, rng.permutation(450)[:200] + 1]
    keep = [b.copy() for b in base]
    rep = lambda r, n: rng.permutation(np.concatenate([r, r[:n], r[:n // 3]]))  # some cells twice, some three times
    if where in ("left", "both"):
        keep[0] = rep(keep[0], 40)
    if where in ("right", "both"):
        keep[1] = rep(keep[1], 60)
        keep[2] = rep(keep[2], 30)
    out = bx.reducedMNN(*B, restrict=keep)
    ref = oracle.reduced_mnn(*B, restrict=keep)
    assert_same_result(out, ref)
    # the same through the tree form, where a node with repeats also turns up as the RIGHT child
    tree = [[3, 2], 1]
    out = bx.reducedMNN(*B, restrict=keep, merge_order=tree)
    ref = oracle.reduced_mnn(*B, restrict=keep, merge_order=tree)
    assert_same_result(out, ref)


def test_exact_path_with_more_queries_than_grid_y(oracle):
    # nr <= 48 references: below the candidate pass's minimum, every query takes the exact FP64 scan, whose launches
    # carry the queries in grid.y (at most 65535 per launch)
    from batchelor_amd import neighbors as nb
    rng = np.random.default_rng(3)
    X = rng.standard_normal((40, 6))
    Q = rng.standard_normal((70000, 6))
    idx, dist = nb.query_knn(X, Q, 7)
    rows = np.concatenate([np.arange(0, 70000, 997), [65534, 65535, 65536, 69999]])
    oi, od = oracle.query_knn(X, Q[rows], 7)
    assert np.array_equal(idx[rows], oi) and np.array_equal(dist[rows], od)
    assert idx.min() >= 1 and idx.max() <= 40 and np.all(np.diff(dist, axis=1) >= 0)


def test_params_struct_of_an_older_header_and_unset_size(oracle):
    # bmx_params_t.struct_size: a caller built against round 1's header (fields up to auto_merge) gets the defaults of
    # the fields added later (var_adj = 0, sigma = 0.1); a struct whose size was never set is refused, not misread
    from batchelor_amd import _lib
    from batchelor_amd.merge_tree import encode_postorder, resolve_merge_order
    from batchelor_amd.reduced_mnn import MnnEngine
    from tests.test_gpu_engine import assert_same_result

    class OldParams(ctypes.Structure):
        _fields_ = [("struct_size", ctypes.c_int32), ("k", ctypes.c_int32), ("prop_k", ctypes.c_double),
                    ("ndist", ctypes.c_double), ("min_batch_skip", ctypes.c_double), ("auto_merge", ctypes.c_int32)]

    B = synth_batches(12, [600, 500], 20)
    eng = MnnEngine()
    try:
        eng.upload(B)
        tree = encode_postorder(resolve_merge_order(2))
        old = OldParams(ctypes.sizeof(OldParams), 20, float("nan"), 3.0, 0.0, 0)
        _lib.check(_lib.lib().bmx_engine_run(eng._h, ctypes.byref(old), _lib.i32p(tree), int(tree.size)))
        assert_same_result(eng.download(), oracle.reduced_mnn(*B))
        unset = OldParams(0, 20, float("nan"), 3.0, 0.0, 0)
        rc = _lib.lib().bmx_engine_run(eng._h, ctypes.byref(unset), _lib.i32p(tree), int(tree.size))
        assert rc != 0 and "struct_size" in _lib.lib().bmx_last_error().decode()
    finally:
        eng.close()


def test_watchdog_turns_a_stuck_stream_into_an_error():
    # the candidate kernels poll LDS words without a bound (a bound ending in s_trap doubled their run time), so the host
    # side never waits without a deadline.  Fault injection: a kernel that keeps the stream busy for 1.5 s (and then ends
    # by itself) in front of a run whose waits are given 0.2 s.
    import time
    import batchelor_amd as bx
    B = synth_batches(13, [500, 400], 10)
    eng = bx.MnnEngine()
    eng.upload(B)
    eng.run()                      # healthy first
    eng.set_watchdog(200)
    eng._debug_stall(1500)
    t0 = time.perf_counter()
    with pytest.raises(bx.BatchelorMI355XError, match="watchdog"):
        eng.run()
    assert time.perf_counter() - t0 < 1.2      # gave up at the deadline, did not sit the stall out
    with pytest.raises(bx.BatchelorMI355XError, match="dead after a watchdog timeout"):
        eng.run()                  # the engine stays dead: refused at once, nothing is queued or waited for ...
    with pytest.raises(bx.BatchelorMI355XError, match="dead after a watchdog timeout"):
        eng.download()
    eng.close()                    # ... and closing it does not wait for the stream either
    time.sleep(1.6)                # the stall kernel ends by itself: the GPU is fine, a new engine works
    fresh = bx.MnnEngine()
    fresh.upload(B)
    fresh.run()
    assert np.all(np.isfinite(fresh.download().corrected))
    fresh.close()


def test_testing_hooks_device_choice_and_cache_trim():
    """bmx_dev_set refuses knobs it does not know (a typo must not silently run the default), bmx_set_device reports the
    runtime's error for a device that is not there, bmx_trim_caches hands the parked device blocks back and the next call
    simply allocates again."""
    import batchelor_amd as bx
    from batchelor_amd import _lib
    L = _lib.lib()
    assert L.bmx_dev_set(b"no_such_knob", 1) != 0 and "unknown knob" in L.bmx_last_error().decode()
    assert L.bmx_dev_set(b"force_c", 2) == 0 and L.bmx_dev_set(b"reset", 0) == 0
    assert L.bmx_set_device(0) == 0
    assert L.bmx_set_device(4096) != 0
    assert L.bmx_set_device(0) == 0
    B = synth_batches(21, [700, 600], 10)
    a = bx.reducedMNN(*B)
    L.bmx_trim_caches.restype = None
    L.bmx_trim_caches()
    b = bx.reducedMNN(*B)
    assert np.array_equal(a.corrected, b.corrected)


def test_restrict_with_repeats_under_variance_adjustment(oracle):
    # repeated cells are repeated points to adjust_shift_variance's sums as well (the reference loops over the restrict
    # vectors as they are, src/adjust_shift_variance.cpp:74,118)
    import batchelor_amd as bx
    rng = np.random.default_rng(79)
    B = synth_batches(9, [500, 400], 15)
    keep = [rng.choice(500, 300, replace=True) + 1, rng.choice(400, 250, replace=True) + 1]
    out = bx.reducedMNN(*B, restrict=keep, var_adj=True, sigma=1.0)
    ref = oracle.reduced_mnn(*B, restrict=keep, var_adj=True, sigma=1.0)
    assert np.array_equal(out.merge_info.pairs[0][0], ref.merge_info.pairs[0][0])
    assert np.array_equal(out.merge_info.pairs[0][1], ref.merge_info.pairs[0][1])
    close = np.isclose(out.corrected, ref.corrected, rtol=1e-5, atol=1e-9).all(axis=1)
    assert close.mean() > 0.99, close.mean()
