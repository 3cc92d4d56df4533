"""GPU parity on the shape of BASELINE.json configs[4], scaled down: 16 unequal batches (log-uniform sizes), 100 PCs,
a balanced merge.order tree over the size-sorted batches -- sibling subtrees, nodes that carry several batch vectors
on BOTH sides of a merge, d = 100 (seven MFMA k-steps).  Whole result against the CPU oracle, pairs bit-exact."""
import numpy as np
import pytest

from tests.conftest import synth_batches
from tests.test_gpu_engine import assert_same_result

pytestmark = pytest.mark.gpu


def balanced_tree(ids):
    if len(ids) == 1:
        return ids[0]
    h = len(ids) // 2
    return [balanced_tree(ids[:h]), balanced_tree(ids[h:])]


def config5_scaled():
    rng = np.random.Generator(np.random.PCG64(20250314 + 5000))
    sizes = [int(x) for x in np.exp(rng.uniform(np.log(300), np.log(6000), 16))]
    order = [int(i) + 1 for i in np.argsort(sizes)[::-1]]
    return sizes, balanced_tree(order)


def test_config5_scaled_sixteen_batch_tree_vs_oracle(oracle):
    import batchelor_amd as bx
    sizes, tree = config5_scaled()
    B = synth_batches(5, sizes, 100)
    out = bx.reducedMNN(*B, merge_order=tree)
    ref = oracle.reduced_mnn(*B, merge_order=tree)
    assert_same_result(out, ref)
    assert len(out.merge_info.pairs) == 15
    # the last merge joins two 8-batch subtrees
    assert len(out.merge_info.left[-1]) == 8 and len(out.merge_info.right[-1]) == 8


def test_k30_d100_candidate_pass_vs_oracle(oracle):
    # k in (20, 36] with d > 84: the shape the split-bf16 kernel does not cover (see knn.hip: tier selection)
    from batchelor_amd import neighbors as nb
    X, Q = synth_batches(6, [5000, 1500], 100)
    idx, dist = nb.query_knn(X, Q, 30)
    oi, od = oracle.query_knn(X, Q, 30)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    B = synth_batches(6, [1500, 1200], 100)
    import batchelor_amd as bx
    assert_same_result(bx.reducedMNN(*B, k=30), oracle.reduced_mnn(*B, k=30))


def _check_var_adj_snapshot(oracle, snap, sigma, cells=None, bitwise=True):
    """adjust_shift_variance of one merge of an engine run against the oracle ON THE SAME INPUTS (the engine's own centred
    nodes and correction vectors, bmx_engine_snapshot_var_adj): the share of cells equal to 1e-8."""
    got = snap["scaling"] if cells is None else snap["scaling"][cells]
    ref = oracle.adjust_shift_variance(snap["left"].T, snap["right"].T, snap["correction"], sigma, snap["restrict1"],
                                       snap["restrict2"], cells=cells)
    if bitwise:
        assert np.array_equal(got, ref, equal_nan=True), int((got != ref).sum())
    return float(np.isclose(got, ref, rtol=1e-8, atol=1e-12, equal_nan=True).mean())


@pytest.mark.parametrize("d", [12, 100])
@pytest.mark.parametrize("sigma", [1.0, 0.3, 0.1])
def test_variance_adjusted_merge_vs_oracle(oracle, dev, sigma, d):
    """configs[4] runs "with adjust_shift_variance on": every right cell's correction vector is stretched by
    pmax(adjust_shift_variance(left, right, correction, sigma), 1) as mnnCorrect(var.adj=TRUE) does
    (R/mnnCorrect.R:331-342,462-481).  Two claims, kept apart:
    (1) GIVEN its inputs the step is the reference's, bit for bit: the engine's own inputs of the call (snapshot) through the
        oracle give the engine's scalings exactly -- with the exact form and with the tiled one (testing hook "asv_fast") --
        and the engine's corrected right cells are right + pmax(scaling, 1) * correction to rounding.
    (2) End to end against the oracle's own run the merge agrees on every cell at sigma = 1.  At small bandwidths the scaling is
        a quantile decided by the LAST BIT of sequential log-sums (src/adjust_shift_variance.cpp:137-157), and the two runs'
        correction vectors differ in their 16th digit (tricube averages summed in different orders, as R's would from both):
        a few per cent of the cells then land on another quantile IN ANY TWO implementations of the preceding steps -- the
        oracle fed its own correction vectors moved by one ulp moves as many (asserted below)."""
    import batchelor_amd as bx
    B = synth_batches(5, [1200, 900], d)
    keep = [None, np.arange(1, 801)]
    ref = oracle.reduced_mnn(*B, var_adj=True, sigma=sigma, restrict=keep)
    shares = {}
    for fast in (0, 1):
        dev("asv_fast", fast)
        eng = bx.MnnEngine()
        eng.upload(B, restrict=keep)
        eng.set_snapshot(0)
        eng.run(var_adj=True, sigma=sigma)
        out, snap = eng.download(), eng.snapshot_var_adj()
        eng.close()
        assert np.array_equal(out.merge_info.pairs[0][0], ref.merge_info.pairs[0][0])
        assert np.array_equal(out.merge_info.pairs[0][1], ref.merge_info.pairs[0][1])
        # (1) the step given its inputs
        assert np.array_equal(snap["restrict2"], np.arange(800)) and snap["restrict1"].size == 1200
        # (exact form: every cell bit for bit; tiled form: the cells that take the histogram way carry their projection from
        # the matrix cores' sums, an ulp from the sequential inner product)
        assert _check_var_adj_snapshot(oracle, snap, sigma, bitwise=fast == 0) == 1.0
        with np.errstate(invalid="ignore"):
            stretch = np.where(snap["scaling"] < 1, 1.0, snap["scaling"])
        np.testing.assert_allclose(out.corrected[1200:], snap["right"] + stretch[:, None] * snap["correction"], rtol=1e-12,
                                   atol=1e-13)
        # (2) end to end
        np.testing.assert_allclose(out.corrected[:1200], ref.corrected[:1200], rtol=1e-5, atol=1e-12)
        close = np.isclose(out.corrected[1200:], ref.corrected[1200:], rtol=1e-5, atol=1e-9).all(axis=1)
        shares[fast] = float(close.mean())
        np.testing.assert_allclose(out.merge_info.lost_var, ref.merge_info.lost_var, rtol=1e-7, atol=1e-12)
    assert shares[0] == shares[1]          # both forms give the same cells
    # the reference's own sensitivity: its correction vectors moved by one ulp
    base = oracle.adjust_shift_variance(snap["left"].T, snap["right"].T, snap["correction"], sigma, snap["restrict1"],
                                        snap["restrict2"])
    moved = oracle.adjust_shift_variance(snap["left"].T, snap["right"].T, np.nextafter(snap["correction"], np.inf), sigma,
                                         snap["restrict1"], snap["restrict2"])
    stable = float(np.isclose(base, moved, rtol=1e-8, atol=1e-12).mean())
    print(f"d={d} sigma={sigma}: end-to-end share of right cells equal {shares[0]:.4f}; the oracle against itself with "
          f"correction vectors one ulp up: {stable:.4f}")
    if sigma >= 1.0:
        assert shares[0] >= 0.999, shares
    else:
        assert shares[0] >= 0.95 and shares[0] >= stable - 0.03, (shares, stable)
    plain = bx.reducedMNN(*B, restrict=keep)
    assert np.array_equal(plain.corrected[:1200], out.corrected[:1200])          # the reference side is untouched by it
    assert not np.array_equal(out.corrected[1200:], plain.corrected[1200:])      # the switch does something
    # a later merge runs on top of the adjusted cells without trouble
    three = bx.reducedMNN(*synth_batches(5, [1200, 900, 700], d), var_adj=True, sigma=sigma)
    assert np.all(np.isfinite(three.corrected)) and len(three.merge_info.pairs) == 2


def _scaled_tree():
    rng = np.random.Generator(np.random.PCG64(20250314 + 5001))
    sizes = [int(x) for x in np.exp(rng.uniform(np.log(150), np.log(2500), 16))]
    return sizes, balanced_tree([int(i) + 1 for i in np.argsort(sizes)[::-1]])


@pytest.mark.parametrize("fast", [0, 1])
def test_config5_scaled_tree_with_variance_adjustment(oracle, dev, fast):
    # BASELINE.json configs[4] as named -- the 16-batch tree "with adjust_shift_variance on" -- at test scale, sigma = 1 (the
    # bench's setting), exact form and tiled form (testing hook): the whole tree against the oracle's own run -- same merges,
    # every pair, every cell
    import batchelor_amd as bx
    dev("asv_fast", fast)
    sizes, tree = _scaled_tree()
    B = synth_batches(5, sizes, 100)
    out = bx.reducedMNN(*B, merge_order=tree, var_adj=True, sigma=1.0)
    again = bx.reducedMNN(*B, merge_order=tree, var_adj=True, sigma=1.0)
    assert np.array_equal(out.corrected, again.corrected)
    ref = oracle.reduced_mnn(*B, merge_order=tree, var_adj=True, sigma=1.0)
    assert out.merge_info.left == ref.merge_info.left and out.merge_info.right == ref.merge_info.right
    assert np.all(np.isfinite(out.corrected)) and len(out.merge_info.pairs) == 15
    close = np.isclose(out.corrected, ref.corrected, rtol=1e-5, atol=1e-9).all(axis=1)
    assert close.mean() >= 0.999, close.mean()
    same = [np.array_equal(ol, rl) and np.array_equal(orr, rr)
            for (ol, orr), (rl, rr) in zip(out.merge_info.pairs, ref.merge_info.pairs)]
    assert sum(same) >= 14, same
    plain = bx.reducedMNN(*B, merge_order=tree)
    assert not np.array_equal(plain.corrected, out.corrected)          # the switch does something


@pytest.mark.parametrize("sigma", [0.3, 0.1])
def test_config5_scaled_tree_every_merge_adjustment_matches_oracle_on_its_inputs(oracle, dev, sigma):
    """The same tree at mnnCorrect's default bandwidth (0.1) and at 0.3: a cell that lands on another quantile because its
    correction vector differs in the 16th digit (see test_variance_adjusted_merge_vs_oracle) changes every later merge, so
    whole-tree equality with the oracle's own run is not a property any two implementations have there.  What IS checked, at
    every one of the 15 merges, tiled form: the merge's adjust_shift_variance on the engine's own inputs against the oracle's
    on the same inputs, every cell to 1e-8 (the re-run cells are bit-equal, the histogram-way cells carry the matrix cores'
    projection).  Every chain of these merges keeps fewer addends than the re-run holds (131 072; at sigma 0.3 the root's
    chains keep thousands of the 7 128 reference cells: sorted in global memory), so EVERY cell must be equal.  (A call whose
    ill-conditioned cells keep more -- sigma 0.3 at BASELINE config 5's full size: a tenth of 2.5 million reference cells per
    chain -- sends those cells the histogram way; tests/test_gpu_primitives.py pins that path, DESIGN.md quantifies it.)"""
    import batchelor_amd as bx
    from batchelor_amd import _lib
    from batchelor_amd.merge_tree import resolve_merge_order
    dev("asv_fast", 1)
    sizes, tree = _scaled_tree()
    B = synth_batches(5, sizes, 100)
    eng = bx.MnnEngine()
    eng.upload(B)
    code = resolve_merge_order(len(sizes), tree)
    first, shares = None, []
    for m in range(15):
        eng.set_snapshot(m)
        _lib.dev_get("asv_tally_reset")
        eng.run(merge_tree=code, var_adj=True, sigma=sigma)
        back = _lib.dev_get("asv_fallback_cells")
        res = eng.download()
        if first is None:
            first = res
        assert np.array_equal(res.corrected, first.corrected)
        snap = eng.snapshot_var_adj()
        share = _check_var_adj_snapshot(oracle, snap, sigma, bitwise=False)
        shares.append((snap["left"].shape[0], snap["right"].shape[0], round(share, 4)))
        assert share == 1.0 and back == 0, (m, shares[-1], back)
    eng.close()
    print(f"sigma {sigma}: (left cells, right cells, share equal to the oracle on the same inputs) per merge: {shares}")
    assert np.all(np.isfinite(first.corrected))


@pytest.fixture(scope="module")
def full5():
    """BASELINE.json configs[4] at full size (the bench.py workload): 16 batches of 8 894 .. 281 334 cells, 100 PCs,
    balanced tree; two runs, each keeping the two matrices one merge searched."""
    import batchelor_amd as bx
    from bench import WORKLOADS
    cfg, sizes, d, k, tree = WORKLOADS["config5"]
    B = synth_batches(cfg, sizes, d)
    eng = bx.MnnEngine()
    eng.upload(B)
    from batchelor_amd.merge_tree import resolve_merge_order
    code = resolve_merge_order(len(sizes), tree)
    runs = {}
    for m in (9, 14):  # a merge in the middle of the tree, and the root: two eight-batch subtrees
        eng.set_snapshot(m)
        eng.run(k=k, merge_tree=code)
        runs[m] = (eng.download(), eng.snapshot())
    eng.close()
    return sizes, d, k, runs


def test_full_size_config5_deterministic(full5):
    sizes, d, k, runs = full5
    a, b = runs[9][0], runs[14][0]
    assert np.array_equal(a.corrected, b.corrected) and np.all(np.isfinite(a.corrected))
    for (l0, r0), (l1, r1) in zip(a.merge_info.pairs, b.merge_info.pairs):
        assert np.array_equal(l0, l1) and np.array_equal(r0, r1)
    assert len(a.merge_info.pairs) == 15 and not a.merge_info.skipped.any()


@pytest.mark.parametrize("m", [9, 14])
def test_full_size_config5_pairs_of_sampled_cells_match_oracle(oracle, full5, m):
    sizes, d, k, runs = full5
    res, (left, right) = runs[m]
    lset, rset = res.merge_info.left[m], res.merge_info.right[m]
    assert left.shape == (sum(sizes[b - 1] for b in lset), d) and right.shape == (sum(sizes[b - 1] for b in rset), d)
    start = np.concatenate([[0], np.cumsum(sizes)])

    def global_ids(bset):  # 1-based ids, in the input's cell order, of a node's rows (its batches in merge order)
        return np.concatenate([np.arange(start[b - 1], start[b]) + 1 for b in bset])
    gl, gr = global_ids(lset), global_ids(rset)
    rng = np.random.default_rng(500 + m)
    rows = np.sort(rng.choice(right.shape[0], 160, replace=False))     # sampled right cells (node rows)
    nn_l, _ = oracle.query_knn(left, right[rows], k)                   # their k nearest left rows (1-based)
    cand = np.unique(nn_l)
    nn_r, _ = oracle.query_knn(right, left[cand - 1], k)
    back = {int(c): set(row.tolist()) for c, row in zip(cand, nn_r)}
    expect = {(int(gl[l - 1]), int(gr[r])) for r, row in zip(rows, nn_l) for l in row.tolist() if int(r) + 1 in back[int(l)]}
    pl, pr = res.merge_info.pairs[m]
    keep = np.isin(pr, gr[rows])
    got = set(zip(pl[keep].tolist(), pr[keep].tolist()))
    assert got == expect and len(expect) > 50


@pytest.mark.parametrize("sigma", [1.0, 0.1])
def test_full_size_config5_with_variance_adjustment(oracle, full5, sigma):
    """BASELINE.json configs[4] as named, on one GPU: the full 16-batch tree WITH adjust_shift_variance on (6e11 cell pairs
    per run; the tiled FP64-MFMA form of legacy.hip), at the bench's bandwidth and at mnnCorrect's default.  The oracle cannot
    run the tree at this size, but it can run CELLS of it: the root merge's adjust_shift_variance call -- the largest of the
    run, two eight-batch subtrees -- is snapshotted and 192 sampled right cells go through the oracle on the same inputs
    (src/adjust_shift_variance.cpp:51: the loop treats every cell on its own).  Also: the run finishes, every coordinate is
    finite, the merge sets are the tree's, and the eight leaf-leaf merges -- upstream of any adjusted cell -- pair exactly as
    without the switch."""
    import batchelor_amd as bx
    from batchelor_amd import _lib
    from bench import WORKLOADS
    from batchelor_amd.merge_tree import resolve_merge_order
    sizes, d, k, runs = full5
    plain = runs[14][0]
    cfg, _, _, _, tree = WORKLOADS["config5"]
    B = synth_batches(cfg, sizes, d)
    eng = bx.MnnEngine()
    eng.upload(B)
    eng.set_snapshot(14)
    _lib.dev_get("asv_tally_reset")
    _lib.dev_set("asv_modes", 400000)       # (testing hook: which way each cell of the LAST call -- the root merge's -- went)
    try:
        eng.run(k=k, merge_tree=resolve_merge_order(len(sizes), tree), var_adj=True, sigma=sigma)
        out = eng.download()
        tally = [_lib.dev_get(n) for n in ("asv_literal_cells", "asv_fallback_cells", "asv_tiled_cells")]
        snap = eng.snapshot_var_adj()
        modes = _lib.dev_get_bytes("asv_modes", snap["right"].shape[0])
    finally:
        _lib.dev_set("asv_modes", 0)
        eng.close()
    assert np.all(np.isfinite(out.corrected)) and len(out.merge_info.pairs) == 15
    assert out.merge_info.left == plain.merge_info.left and out.merge_info.right == plain.merge_info.right
    leaf_merges = [m for m in range(15) if len(plain.merge_info.left[m]) == 1 and len(plain.merge_info.right[m]) == 1]
    assert len(leaf_merges) == 8
    for m in leaf_merges:
        assert np.array_equal(out.merge_info.pairs[m][0], plain.merge_info.pairs[m][0])
        assert np.array_equal(out.merge_info.pairs[m][1], plain.merge_info.pairs[m][1])
    assert not np.array_equal(out.corrected, plain.corrected)
    assert tally[2] > 0 and (sigma >= 1.0 or tally[0] > 0)     # (every call of the tree takes the tiled form at this size)
    assert set(np.unique(modes).tolist()) <= {0, 1, 2}
    cells = np.sort(np.random.default_rng(514).choice(snap["right"].shape[0], 192, replace=False)).astype(np.int32)
    got = snap["scaling"][cells]
    ref = oracle.adjust_shift_variance(snap["left"].T, snap["right"].T, snap["correction"], sigma, snap["restrict1"],
                                       snap["restrict2"], cells=cells)
    close = np.isclose(got, ref, rtol=1e-8, atol=1e-12, equal_nan=True)
    way = modes[cells]
    print(f"config 5 at full size, sigma {sigma}: tiled cells {tally[2]}, re-run literally {tally[0]}, flagged beyond the re-run "
          f"{tally[1]}; root merge ({snap['left'].shape[0]} x {snap['right'].shape[0]} cells): of its cells "
          f"{(modes == 0).mean():.4f} took the histogram way unflagged, {(modes == 1).mean():.4f} the re-run, "
          f"{(modes == 2).mean():.4f} were flagged beyond it; of 192 sampled cells {close.mean():.4f} equal the oracle on the same "
          f"inputs ({close[way == 2].mean() if (way == 2).any() else float('nan'):.3f} of the {int((way == 2).sum())} beyond the re-run)")
    # every sampled cell the re-run took is the oracle's bit for bit; every well-conditioned one to rounding; what parts is
    # confined to the ill-conditioned cells with more significant pairs than the re-run holds (adjust_shift_variance)
    assert np.array_equal(got[way == 1], ref[way == 1])
    assert close[way != 2].all(), int((~close[way != 2]).sum())
    assert close.mean() >= 0.99, close.mean()
    assert (modes == 2).mean() < 0.01


@pytest.mark.parametrize("sigma,known_worst", [(1.0, None), (0.3, 13)])
def test_full_size_config5_cells_beyond_the_rerun_by_mode(oracle, full5, sigma, known_worst):
    """the cells the tiled adjust_shift_variance FLAGS but does not re-run (mode 2: ill-conditioned, more
    significant pairs than the re-run's lists hold -- 2.5 % of config 5's cells at sigma 1, next to none of them in the root
    merge) sampled where they ARE: a first full-size run records every merge's tallies (testing hook "asv_modes"), the merge
    with the most such cells is snapshotted by a second run, and 96 of its mode-2 cells + 64 others go through the oracle on
    the same inputs.  What is asserted is what DESIGN.md section 2 claims: every re-run cell bit-equal, every unflagged cell
    equal to rounding; the agreement of the mode-2 cells is MEASURED and printed (and bounded from below by what was seen)."""
    import batchelor_amd as bx
    from batchelor_amd import _lib
    from bench import WORKLOADS
    from batchelor_amd.merge_tree import resolve_merge_order
    sizes, d, k, runs = full5
    cfg, _, _, _, tree = WORKLOADS["config5"]
    B = synth_batches(cfg, sizes, d)
    code = resolve_merge_order(len(sizes), tree)
    eng = bx.MnnEngine()
    eng.upload(B)
    _lib.dev_set("asv_modes", 600000)
    try:
        # (sigma 0.3 re-runs 58 % of the cells with long lists: 100 s a run -- the merge that holds the most flagged cells is
        # known from a recorded run and checked against this run's tallies, so that one run serves)
        if known_worst is not None:
            eng.set_snapshot(known_worst)
        eng.run(k=k, merge_tree=code, var_adj=True, sigma=sigma)
        tallies = eng.var_adj_tally()
        worst = int(np.argmax([t["beyond"] for t in tallies]))
        if known_worst is None:
            eng.set_snapshot(worst)
            eng.run(k=k, merge_tree=code, var_adj=True, sigma=sigma)
        else:
            assert worst == known_worst, (worst, [t["beyond"] for t in tallies])
        snap = eng.snapshot_var_adj()
        modes = eng.snapshot_var_adj_modes(snap["right"].shape[0])
    finally:
        _lib.dev_set("asv_modes", 0)
        eng.close()
    tot = {key: sum(max(t[key], 0) for t in tallies) for key in ("rerun", "beyond", "tiled")}
    assert tot["tiled"] == sum(t["tiled"] for t in tallies) and tot["tiled"] > 1000000
    assert set(np.unique(modes).tolist()) <= {0, 1, 2}
    rng = np.random.default_rng(515)
    m2, rest = np.flatnonzero(modes == 2), np.flatnonzero(modes != 2)
    pick2 = rng.choice(m2, min(96, m2.size), replace=False) if m2.size else np.zeros(0, dtype=np.int64)
    cells = np.sort(np.concatenate([pick2, rng.choice(rest, min(64, rest.size), replace=False)])).astype(np.int32)
    got = snap["scaling"][cells]
    ref = oracle.adjust_shift_variance(snap["left"].T, snap["right"].T, snap["correction"], sigma, snap["restrict1"],
                                       snap["restrict2"], cells=cells)
    close = np.isclose(got, ref, rtol=1e-8, atol=1e-12, equal_nan=True)
    way = modes[cells]
    agree2 = float(close[way == 2].mean()) if (way == 2).any() else float("nan")
    print(f"config 5 at full size, sigma {sigma}: of {tot['tiled']} cells {tot['rerun']} re-run, {tot['beyond']} flagged beyond the "
          f"re-run ({tot['beyond'] / tot['tiled']:.4f}); per merge beyond: {[t['beyond'] for t in tallies]}; merge {worst} "
          f"({snap['left'].shape[0]} x {snap['right'].shape[0]} cells) holds {(modes == 2).sum()} of them: of {int((way == 2).sum())} "
          f"sampled {agree2:.4f} equal the oracle on the same inputs; of the {int((way != 2).sum())} other sampled cells "
          f"{close[way != 2].mean():.4f}")
    assert np.array_equal(got[way == 1], ref[way == 1])
    assert close[way == 0].all(), int((~close[way == 0]).sum())
    if sigma >= 1.0 and (way == 2).sum() >= 32:
        assert agree2 >= 0.9, agree2   # (the mildly ill-conditioned cells of sigma 1: the histogram quantile is the reference's)
