"""GPU checks of the BENCH workload (BASELINE.json configs[2]: 8 batches x 100 000 cells x 50 PCs, progressive
merge 1..8) -- the shape whose merges have a growing left side (up to 700k reference cells), seven accumulated batch
vectors, a threshold sample pass and split reference ranges.

* reduced size (8 x 6000 x 50): the whole result against the CPU oracle, pairs bit-exact;
* a shape that takes the sample pass (reference >= 32 768 rows) and, with the testing hook "force_c", split reference ranges with
  shared thresholds, against the oracle;
* FULL size (8 x 100 000): properties that do not need the oracle to repeat the job.  For three merges the engine
  keeps the two matrices it hands to findMutualNN (bmx_engine_set_snapshot); for a random sample of right cells the
  oracle finds their neighbours in the FULL left matrix and those neighbours' neighbours in the FULL right matrix,
  i.e. the exact MNN pairs of the sampled cells, which must be the engine's pairs of those cells, order included.
  Plus: run-to-run bit identity, the exact-path counter stays small, outputs finite.
"""
import os

import numpy as np
import pytest

from tests.conftest import synth_batches
from tests.test_gpu_engine import assert_same_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bx():
    import batchelor_amd
    return batchelor_amd


def test_config3_reduced_eight_batches_vs_oracle(oracle, bx):
    B = synth_batches(3, [6000] * 8, 50)
    out = bx.reducedMNN(*B)
    ref = oracle.reduced_mnn(*B)
    assert_same_result(out, ref)
    assert len(out.merge_info.pairs) == 7 and all(p[0].size > 0 for p in out.merge_info.pairs)
    assert not out.merge_info.skipped.any()


@pytest.mark.parametrize("force_c", [None, "3"])
def test_sample_pass_and_split_ranges_vs_oracle(oracle, bx, dev, force_c):
    # left = 40 000 cells: the candidate pass runs its threshold sample first; the testing hook "force_c" splits every
    # query block into three reference ranges that share their thresholds through global memory
    B = synth_batches(3, [40000, 3000, 2500], 50)
    if force_c is not None:
        dev("force_c", force_c)
    out = bx.reducedMNN(*B)
    ref = oracle.reduced_mnn(*B)
    assert_same_result(out, ref)


N, D, K, NB = 100_000, 50, 20, 8


@pytest.fixture(scope="module")
def full(bx):
    B = synth_batches(3, [N] * NB, D)
    eng = bx.MnnEngine()
    eng.upload(B)
    runs = {}
    for m in (0, 3, 6):
        eng.set_snapshot(m)
        eng.run(k=K)
        res = eng.download()
        runs[m] = (res, eng.snapshot(), eng.profile()["exact_fallbacks"])
    eng.close()
    return B, runs


def test_full_size_config3_is_deterministic_and_finite(full):
    _, runs = full
    a, b, c = runs[0][0], runs[3][0], runs[6][0]
    for other in (b, c):
        assert np.array_equal(a.corrected, other.corrected)            # three runs, bit for bit
        for (l0, r0), (l1, r1) in zip(a.merge_info.pairs, other.merge_info.pairs):
            assert np.array_equal(l0, l1) and np.array_equal(r0, r1)
        assert np.array_equal(a.merge_info.lost_var, other.merge_info.lost_var)
    assert np.all(np.isfinite(a.corrected))
    assert not a.merge_info.skipped.any() and np.all(a.merge_info.batch_size > 0.2)
    lost = np.asarray(a.merge_info.lost_var)
    assert lost.shape == (NB - 1, NB) and np.all(lost > -1e-9) and np.all(lost < 0.5)
    # every query that is not certified by the candidate pass goes to the exact FP64 path: a handful per million
    for m in runs:
        assert runs[m][2] <= 3000, runs[m][2]
    # pairs: left ids inside the left node's rows, right ids inside the right batch, left ascending
    for m, (pl, pr) in enumerate(a.merge_info.pairs):
        assert pl.min() >= 1 and pl.max() <= (m + 1) * N
        assert pr.min() >= (m + 1) * N + 1 and pr.max() <= (m + 2) * N
        assert np.all(np.diff(pl) >= 0)
        assert np.unique(pl.astype(np.int64) * (NB * N + 1) + pr).size == pl.size          # no duplicate pair


@pytest.mark.parametrize("m", [0, 3, 6])
def test_full_size_config3_pairs_of_sampled_cells_match_oracle(oracle, full, m):
    _, runs = full
    res, (left, right), _ = runs[m]
    assert left.shape == ((m + 1) * N, D) and right.shape == (N, D)
    rng = np.random.default_rng(100 + m)
    rows = np.sort(rng.choice(N, 240, replace=False))                  # sampled right cells (0-based)
    nn_l, _ = oracle.query_knn(left, right[rows], K)                   # their K nearest left cells (1-based)
    cand = np.unique(nn_l)                                             # left cells that could pair with them
    nn_r, _ = oracle.query_knn(right, left[cand - 1], K)               # those cells' K nearest right cells
    back = {int(c): set(row.tolist()) for c, row in zip(cand, nn_r)}
    expect = set()
    for r, row in zip(rows, nn_l):
        for l in row.tolist():
            if int(r) + 1 in back[int(l)]:
                expect.add((int(l), int(r) + 1))
    pl, pr = res.merge_info.pairs[m]
    pr_local = pr - (m + 1) * N                                        # right ids within the right batch
    keep = np.isin(pr_local, rows + 1)
    got = list(zip(pl[keep].tolist(), pr_local[keep].tolist()))
    assert len(got) == len(set(got))
    assert set(got) == expect
    assert len(expect) > 100
    # order: left ascending, and within a left cell by its neighbour rank = ascending distance to the right cells
    order = sorted(expect, key=lambda lr: (lr[0], float(np.sum((left[lr[0] - 1] - right[lr[1] - 1]) ** 2)), lr[1]))
    assert got == order
