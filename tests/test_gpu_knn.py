"""GPU parity: HIP exact kNN (through the C ABI) vs the CPU oracle -- indices bit-exact, distances bitwise."""
import numpy as np
import pytest

from tests.conftest import synth_batches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nb():
    from batchelor_amd import neighbors
    return neighbors


@pytest.mark.parametrize("nx,nq,d,k", [(2000, 2000, 50, 20), (5000, 3000, 50, 20), (700, 1300, 10, 5),
                                       (4000, 1000, 100, 20), (3000, 500, 20, 30), (300, 300, 2, 1),
                                       # one case per fragment-count class of the split-bf16 kernel (NS = 1 ... 24)
                                       (1500, 900, 4, 10), (1500, 900, 9, 10), (1500, 900, 15, 10),
                                       (1500, 900, 31, 20), (2500, 1200, 41, 20), (2500, 1200, 60, 20),
                                       (2500, 1200, 70, 20), (2500, 1200, 84, 20), (2500, 1200, 120, 20)])
def test_query_knn_matches_oracle(oracle, nb, nx, nq, d, k):
    X, Q = synth_batches(1, [nx, nq], d)
    idx, dist = nb.query_knn(X, Q, k)
    oi, od = oracle.query_knn(X, Q, k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)  # same FP64 summation order -> bitwise
    assert nb.last_knn_exact_fallbacks() <= nq // 100


def test_query_knn_exact_path_and_ties(oracle, nb):
    # duplicated points: every query is a tie; lowest index must win
    core = np.column_stack([np.repeat(np.arange(1, 11), 10), np.tile(np.arange(1, 11), 10)]).astype(np.float64)
    X = np.vstack([core, core, core])
    idx, dist = nb.query_knn(X, core, 3)
    oi, od = oracle.query_knn(X, core, 3)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    # forced exact path on ordinary data gives the same answer as the MFMA path
    X, Q = synth_batches(2, [1500, 900], 50)
    a = nb.query_knn(X, Q, 20)
    nb.set_force_exact_knn(True)
    try:
        b = nb.query_knn(X, Q, 20)
    finally:
        nb.set_force_exact_knn(False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_query_knn_small_and_edge_shapes(oracle, nb):
    X, Q = synth_batches(3, [37, 5], 7)
    for k in (1, 20, 37):
        idx, dist = nb.query_knn(X, Q, k)
        oi, od = oracle.query_knn(X, Q, k)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    with pytest.raises(RuntimeError, match="exceeds"):
        nb.query_knn(X, Q, 38)
    idx, dist = nb.query_knn(X, Q[:0], 3)
    assert idx.shape == (0, 3)


@pytest.mark.parametrize("nx,nq,d,k", [(20000, 3000, 50, 37), (20000, 3000, 50, 50), (20000, 3000, 50, 100),
                                       (30000, 2000, 100, 64), (9000, 1500, 20, 120), (2500, 400, 50, 50)])
def test_query_knn_beyond_the_tiers_lists(oracle, nb, nx, nq, d, k):
    """k > 36 (`prop.k`, R/MNN_tree.R:140-146; k = 50 is an everyday setting): the reference is dealt into ceil(k / 16) strided
    partitions, each searched at k = 36 on the matrix cores, the candidates merged exactly with a certificate that no
    partition's list ends inside the first k (knn.hip: large_k_search).  Same bar as everywhere: indices equal, distances
    bitwise.  (The last shape is too small for partitions of 144 cells: it takes the FP64 scan, as k > 36 did before.)"""
    X, Q = synth_batches(1, [nx, nq], d)
    idx, dist = nb.query_knn(X, Q, k)
    oi, od = oracle.query_knn(X, Q, k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    if nx >= 9000:
        assert nb.last_knn_exact_fallbacks() <= nq // 20        # (queries that went to the FP64 scan)


@pytest.mark.parametrize("nx,nq,d,k", [(40000, 600, 50, 1000), (60000, 400, 20, 2500), (100000, 300, 50, 5000)])
def test_query_knn_k_beyond_900(oracle, nb, nx, nq, d, k):
    """`prop.k` = 0.05 of 100 000 cells is k = 5 000 (R/MNN_tree.R:140-146): more candidates a query (P x 36, P = ceil(k / 14))
    than the rank-counting merges take -- knn.hip: lk_merge_big, the k-th smallest by bisection on the distances' bit patterns,
    only the selected sorted.  Same bar: indices equal, distances bitwise, none of it through the FP64 scan."""
    X, Q = synth_batches(3, [nx, nq], d)
    idx, dist = nb.query_knn(X, Q, k)
    oi, od = oracle.query_knn(X, Q, k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    assert nb.last_knn_exact_fallbacks() <= nq // 20


def test_query_knn_k_beyond_900_with_exact_duplicates(oracle, nb):
    # every reference cell twice: each distance occurs twice, the k-th place splits a pair of equal distances by position
    X0, Q = synth_batches(4, [15000, 200], 30)
    X = np.concatenate([X0, X0])
    idx, dist = nb.query_knn(X, Q, 1001)
    oi, od = oracle.query_knn(X, Q, 1001)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


@pytest.mark.parametrize("nx,nq,d,k,dup", [(3000, 150, 130, 100, 1), (2500, 60, 140, 65, 4), (9000, 40, 128, 2049, 3),
                                           (700, 30, 200, 700, 2), (8192, 20, 129, 8192, 1)])
def test_full_scan_selection_beyond_64_neighbours(oracle, nb, nx, nq, d, k, dup):
    """d > 127: every query takes the FP64 scan; k > 64: its selection is the bisection on the distances' bit patterns
    (knn.hip: knn_exact_select_big).  Each reference cell `dup` times: the k-th place falls among equal distances and is
    decided by position, as the reference's search order decides it."""
    X0, Q = synth_batches(11, [nx // dup, nq], d)
    X = np.concatenate([X0] * dup)
    idx, dist = nb.query_knn(X, Q, k)
    oi, od = oracle.query_knn(X, Q, k)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert nb.last_knn_exact_fallbacks() == nq


@pytest.mark.parametrize("k", [100, 300, 1000])
def test_large_k_partitions_seeded_by_the_first(oracle, nb, dev, k):
    """knn.hip, large_k_search: the partitions (reference rows g with the same g % P) after the first are searched within the
    first's 36th distance.  Same answers with and without; and the case the seed is worthless for: 40 exact copies of a query,
    all in the first partition -- its 36th distance is 0, the other partitions' lists come back empty, fewer than k candidates in
    all: the certificate must send those queries to the exact scan."""
    P = max(2, -(-k // 16))
    if P * 36 > 2048:
        P = -(-k // 14)
    X, Q = synth_batches(21, [30000, 300], 40)
    for j in range(6):
        X[(np.arange(40) + 40 * j) * P] = Q[j]       # rows that are multiples of P: the first partition
    oi, od = oracle.query_knn(X, Q, k)
    for seeded in (1, 0):
        dev("lk_seed", seeded)
        idx, dist = nb.query_knn(X, Q, k)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od), seeded
        if seeded:
            # (the six, and their rows in the other partitions: a seed of 0 certifies nothing, the bounded FP64 sweep finds
            # the empty answer; plain queries: none -- scripts/lk_seed_probe.py)
            assert 6 <= nb.last_knn_exact_fallbacks() <= 6 * P


def test_query_knn_beyond_the_tiers_lists_clustered(oracle, nb):
    # a reference whose ORDER follows its geometry (cells sorted by cluster): the strided deal keeps every partition a fair
    # sample of every cluster; with contiguous partitions a query's neighbours would all sit in one
    rng = np.random.default_rng(77)
    centres = rng.standard_normal((40, 30)) * 4.0
    X = np.concatenate([c + 0.3 * rng.standard_normal((500, 30)) for c in centres])
    Q = centres[rng.integers(0, 40, 1000)] + 0.3 * rng.standard_normal((1000, 30))
    idx, dist = nb.query_knn(X, Q, 80)
    oi, od = oracle.query_knn(X, Q, 80)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    # (500 cells 0.3 apart in 30 dimensions are near-ties for the fp16 pass whatever k is: many of these queries are finished
    # by the bounded FP64 sweep inside their partitions' searches -- the point here is the merge)


def test_query_knn_near_ties_take_the_bounded_exact_path(oracle, nb):
    # clusters of 70 near-duplicates (1e-9 apart), more than either candidate tier keeps per query (32 / 24): neither
    # pass can certify any of these queries, so all of them go through knn_exact_filter / knn_exact_pick and must still
    # match the oracle bit for bit
    rng = np.random.default_rng(5)
    base = rng.standard_normal((70, 20))
    X = np.repeat(base, 70, axis=0) + 1e-9 * rng.standard_normal((4900, 20))
    Q = base[:64] + 1e-9 * rng.standard_normal((64, 20))
    idx, dist = nb.query_knn(X, Q, 20)
    oi, od = oracle.query_knn(X, Q, 20)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert nb.last_knn_exact_fallbacks() >= 32


def test_query_knn_clusters_the_first_tier_certifies(oracle, nb):
    # 30 near-duplicates per cluster: the fp16 tier keeps 32 candidates, the whole cluster, and the gap to the next
    # cluster certifies it; the exact FP64 re-rank orders the duplicates
    rng = np.random.default_rng(5)
    base = rng.standard_normal((150, 20))
    X = np.repeat(base, 30, axis=0) + 1e-9 * rng.standard_normal((4500, 20))
    Q = base[:64] + 1e-9 * rng.standard_normal((64, 20))
    idx, dist = nb.query_knn(X, Q, 20)
    oi, od = oracle.query_knn(X, Q, 20)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


@pytest.mark.parametrize("sample,force_c", [("0", None), ("0", "3"), ("1024", "2")])
def test_query_knn_spill_queue_under_pressure(oracle, nb, dev, sample, force_c):
    # the candidate kernel's consumers hand every survivor to service waves through a 64-record queue.  Without a
    # sampled threshold ("0") a 20 000-row sweep starts with every value a survivor: the queue wraps many times, the
    # consumers wait for room, and lists are compacted while more is appended; with ranges, thresholds are shared
    # through global words as well.  Must still be the oracle's answer, bit for bit.
    dev("sample", sample)
    if force_c:
        dev("force_c", force_c)
    X, Q = synth_batches(9, [20000, 2500], 20)
    idx, dist = nb.query_knn(X, Q, 20)
    oi, od = oracle.query_knn(X, Q, 20)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert nb.last_knn_exact_fallbacks() <= 25


def test_query_knn_random_shapes(oracle, nb, dev):
    # seeded draw of shapes, range counts and sample sizes around the candidate kernel's corner cases: rings shorter
    # than their slot count, one-slot ranges, references just above / below the sampling limits, single queries,
    # k at both list sizes, every fragment-count class
    rng = np.random.default_rng(20250314)
    for case in range(24):
        nx = int(rng.choice([70, 130, 500, 2100, 4095, 4100, 9000, 33000]))
        nq = int(rng.choice([1, 31, 257, 900, 2500]))
        d = int(rng.choice([2, 7, 16, 29, 45, 50, 61, 64, 77, 100, 125]))
        k = int(min(rng.choice([1, 5, 20, 21, 36]), nx))
        force_c = rng.choice(["", "1", "2", "5"])
        sample = rng.choice(["", "0", "1024"])
        dev("force_c", int(force_c) if force_c else 0)
        dev("sample", int(sample) if sample else -1)
        X, Q = synth_batches(100 + case, [nx, nq], d)
        idx, dist = nb.query_knn(X, Q, k)
        oi, od = oracle.query_knn(X, Q, k)
        assert np.array_equal(idx, oi), (case, nx, nq, d, k, force_c, sample)
        assert np.array_equal(dist, od), (case, nx, nq, d, k, force_c, sample)


@pytest.mark.parametrize("nx,nq,d,k,tier", [(2500, 1200, 84, 20, "2"), (2500, 1200, 120, 20, "2"), (1500, 900, 31, 20, "2"),
                                            (2500, 1200, 50, 30, "2"), (3000, 2000, 100, 30, None),
                                            (3000, 2000, 70, 36, None)])
def test_query_knn_second_tier_shapes(oracle, nb, dev, nx, nq, d, k, tier):
    # the split-bf16 kernel runs 8 or 4 consumer waves per workgroup (long rows and long lists: 4), and the host has to
    # size its query blocks accordingly.  It is the first tier for k in 21..36 beyond 61 dimensions; the testing hook
    # "knn_tier" = 2 sends the other shapes through it as well.
    if tier:
        dev("knn_tier", tier)
    X, Q = synth_batches(11, [nx, nq], d)
    idx, dist = nb.query_knn(X, Q, k)
    oi, od = oracle.query_knn(X, Q, k)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_rank_cut_without_the_margin_and_near_ties(oracle, nb, dev):
    """Round 3: the fp16 pass cuts its lists at (k-th best + twice the error bound); the round-2 rule (the KS-th best) is
    what remains when that keeps too much.  Both must give the oracle's rows: (a) the rank cut alone (testing hook
    "no_margin"); (b) the margin cut on data where thousands of references sit within the margin of each query's k-th
    neighbour (clusters of near-duplicates: the margin keeps more than a list holds and the rank cut has to take over in
    mid-sweep)."""
    dev("no_margin", 1)
    for nx, nq, d, k in ((9000, 2500, 50, 20), (20000, 900, 100, 20), (6000, 700, 30, 5)):
        X, Q = synth_batches(77, [nx, nq], d)
        idx, dist = nb.query_knn(X, Q, k)
        oi, od = oracle.query_knn(X, Q, k)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od), (nx, nq, d, k)
    dev("no_margin", 0)
    rng = np.random.default_rng(5150)
    centres = rng.standard_normal((40, 50)) * 2.0
    X = np.repeat(centres, 200, axis=0) + 1e-4 * rng.standard_normal((8000, 50))  # 200 near-duplicates per centre
    Q = centres[rng.integers(0, 40, 600)] + 1e-4 * rng.standard_normal((600, 50))
    idx, dist = nb.query_knn(X, Q, 20)
    oi, od = oracle.query_knn(X, Q, 20)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
