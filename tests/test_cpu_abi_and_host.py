"""CPU-side checks (no GPU): the C ABI library loads and exports every symbol include/*.h declares, the host-side
mirror of the reference's merge-tree / batch-splitting logic matches the reference's known answers, and the product
path refuses to run without its HIP device instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from batchelor_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "batchelor_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(bmx_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 25, names
    lib = built.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_gpu_means_loud_failure_not_fallback(built):
    import batchelor_amd as bx
    if bx.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(bx.BatchelorMI355XError, match="no CPU fallback"):
        bx.reducedMNN(np.zeros((5, 2)), np.ones((5, 2)))
    with pytest.raises(bx.BatchelorMI355XError, match="no CPU fallback"):
        bx.query_knn(np.zeros((5, 2)), np.zeros((5, 2)), 1)
    # nothing in the product imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "batchelor_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f


def test_shard_range_partitions_rows(built):
    from batchelor_amd.dist import shard_range
    for n in (0, 1, 7, 100000, 100001):
        for world in (1, 2, 3, 8):
            covered = []
            per = (n + world - 1) // world
            for r in range(world):
                b, e = shard_range(n, r, world)
                assert 0 <= b <= e <= n and e - b <= per
                assert b == min(n, r * per)            # slices are the padded, contiguous layout of the exchange
                covered.extend(range(b, e))
            assert covered == list(range(n))


def test_host_thread_pool_copies_exactly_under_concurrent_callers(built):
    """The pool behind the pinned staging ring (csrc/host_xfer.hpp): jobs of every size class, from several caller threads at
    once (ctypes releases the GIL), bursts of short jobs back to back -- every byte where it belongs, none beyond."""
    import threading
    lib = built.lib()
    lib.bmx_dev_host_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    sizes = [0, 1, 4097, 128 << 10, (128 << 10) + 1, 700001, (8 << 20) - 3, (8 << 20) + 5, 23 << 20]
    errors = []

    def caller(seed):
        rng = np.random.default_rng(seed)
        for rep in range(40):
            n = sizes[int(rng.integers(len(sizes)))] if rep % 4 else int(rng.integers(1, 3 << 20))
            src = rng.integers(0, 256, n + 64, dtype=np.uint8)
            dst = np.full(n + 64, 0xA5, dtype=np.uint8)
            rc = lib.bmx_dev_host_copy(dst.ctypes.data + 32, src.ctypes.data + 32, n)
            if rc != 0 or not np.array_equal(dst[32:32 + n], src[32:32 + n]) or (dst[:32] != 0xA5).any() or (dst[32 + n:] != 0xA5).any():
                errors.append((seed, rep, n, rc))

    threads = [threading.Thread(target=caller, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    assert lib.bmx_dev_host_copy(None, None, 5) != 0 and lib.bmx_dev_host_copy(None, None, 0) == 0


# ---------------------------------------------------------------- tests/testthat/test-tree.R:4-104 on the PRODUCT code
def test_product_binarize_tree_kats():
    from batchelor_amd.merge_tree import binarize_tree, encode_postorder, resolve_merge_order
    assert binarize_tree([1, 2, 3]) == [[1, 2], 3]
    assert binarize_tree([1, 2, 3, 4, 5]) == [[[[1, 2], 3], 4], 5]
    assert binarize_tree([[1, 2, 3]]) == [[1, 2], 3]
    assert binarize_tree([[1], [2]]) == [1, 2]
    assert binarize_tree([[1, 2, 3], [4, 5, 6]]) == [[[1, 2], 3], [[4, 5], 6]]
    assert binarize_tree([[np.arange(1, 4)], [np.arange(4, 7)]]) == [[[1, 2], 3], [[4, 5], 6]]
    ref = [[[1, 2], [3, 4]], [[5, 6], [7, 8]]]
    assert binarize_tree(ref) == ref
    with pytest.raises(ValueError, match="node with no children"):
        binarize_tree([[], [1, 2, 3], [4, 5, 6]])
    assert resolve_merge_order(3) == [[1, 2], 3]
    assert resolve_merge_order(3, [3, 2, 1]) == [[3, 2], 1]
    assert resolve_merge_order(4, [[1, 4], [3, 2]]) == [[1, 4], [3, 2]]
    assert resolve_merge_order(3, ["A", "B", "C"], names=["A", "B", "C"]) == resolve_merge_order(3, [1, 2, 3])
    assert resolve_merge_order(4, [["a", "d"], ["c", "b"]], names=list("abcd")) == [[1, 4], [3, 2]]
    for bad, names in (([1, 2, 3], None), ([1, 1], None), (["A", "C"], ["A", "B"])):
        with pytest.raises(ValueError, match="invalid leaf nodes specified in 'merge.order'"):
            resolve_merge_order(2, bad, names)
    assert encode_postorder([[1, 2], 3]).tolist() == [1, 2, 0, 3, 0]
    assert encode_postorder([[1, 4], [3, 2]]).tolist() == [1, 4, 0, 3, 2, 0, 0]


def test_product_divide_into_batches_matches_oracle(oracle):
    from batchelor_amd.reduced_mnn import divideIntoBatches, _reindex_pairings
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 3))
    batch = rng.choice(["b", "a", "c"], 50)
    mask = rng.random(50) < 0.7
    d = divideIntoBatches(x, batch, mask)
    ob, lev, reo, rst = oracle.divide_into_batches(x, batch, mask)
    assert d["levels"] == lev and np.array_equal(d["reorder"], reo)
    for a, b in zip(d["batches"], ob):
        assert np.array_equal(a, b)
    for a, b in zip(d["restricted"], rst):
        assert np.array_equal(a, b)
    S = rng.permutation(40) + 1
    pairings = [(rng.integers(1, 11, 20), np.arange(11, 31))]
    out = _reindex_pairings(pairings, S)
    assert np.array_equal(S[out[0][0] - 1], pairings[0][0]) and np.array_equal(S[out[0][1] - 1], pairings[0][1])
