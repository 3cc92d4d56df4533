"""CPU restatement (numpy + oracle/mnn_oracle.c) of batchelor's fastMNN / reducedMNN merge engine.

TEST INFRASTRUCTURE ONLY -- the product package (batchelor_amd/) never imports this module.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker / timed CPU baseline.

Every function cites the reference lines it follows (paths relative to /root/reference).  R conventions are kept:
matrices are cells x dims, cell / batch / pair indices are 1-based, `None` stands for R's NULL.

Parity status (see DESIGN.md "Oracle"): R is absent from the build image, so nothing here was compared with an
actual R run.  It is pinned by the reference's own known-answer tests (tests/testthat/test-reduced-mnn.R:80-105,
test-tree.R:4-104, test-utils.R:82-152) and its executable specifications (test-fast-mnn.R:7-92), re-expressed in
tests/test_oracle_*.py.  The third-party exact kNN (BiocNeighbors, version unpinned by DESCRIPTION:17) is restated
from its contract; tie order there is "parity unpinned" and fixed here as (distance, lowest index).
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f64p = ctypes.POINTER(ctypes.c_double)

ORC_MESSAGES = {
    -2: "number of genes do not match up between matrices",
    -3: "number of cells do not match up between matrices",
    -4: "subset indices out of range",
    -5: "'index' must have length equal to number of rows in 'averaged'",
    -6: "out of memory",
}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmnn_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------------------------------
# Third-party contract: BiocNeighbors::queryKNN / findMutualNN (call sites R/MNN_tree.R:129, R/fastMNN.R:605)
# --------------------------------------------------------------------------------------------------
def query_knn(X, query, k, nthreads=0):
    """queryKNN(X, query, k): index [nq x k] 1-based into rows of X, distance [nq x k] Euclidean, ascending."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    query = np.ascontiguousarray(query, dtype=np.float64)
    nr, d = X.shape
    nq = query.shape[0]
    k = int(min(k, nr))
    idx = np.zeros((nq, k), dtype=np.int32)
    dist = np.zeros((nq, k), dtype=np.float64)
    if k > 0 and nq > 0:
        rc = lib().orc_knn(_p(X, c_f64p), nr, _p(query, c_f64p), nq, d, k, _p(idx, c_i32p), _p(dist, c_f64p),
                           int(nthreads))
        if rc:
            raise RuntimeError(ORC_MESSAGES.get(rc, str(rc)))
    return idx + 1, dist


def find_mutual_nns(left, right):
    """src/find_mutual_nns.cpp:8-41.  left [nL x k2], right [nR x k1], 1-based -> (first, second) 1-based."""
    left = np.asfortranarray(left, dtype=np.int32)
    right = np.asfortranarray(right, dtype=np.int32)
    nL, k2 = left.shape
    nR, k1 = right.shape
    outL = np.zeros(max(1, nL * k2), dtype=np.int32)
    outR = np.zeros(max(1, nL * k2), dtype=np.int32)
    n = ctypes.c_int64(0)
    rc = lib().orc_find_mutual_nns(_p(left, c_i32p), nL, k2, _p(right, c_i32p), nR, k1, _p(outL, c_i32p),
                                   _p(outR, c_i32p), ctypes.byref(n))
    if rc:
        raise RuntimeError(ORC_MESSAGES.get(rc, str(rc)))
    return outL[: n.value].copy(), outR[: n.value].copy()


def find_mutual_nn(data1, data2, k1, k2, nthreads=0):
    """findMutualNN(data1, data2, k1, k2): for each cell of data1 its k2 nearest in data2, for each cell of data2
    its k1 nearest in data1, then the mutual intersection (in-tree spec: src/find_mutual_nns.cpp)."""
    idx12, _ = query_knn(data2, data1, k2, nthreads)  # rows: data1 cells -> ids in data2
    idx21, _ = query_knn(data1, data2, k1, nthreads)  # rows: data2 cells -> ids in data1
    return find_mutual_nns(idx12, idx21)


# --------------------------------------------------------------------------------------------------
# Legacy natives (classic mnnCorrect)
# --------------------------------------------------------------------------------------------------
def smooth_gaussian_kernel(averaged, index, mat, sigma2):
    """src/smooth_gaussian_kernel.cpp:11-118.  averaged [g x U], index [U] 0-based, mat [gd x n] -> [g x n]."""
    averaged = np.asfortranarray(averaged, dtype=np.float64)
    mat = np.asfortranarray(mat, dtype=np.float64)
    index = np.ascontiguousarray(index, dtype=np.int32)
    g, U = averaged.shape
    gd, n = mat.shape
    out = np.zeros((g, n), dtype=np.float64, order="F")
    rc = lib().orc_smooth_gaussian_kernel(_p(averaged, c_f64p), g, U, _p(index, c_i32p), index.size, _p(mat, c_f64p),
                                          gd, n, ctypes.c_double(sigma2), _p(out, c_f64p))
    if rc:
        raise RuntimeError(ORC_MESSAGES.get(rc, str(rc)))
    return out


def adjust_shift_variance(data1, data2, vect, sigma2, restrict1, restrict2, cells=None):
    """src/adjust_shift_variance.cpp:30-164.  data1 [g x n1], data2 [g x n2], vect [n2 x g], restrict* 0-based.
    `cells` (0-based) evaluates only those cells of data2 -- the loop at :51 treats every cell on its own."""
    data1 = np.asfortranarray(data1, dtype=np.float64)
    data2 = np.asfortranarray(data2, dtype=np.float64)
    vect = np.asfortranarray(vect, dtype=np.float64)
    r1 = np.ascontiguousarray(restrict1, dtype=np.int32)
    r2 = np.ascontiguousarray(restrict2, dtype=np.int32)
    cl = None if cells is None else np.ascontiguousarray(cells, dtype=np.int32)
    out = np.zeros(data2.shape[1] if cl is None else cl.size, dtype=np.float64)
    rc = lib().orc_adjust_shift_variance_cells(_p(data1, c_f64p), data1.shape[0], data1.shape[1], _p(data2, c_f64p),
                                               data2.shape[0], data2.shape[1], _p(vect, c_f64p), vect.shape[0],
                                               vect.shape[1], ctypes.c_double(sigma2), _p(r1, c_i32p), r1.size,
                                               _p(r2, c_i32p), r2.size, None if cl is None else _p(cl, c_i32p),
                                               0 if cl is None else cl.size, _p(out, c_f64p))
    if rc:
        raise RuntimeError(ORC_MESSAGES.get(rc, str(rc)))
    return out


# --------------------------------------------------------------------------------------------------
# Numeric primitives of the merge loop
# --------------------------------------------------------------------------------------------------
def choose_k(k, prop_k, N):
    """R/MNN_tree.R:140-146.  R's round() is half-to-even, as is Python's."""
    if prop_k is None:
        return int(k)
    return int(min(N, max(k, round(prop_k * N))))


def restricted_mnn(left_data, left_restrict, right_data, right_restrict, k, prop_k=None, nthreads=0):
    """R/MNN_tree.R:113-138."""
    ld = left_data if left_restrict is None else left_data[np.asarray(left_restrict) - 1]
    rd = right_data if right_restrict is None else right_data[np.asarray(right_restrict) - 1]
    k1 = choose_k(k, prop_k, ld.shape[0])
    k2 = choose_k(k, prop_k, rd.shape[0])
    first, second = find_mutual_nn(ld, rd, k1=k1, k2=k2, nthreads=nthreads)
    if left_restrict is not None:
        first = np.asarray(left_restrict, dtype=np.int32)[first - 1]
    if right_restrict is not None:
        second = np.asarray(right_restrict, dtype=np.int32)[second - 1]
    return first, second


def average_correction(refdata, mnn1, curdata, mnn2):
    """R/fastMNN.R:567-580: per-pair vectors summed by `rowsum` (groups ascending, rows in pair order) / counts."""
    mnn1 = np.asarray(mnn1, dtype=np.int64)
    mnn2 = np.asarray(mnn2, dtype=np.int64)
    d = curdata.shape[1]
    if mnn2.size == 0:
        return np.zeros((0, d)), np.zeros(0, dtype=np.int32)
    corvec = refdata[mnn1 - 1] - curdata[mnn2 - 1]
    second, inv = np.unique(mnn2, return_inverse=True)
    summed = np.zeros((second.size, d))
    np.add.at(summed, inv, corvec)  # unbuffered: adds rows in pair order, like rowsum
    npairs = np.bincount(inv, minlength=second.size)
    return summed / npairs[:, None], second.astype(np.int32)


def get_batch_magnitude(correction, ave=None):
    """R/fastMNN.R:582-595."""
    if ave is None:
        ave = correction.mean(axis=0)
    ave_l2sq = float(np.sum(np.mean(correction ** 2, axis=0)))
    if ave_l2sq == 0:
        return 0.0
    return math.sqrt(float(np.sum(ave ** 2)) / ave_l2sq)


def center_along_batch_vector(mat, batch_vec, restrict=None):
    """R/fastMNN.R:626-640."""
    batch_vec = np.asarray(batch_vec, dtype=np.float64)
    batch_vec = batch_vec / math.sqrt(float(np.sum(batch_vec ** 2)))
    loc = mat @ batch_vec
    central = loc.mean() if restrict is None else loc[np.asarray(restrict) - 1].mean()
    return mat + np.outer(central - loc, batch_vec)


def orthogonalize_other(data, restrict, vectors):
    """R/fastMNN.R:642-647."""
    for vec in vectors:
        data = center_along_batch_vector(data, vec, restrict=restrict)
    return data


def compute_perbatch_var(data, index, origin):
    """R/fastMNN.R:651-658: per original batch, sum over dims of the sample variance (n-1)."""
    out = np.zeros(len(index))
    for i, b in enumerate(index):
        rows = data[origin == b]
        out[i] = np.sum(np.var(rows, axis=0, ddof=1)) if rows.shape[0] > 1 else np.nan
    return out


def compute_tricube_average(vals, indices, distances, bandwidth=None, ndist=3):
    """R/utils_tricube.R:1-27.  indices 1-based into rows of `vals`."""
    indices = np.asarray(indices)
    distances = np.asarray(distances, dtype=np.float64)
    nk = indices.shape[1]
    if nk == 0:  # R: output stays the scalar 0 -> matrix(0, nrow(vals), ncol(vals))
        return np.zeros(vals.shape)
    if bandwidth is None:
        middle = int(math.ceil(nk / 2))
        bandwidth = distances[:, middle - 1] * ndist
    bandwidth = np.maximum(1e-8, bandwidth)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = distances / bandwidth[:, None]
        rel[rel > 1] = 1
        tricube = (1 - rel ** 3) ** 3
        weight = tricube / tricube.sum(axis=1)[:, None]
    out = np.zeros((indices.shape[0], vals.shape[1]))
    for kdx in range(nk):
        out = out + vals[indices[:, kdx] - 1] * weight[:, kdx][:, None]
    return out


def tricube_weighted_correction(curdata, correction, in_mnn, k=20, ndist=3, nthreads=0, var_adj=None):
    """R/fastMNN.R:599-608.  var_adj = (refdata, sigma, restrict1, restrict2) additionally rescales every cell's
    correction vector as mnnCorrect(var.adj=TRUE) does (R/mnnCorrect.R:331-342, .adjust_shift_variance :462-481:
    pmax(scaling, 1) * correction) -- BASELINE.json configs[4]; fastMNN() itself has no such switch."""
    cur_uniq = curdata[np.asarray(in_mnn) - 1]
    safe_k = min(k, cur_uniq.shape[0])
    idx, dist = query_knn(cur_uniq, curdata, safe_k, nthreads)
    corr = compute_tricube_average(correction, idx, dist, ndist=ndist)
    if var_adj is not None:
        refdata, sigma, r1, r2 = var_adj
        r1 = np.arange(refdata.shape[0]) if r1 is None else np.asarray(r1) - 1
        r2 = np.arange(curdata.shape[0]) if r2 is None else np.asarray(r2) - 1
        scaling = adjust_shift_variance(refdata.T, curdata.T, corr, sigma, r1, r2)
        with np.errstate(invalid="ignore"):
            scaling = np.where(scaling < 1, 1.0, scaling)  # pmax(scaling, 1): NaN stays NaN
        corr = scaling[:, None] * corr
    return curdata + corr


def combine_restrict(left_data, left_restrict, right_data, right_restrict):
    """R/fastMNN.R:610-622."""
    if left_restrict is None and right_restrict is None:
        return None
    if left_restrict is None:
        left_restrict = np.arange(1, left_data.shape[0] + 1)
    if right_restrict is None:
        right_restrict = np.arange(1, right_data.shape[0] + 1)
    return np.concatenate([np.asarray(left_restrict), np.asarray(right_restrict) + left_data.shape[0]]).astype(np.int32)


# --------------------------------------------------------------------------------------------------
# Re-ordering utilities (R/utils_reorder.R)
# --------------------------------------------------------------------------------------------------
def restore_original_order(batch_ordering, ncells_per_batch):
    """R/utils_reorder.R:1-20."""
    batch_ordering = list(batch_ordering)
    ncells_per_batch = list(ncells_per_batch)
    if len(batch_ordering) != len(ncells_per_batch):
        raise ValueError("length of batch information vectors are not equal")
    if not batch_ordering:
        return np.zeros(0, dtype=np.int64)
    reorder: List[Any] = [None] * len(batch_ordering)
    last = 0
    for idx in batch_ordering:
        n = int(ncells_per_batch[int(idx) - 1])
        reorder[int(idx) - 1] = last + np.arange(1, n + 1)
        last += n
    return np.concatenate(reorder).astype(np.int64)


def reindex_pairings(pairings, new_order):
    """R/utils_reorder.R:23-36.  pairings: list of (left, right) 1-based arrays."""
    new_order = np.asarray(new_order, dtype=np.int64)
    rev = np.zeros(new_order.size + 1, dtype=np.int64)
    rev[new_order] = np.arange(1, new_order.size + 1)
    return [(rev[np.asarray(l, dtype=np.int64)], rev[np.asarray(r, dtype=np.int64)]) for l, r in pairings]


# --------------------------------------------------------------------------------------------------
# Merge tree (R/MNN_tree.R:2-109).  R lists are Python lists; leaves are ints or strings.
# --------------------------------------------------------------------------------------------------
@dataclass
class TreeNode:
    """MNN_treenode, R/MNN_tree.R:2-6."""
    index: List[int]
    data: np.ndarray
    restrict: Optional[np.ndarray]
    origin: np.ndarray = None
    extras: List[np.ndarray] = field(default_factory=list)

    def __post_init__(self):
        if self.origin is None:
            self.origin = np.repeat(np.asarray(self.index, dtype=np.int32), self.data.shape[0])


def _is_leafspec(x):
    return not isinstance(x, (list, tuple))


def binarize_tree(tree):
    """R/MNN_tree.R:21-46.  A non-list vector of length > 1 (e.g. 1:3) is passed as a tuple/np.ndarray leaf group."""
    if isinstance(tree, np.ndarray):
        tree = [x.item() for x in tree]
    if _is_leafspec(tree):
        return tree
    N = len(tree)
    if N == 1:
        return binarize_tree(tree[0])
    if N == 2:
        return [binarize_tree(tree[0]), binarize_tree(tree[1])]
    if N > 2:
        cur = [binarize_tree(tree[0]), binarize_tree(tree[1])]
        for i in range(2, N):
            cur = [cur, binarize_tree(tree[i])]
        return cur
    raise ValueError("merge tree contains a node with no children")


def _leaves(tree):
    if _is_leafspec(tree):
        return [tree]
    out = []
    for ch in tree:
        out.extend(_leaves(ch))
    return out


def _relist(tree, it):
    if _is_leafspec(tree):
        return next(it)
    return [_relist(ch, it) for ch in tree]


def resolve_merge_tree(nbatches, merge_order=None, names=None):
    """R/MNN_tree.R:80-107 without the data fill: returns the binary tree with 1-based integer leaves."""
    if merge_order is None:
        merge_order = list(range(1, nbatches + 1))
    if isinstance(merge_order, np.ndarray):
        merge_order = [x.item() for x in merge_order]
    is_flat = isinstance(merge_order, (list, tuple)) and all(_is_leafspec(x) for x in merge_order)
    # A plain vector (`merge.order=c(3,1,2)`) is a progressive merge; a flat Python list plays that role
    # (binarize_tree gives the same tree for the equivalent R list(3,1,2)).
    if is_flat and len(merge_order) > 1:
        tree = [merge_order[0], merge_order[1]]
        for x in merge_order[2:]:
            tree = [tree, x]
    else:
        tree = merge_order
    tree = binarize_tree(tree)
    leaves = _leaves(tree)
    if all(isinstance(x, (int, np.integer, float)) and not isinstance(x, bool) for x in leaves):
        resolved = [int(x) for x in leaves]
    else:
        lookup = {} if names is None else {str(n): i + 1 for i, n in enumerate(names)}
        resolved = [lookup.get(str(x), None) for x in leaves]
    if (any(x is None for x in resolved) or len(set(resolved)) != len(resolved)
            or any(x < 1 or x > nbatches for x in resolved)):
        raise ValueError("invalid leaf nodes specified in 'merge.order'")
    return _relist(tree, iter(resolved))


def fill_tree(tree, batches, restrict):
    """R/MNN_tree.R:48-59."""
    if _is_leafspec(tree):
        r = None if restrict is None else restrict[tree - 1]
        return TreeNode(index=[tree], data=batches[tree - 1], restrict=None if r is None else np.asarray(r))
    if len(tree) != 2:
        raise ValueError("merge tree structure should contain two children per node")
    return [fill_tree(tree[0], batches, restrict), fill_tree(tree[1], batches, restrict)]


def create_tree_predefined(batches, restrict, merge_order, names=None):
    """R/MNN_tree.R:80-109."""
    return fill_tree(resolve_merge_tree(len(batches), merge_order, names), batches, restrict)


def get_next_merge(tree, path=()):
    """R/MNN_tree.R:61-69."""
    if not isinstance(tree[0], list) and not isinstance(tree[1], list):
        return tree[0], tree[1], path
    if isinstance(tree[1], list):
        return get_next_merge(tree[1], path + (1,))
    return get_next_merge(tree[0], path + (0,))


def update_tree(tree, path, node):
    """R/MNN_tree.R:71-77."""
    if len(path) == 0:
        return node
    tree = list(tree)
    tree[path[0]] = update_tree(tree[path[0]], path[1:], node)
    return tree


# --------------------------------------------------------------------------------------------------
# Auto-merge (R/MNN_tree.R:154-226)
# --------------------------------------------------------------------------------------------------
def _count_mnn_pairs(left, remainders, upto, k, prop_k, nthreads):
    """R/MNN_tree.R:171-193.  NB: left.data keeps the orthogonalisations of earlier j (as upstream does)."""
    left_data = left.data
    n = np.zeros(upto, dtype=np.int64)
    for j in range(upto):
        right = remainders[j]
        right_data = orthogonalize_other(right.data, right.restrict, left.extras)
        left_data = orthogonalize_other(left_data, left.restrict, right.extras)
        f, _ = restricted_mnn(left_data, left.restrict, right_data, right.restrict, k, prop_k, nthreads)
        n[j] = f.size
    return n


def initialize_auto_search(batches, restrict, k, prop_k, nthreads):
    """R/MNN_tree.R:154-168."""
    rem = [TreeNode(index=[i + 1], data=batches[i], restrict=None if restrict is None or restrict[i] is None
                    else np.asarray(restrict[i])) for i in range(len(batches))]
    B = len(rem)
    collected = np.zeros((B, B), dtype=np.int64)
    for i in range(B):
        collected[i, :i] = _count_mnn_pairs(rem[i], rem, i, k, prop_k, nthreads)
    return rem, collected


def pick_best_merge(rem, stats):
    """R/MNN_tree.R:196-202: which(stats==max(stats), arr.ind=TRUE)[1,] = first max in column-major order."""
    hits = np.argwhere(stats.T == stats.max())  # rows of (col, row), column-major scan
    col, row = hits[0]
    return rem[row], rem[col], (int(row), int(col))


def update_remainders(rem, stats, chosen, node, k, prop_k, nthreads):
    """R/MNN_tree.R:205-226."""
    keep = [i for i in range(len(rem)) if i not in chosen]
    rem = [rem[i] for i in keep]
    if not rem:
        return node, None
    old = stats[np.ix_(keep, keep)]
    new_stats = _count_mnn_pairs(node, rem, len(rem), k, prop_k, nthreads)
    new = np.vstack([old, new_stats[None, :]])
    new = np.hstack([new, np.zeros((new.shape[0], 1), dtype=np.int64)])
    return rem + [node], new


# --------------------------------------------------------------------------------------------------
# The merge engine: .fast_mnn / .fast_mnn_core (R/fastMNN.R:398-562)
# --------------------------------------------------------------------------------------------------
@dataclass
class MergeInfo:
    left: List[List[int]]
    right: List[List[int]]
    pairs: List[Any]  # list of (left, right) 1-based output-row indices
    batch_size: np.ndarray
    skipped: np.ndarray
    lost_var: np.ndarray


@dataclass
class FastMnnResult:
    corrected: np.ndarray
    batch: np.ndarray
    merge_info: MergeInfo


def fast_mnn(batches: Sequence[np.ndarray], k=20, prop_k=None, restrict=None, ndist=3, merge_order=None,
             auto_merge=False, min_batch_skip=0.0, names=None, nthreads=0, var_adj=False, sigma=0.1) -> FastMnnResult:
    """R/fastMNN.R:398-562 (.fast_mnn + .fast_mnn_core).  `min_batch_skip=None` is R's NA."""
    batches = [np.ascontiguousarray(b, dtype=np.float64) for b in batches]
    nbatches = len(batches)
    if nbatches < 2:
        raise ValueError("at least two batches must be specified")
    if names is not None and len(set(names)) != len(names):
        raise ValueError("names of batches should be unique")
    if not auto_merge:
        tree = create_tree_predefined(batches, restrict, merge_order, names)
        stats = None
    else:
        tree, stats = initialize_auto_search(batches, restrict, k, prop_k, nthreads)

    nmerges = nbatches - 1
    pairings, left_set, right_set = [], [], []
    batch_size = np.full(nmerges, np.nan)
    skipped = np.zeros(nmerges, dtype=bool)
    var_kept = np.ones((nmerges, nbatches))

    for mdx in range(nmerges):
        if not auto_merge:
            left, right, chosen = get_next_merge(tree)
        else:
            left, right, chosen = pick_best_merge(tree, stats)
        left_data, right_data = left.data, right.data

        left_old = compute_perbatch_var(left_data, left.index, left.origin)  # :467-468
        right_old = compute_perbatch_var(right_data, right.index, right.origin)
        left_set.append(list(left.index))
        right_set.append(list(right.index))

        right_data = orthogonalize_other(right_data, right.restrict, left.extras)  # :473-474
        left_data = orthogonalize_other(left_data, left.restrict, right.extras)

        first, second = restricted_mnn(left_data, left.restrict, right_data, right.restrict, k, prop_k, nthreads)
        if first.size == 0:
            # colMeans of a 0-row matrix is NaN and `if (NaN == 0)` is an R error (R/fastMNN.R:588-589)
            raise RuntimeError("no mutual nearest neighbours found between batches")
        averaged, _ = average_correction(left_data, first, right_data, second)  # :480-481
        overall = averaged.mean(axis=0)

        do_correct = True
        if min_batch_skip is not None and not (isinstance(min_batch_skip, float) and math.isnan(min_batch_skip)):
            mag = get_batch_magnitude(averaged, overall)  # :484-492
            batch_size[mdx] = mag
            if mag < min_batch_skip:
                do_correct = False
                skipped[mdx] = True

        if do_correct:
            left_data = center_along_batch_vector(left_data, overall, left.restrict)  # :496-497
            right_data = center_along_batch_vector(right_data, overall, right.restrict)
            left_new = compute_perbatch_var(left_data, left.index, left.origin)
            right_new = compute_perbatch_var(right_data, right.index, right.origin)
            to_add = [overall]
            re_avg, re_second = average_correction(left_data, first, right_data, second)  # :505-507
            right_data = tricube_weighted_correction(right_data, re_avg, re_second,
                                                     k=choose_k(k, prop_k, right_data.shape[0]), ndist=ndist,
                                                     nthreads=nthreads,
                                                     var_adj=(left_data, sigma, left.restrict, right.restrict)
                                                     if var_adj else None)
        else:
            to_add = []
            left_new = compute_perbatch_var(left_data, left.index, left.origin)
            right_new = compute_perbatch_var(right_data, right.index, right.origin)

        var_kept[mdx, np.asarray(left.index) - 1] = left_new / left_old  # :516-518
        var_kept[mdx, np.asarray(right.index) - 1] = right_new / right_old
        pairings.append((first.astype(np.int64), second.astype(np.int64)))

        node = TreeNode(index=list(left.index) + list(right.index), data=np.vstack([left_data, right_data]),
                        restrict=combine_restrict(left_data, left.restrict, right_data, right.restrict),
                        origin=np.concatenate([left.origin, right.origin]),
                        extras=list(left.extras) + list(right.extras) + to_add)  # :520-525
        if not auto_merge:
            tree = update_tree(tree, chosen, node)
        else:
            tree, stats = update_remainders(tree, stats, chosen, node, k, prop_k, nthreads)

    full = tree
    full_data, full_order, full_origin = full.data, list(full.index), full.origin

    for mdx in range(nmerges):  # :533-538
        b1 = int(np.argmax(full_origin == left_set[mdx][0]))
        b2 = int(np.argmax(full_origin == right_set[mdx][0]))
        l, r = pairings[mdx]
        pairings[mdx] = (l + b1, r + b2)

    if any(full_order[i] > full_order[i + 1] for i in range(len(full_order) - 1)):  # is.unsorted, :541-547
        ncells = np.bincount(full_origin, minlength=nbatches + 1)[1:]
        ordering = restore_original_order(full_order, ncells)
        full_data = full_data[ordering - 1]
        full_origin = full_origin[ordering - 1]
        pairings = reindex_pairings(pairings, ordering)

    info = MergeInfo(left=left_set, right=right_set, pairs=pairings, batch_size=batch_size, skipped=skipped,
                     lost_var=1 - var_kept)
    return FastMnnResult(corrected=full_data, batch=full_origin.astype(np.int32), merge_info=info)


def divide_into_batches(x, batch, restrict=None):
    """R/divideIntoBatches.R:36-84 with byrow=TRUE.  Levels = sorted unique values (factor())."""
    batch = np.asarray(batch)
    levels = sorted(set(batch.tolist()))
    n = x.shape[0]
    mask = None
    if restrict is not None:
        mask = np.zeros(n, dtype=bool)
        r = np.asarray(restrict)
        if r.dtype == bool:
            mask[:] = r
        else:
            mask[r - 1] = True
    out, restricted = [], ([] if restrict is not None else None)
    reorder = np.zeros(n, dtype=np.int64)
    last = 0
    for b in levels:
        keep = batch == b
        cur = x[keep]
        if mask is not None:
            cr = np.flatnonzero(mask[keep]) + 1
            if cr.size == 0:
                raise ValueError("no cells remaining in a batch after restriction")
            restricted.append(cr.astype(np.int32))
        out.append(cur)
        reorder[keep] = last + np.arange(1, cur.shape[0] + 1)
        last += cur.shape[0]
    return out, levels, reorder, restricted


def reduced_mnn(*batches, batch=None, k=20, prop_k=None, restrict=None, ndist=3, merge_order=None, auto_merge=False,
                min_batch_skip=0.0, names=None, nthreads=0, var_adj=False, sigma=0.1) -> FastMnnResult:
    """R/reducedMNN.R:61-95."""
    if len(batches) == 1:
        r0 = None if restrict is None else restrict[0]
        divided, levels, reorder, restricted = divide_into_batches(np.asarray(batches[0], dtype=np.float64), batch, r0)
        out = fast_mnn(divided, k=k, prop_k=prop_k, restrict=restricted, ndist=ndist, merge_order=merge_order,
                       auto_merge=auto_merge, min_batch_skip=min_batch_skip, names=[str(l) for l in levels], var_adj=var_adj, sigma=sigma,
                       nthreads=nthreads)
        out.corrected = out.corrected[reorder - 1]
        out.batch = out.batch[reorder - 1]
        out.merge_info.pairs = reindex_pairings(out.merge_info.pairs, reorder)
        return out
    return fast_mnn(list(batches), k=k, prop_k=prop_k, restrict=restrict, ndist=ndist, merge_order=merge_order,
                    auto_merge=auto_merge, min_batch_skip=min_batch_skip, names=names, nthreads=nthreads,
                    var_adj=var_adj, sigma=sigma)
