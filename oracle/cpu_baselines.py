"""CPU baselines for bench.py (test / bench infrastructure only; never imported by the product).

The reference path itself (R + Rcpp + BiocNeighbors) cannot run in this image, so bench.py times two CPU statements of
its dominant step, the exact kNN inside findMutualNN (R/MNN_tree.R:129) and queryKNN (R/fastMNN.R:605):

A. `kmknn_knn`: ONE thread, pruned exact search in the manner of BiocNeighbors::KmknnParam() -- what fastMNN() runs by
   default (BNPARAM=KmknnParam(), BPPARAM=SerialParam(): R/fastMNN.R:287).  oracle/kmknn_baseline.c.
B. `blas_knn`: ALL cores, blocked brute force on the host BLAS: |q|^2 + |r|^2 - 2 q.r by DGEMM blocks, argpartition,
   exact re-evaluation of the k kept -- the strongest simple CPU formulation of the same search.  The selection
   (argpartition) is single-threaded in numpy and costs more than the DGEMM, so the parallelism is over QUERY BLOCKS: a
   pool of worker threads, each running its blocks with the BLAS pinned to one thread (threadpoolctl); both numbers are
   reported.
"""
from __future__ import annotations

import ctypes
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _kmknn():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.path.join(_HERE, "libkmknn_baseline.so"))
        _lib.kmknn_build.restype = ctypes.c_void_p
        _lib.kmknn_build.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        _lib.kmknn_query.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        _lib.kmknn_free.argtypes = [ctypes.c_void_p]
    return _lib


def kmknn_knn(X, Q, k, iters=5):
    """Exact kNN, single thread.  Returns (idx 1-based [nq x k], dist, stats) with stats = build seconds, query
    seconds and the fraction of reference points whose distance to a query was actually computed."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    n, d = X.shape
    nq = Q.shape[0]
    lib = _kmknn()
    t0 = time.perf_counter()
    tree = lib.kmknn_build(X.ctypes.data, n, d, int(iters))
    if not tree:
        raise MemoryError("kmknn_build")
    t1 = time.perf_counter()
    idx = np.zeros((nq, k), dtype=np.int32)
    dist = np.zeros((nq, k), dtype=np.float64)
    ev = ctypes.c_int64(0)
    rc = lib.kmknn_query(tree, Q.ctypes.data, nq, int(k), idx.ctypes.data, dist.ctypes.data, ctypes.byref(ev))
    t2 = time.perf_counter()
    lib.kmknn_free(tree)
    if rc:
        raise RuntimeError(f"kmknn_query failed ({rc})")
    return idx + 1, dist, {"build_s": t1 - t0, "query_s": t2 - t1, "visited": ev.value / max(1, nq * n)}


def _blas_block(X, rn, q, k, keep):
    v = rn[None, :] - 2.0 * (q @ X.T)
    part = np.argpartition(v, keep - 1, axis=1)[:, :keep]
    diff = X[part] - q[:, None, :]
    d2 = np.einsum("ijk,ijk->ij", diff, diff)
    order = np.lexsort((part, d2), axis=1)[:, :k]
    rows = np.arange(q.shape[0])[:, None]
    return part[rows, order], np.sqrt(d2[rows, order])


def blas_knn(X, Q, k, block=256, workers=None):
    """Exact kNN on the host BLAS.  `workers` threads each take query blocks of `block` rows with the BLAS limited to one
    thread per worker (numpy releases the GIL in the DGEMM, the partition and the gathers).  Returns (idx 1-based, dist,
    info) with info = {"workers", "blas_threads_per_worker", "blas"}."""
    from concurrent.futures import ThreadPoolExecutor
    X = np.ascontiguousarray(X, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    rn = np.einsum("ij,ij->i", X, X)
    nq = Q.shape[0]
    idx = np.zeros((nq, k), dtype=np.int64)
    dist = np.zeros((nq, k), dtype=np.float64)
    keep = min(X.shape[0], k + 8)  # slack for the rounding of the expanded form; the kept ones are re-evaluated exactly
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if workers is None:
        # a block holds block x n doubles twice (values + the partition's indices): bound the pool by ~32 GB of them
        workers = max(1, min(cores, 64, int(32e9 // max(1, 2 * block * X.shape[0] * 8))))
    info = {"workers": workers, "blas_threads_per_worker": 1, "blas": "numpy default"}
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        libs = [i.get("internal_api", "?") + " " + str(i.get("version", "")) for i in threadpool_info() if i.get("user_api") == "blas"]
        info["blas"] = ", ".join(libs) or "numpy default"
        limiter = threadpool_limits(limits=1, user_api="blas")
    except Exception:  # threadpoolctl missing: the BLAS keeps its own threading
        limiter = None
        info["blas_threads_per_worker"] = "library default"
    starts = list(range(0, nq, block))

    def work(b0):
        i, dd = _blas_block(X, rn, Q[b0:b0 + block], k, keep)
        idx[b0:b0 + block] = i
        dist[b0:b0 + block] = dd

    try:
        if workers == 1:
            for b0 in starts:
                work(b0)
        else:
            with ThreadPoolExecutor(workers) as pool:
                list(pool.map(work, starts))
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    return idx + 1, dist, info
