"""CPU baselines for bench.py (test / bench infrastructure only; never imported by the product).

The reference path itself (R + Rcpp + BiocNeighbors) cannot run in this image, so bench.py times two CPU statements of
its dominant step, the exact kNN inside findMutualNN (R/MNN_tree.R:129) and queryKNN (R/fastMNN.R:605):

A. `kmknn_knn`: ONE thread, pruned exact search in the manner of BiocNeighbors::KmknnParam() -- what fastMNN() runs by
   default (BNPARAM=KmknnParam(), BPPARAM=SerialParam(): R/fastMNN.R:287).  oracle/kmknn_baseline.c.
B. `blas_knn`: ALL host threads, blocked brute force on the host BLAS: |r|^2 - 2 q.r by DGEMM tiles of 256 queries x 4 096
   reference cells with a fused running-threshold filter (a query's current (k + 8)-th best: only values below it leave the
   tile; round 4's argpartition over 100 000-wide rows cost more than the DGEMM), exact re-evaluation of the kept -- the
   strongest simple CPU formulation of the same search.  The parallelism is over QUERY BLOCKS: a pool of worker threads, one
   per host thread, each running its blocks with the BLAS pinned to one thread (threadpoolctl).
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


def _blas_block_filtered(X, rn, q, k, keep, chunk):
    """One query block against the reference in chunks of `chunk` rows with a RUNNING THRESHOLD per query: the first chunk is
    partitioned, every later chunk only hands over the values below the query's current keep-th best (a fused filter on the
    DGEMM tile, k log(n / chunk) insertions per query in all) -- no argpartition over n-wide rows."""
    nq, n = q.shape[0], X.shape[0]
    c1 = min(chunk, n)
    v = rn[None, :c1] - 2.0 * (q @ X[:c1].T)
    kk = min(keep, c1)
    part = np.argpartition(v, kk - 1, axis=1)[:, :kk]
    rows = np.arange(nq)[:, None]
    best_v = np.full((nq, keep), np.inf)
    best_i = np.zeros((nq, keep), dtype=np.int64)
    best_v[:, :kk] = v[rows, part]
    best_i[:, :kk] = part
    thr = best_v.max(axis=1)
    for c0 in range(c1, n, chunk):
        c2 = min(c0 + chunk, n)
        v = rn[None, c0:c2] - 2.0 * (q @ X[c0:c2].T)
        hr, hc = np.nonzero(v < thr[:, None])
        if hr.size == 0:
            continue
        # the rows with a hit: their kept values and this chunk's hits side by side, keep the `keep` smallest
        ur, start = np.unique(hr, return_index=True)           # (nonzero returns row-major order: hits of a row are contiguous)
        cnt = np.diff(np.append(start, hr.size))
        width = int(cnt.max())
        slot = np.arange(hr.size) - np.repeat(start, cnt)
        hv = np.full((ur.size, width), np.inf)
        hi = np.zeros((ur.size, width), dtype=np.int64)
        rpos = np.repeat(np.arange(ur.size), cnt)
        hv[rpos, slot] = v[hr, hc]
        hi[rpos, slot] = hc + c0
        mv = np.concatenate([best_v[ur], hv], axis=1)
        mi = np.concatenate([best_i[ur], hi], axis=1)
        sel = np.argpartition(mv, keep - 1, axis=1)[:, :keep]
        rr = np.arange(ur.size)[:, None]
        best_v[ur] = mv[rr, sel]
        best_i[ur] = mi[rr, sel]
        thr[ur] = best_v[ur].max(axis=1)
    valid = np.isfinite(best_v)
    part = np.where(valid, best_i, 0)
    diff = X[part] - q[:, None, :]
    d2 = np.einsum("ijk,ijk->ij", diff, diff)
    d2 = np.where(valid, d2, np.inf)
    order = np.lexsort((part, d2), axis=1)[:, :k]
    return part[rows, order], np.sqrt(d2[rows, order])


def blas_knn(X, Q, k, block=256, workers=None, chunk=4096):
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
        # (all host threads: a worker holds a block x chunk tile of values, not block x n)
        workers = max(1, cores if chunk and X.shape[0] > 2 * chunk else min(cores, 64, int(32e9 // max(1, 2 * block * X.shape[0] * 8))))
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
        if chunk and X.shape[0] > 2 * chunk:
            i, dd = _blas_block_filtered(X, rn, Q[b0:b0 + block], k, keep, chunk)
        else:
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


def physical_cores():
    """Host cores without their SMT siblings (FMA-bound code gains nothing from the second thread of a core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/devices/system/cpu/smt/active") as f:
            if f.read().strip() == "1":
                n = max(1, n // 2)
    except OSError:
        pass
    return n


def cpu_quota():
    """CPUs' worth of time the container's cgroup grants this process (cpu.max; None: unlimited / unknown).  A box may show
    256 hardware threads and grant 16: more threads than that only add throttling."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                return float(q) / float(per)
        except (OSError, ValueError):
            pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def usable_cores():
    """Threads worth starting: the physical cores of the affinity mask, capped by the cgroup's CPU quota."""
    n = physical_cores()
    q = cpu_quota()
    return max(1, min(n, int(q + 0.5))) if q else n


_TILED = None


def tiled_knn(X, Q, k, nthreads=0):
    """Exact kNN, all cores, the way a CPU wants it (oracle/tiled_knn_baseline.c: packed operands, a 4 x 8 AVX2 + FMA
    micro-kernel, the reference block resident in L2, the threshold filter on the tile in registers, exact re-evaluation of
    the kept).  Returns (idx 1-based [nq x k], dist [nq x k])."""
    global _TILED
    if _TILED is None:
        path = os.path.join(_HERE, "libtiled_knn_baseline.so")
        if not os.path.exists(path):
            raise RuntimeError("baseline not built: run `make -C oracle`")
        _TILED = ctypes.CDLL(path)
    X = np.ascontiguousarray(X, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    nq, d = Q.shape
    idx = np.zeros((nq, k), dtype=np.int32)
    dist = np.zeros((nq, k), dtype=np.float64)
    f64p, i32p = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    rc = _TILED.tiled_knn(X.ctypes.data_as(f64p), X.shape[0], Q.ctypes.data_as(f64p), nq, d, int(k),
                          idx.ctypes.data_as(i32p), dist.ctypes.data_as(f64p), int(nthreads))
    if rc:
        raise RuntimeError("tiled_knn: bad arguments or out of memory")
    return idx + 1, dist
