"""BiocNeighbors-shaped entry points used on the fastMNN path (queryKNN, findMutualNN), served by the HIP kernels."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def query_knn(X, query, k, get_index=True, get_distance=True):
    """queryKNN(X, query, k) (call sites R/MNN_tree.R:129 via findMutualNN, R/fastMNN.R:605).

    Returns (index [nq x k] 1-based into rows of X, distance [nq x k] Euclidean), ascending distance, exact.
    """
    _lib.require_gpu()
    X = _lib.as_f(X)
    query = _lib.as_f(query)
    if X.ndim != 2 or query.ndim != 2 or X.shape[1] != query.shape[1]:
        raise ValueError("number of dimensions do not match between 'X' and 'query'")
    nx, d = X.shape
    nq = query.shape[0]
    k = int(k)
    index = np.zeros((nq, k), dtype=np.int32, order="F")
    distance = np.zeros((nq, k), dtype=np.float64, order="F")
    _lib.check(_lib.lib().bmx_query_knn(_lib.f64p(X), nx, _lib.f64p(query), nq, d, k,
                                        _lib.i32p(index) if get_index else None,
                                        _lib.f64p(distance) if get_distance else None))
    return index, distance


def last_knn_exact_fallbacks():
    return int(_lib.lib().bmx_last_knn_exact_fallbacks())


def set_force_exact_knn(on):
    _lib.lib().bmx_set_force_exact_knn(ctypes.c_int32(1 if on else 0))
