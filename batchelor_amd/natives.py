"""Python-side stubs of batchelor's three registered .Call kernels (R/RcppExports.R:4-14), bound to the HIP
implementations through the C ABI, plus the single merge-step primitives used by the parity tests."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def find_mutual_nns(left, right):
    """find_mutual_nns(left, right) (src/find_mutual_nns.cpp:8-41): 1-based kNN index matrices -> (first, second)."""
    _lib.require_gpu()
    left = np.asfortranarray(left, dtype=np.int32)
    right = np.asfortranarray(right, dtype=np.int32)
    pl, pr, n = _lib.c_i32p(), _lib.c_i32p(), ctypes.c_int64(0)
    _lib.check(_lib.lib().bmx_find_mutual_nns(_lib.i32p(left), left.shape[0], left.shape[1], _lib.i32p(right),
                                              right.shape[0], right.shape[1], ctypes.byref(pl), ctypes.byref(pr),
                                              ctypes.byref(n)))
    return _lib.take_i32(pl, n.value), _lib.take_i32(pr, n.value)


def smooth_gaussian_kernel(averaged, index, mat, sigma2):
    """smooth_gaussian_kernel(averaged, index, mat, sigma2) (src/smooth_gaussian_kernel.cpp:11-118)."""
    _lib.require_gpu()
    averaged = _lib.as_f(averaged)
    mat = _lib.as_f(mat)
    index = np.ascontiguousarray(index, dtype=np.int32)
    g, U = averaged.shape
    gd, n = mat.shape
    out = np.zeros((g, n), dtype=np.float64, order="F")
    _lib.check(_lib.lib().bmx_smooth_gaussian_kernel(_lib.f64p(averaged), g, U, _lib.i32p(index), index.size,
                                                     _lib.f64p(mat), gd, n, ctypes.c_double(sigma2), _lib.f64p(out)))
    return out


def adjust_shift_variance(data1, data2, vect, sigma2, restrict1, restrict2):
    """adjust_shift_variance(data1, data2, vect, sigma2, restrict1, restrict2) (src/adjust_shift_variance.cpp:30-164)."""
    _lib.require_gpu()
    data1, data2, vect = _lib.as_f(data1), _lib.as_f(data2), _lib.as_f(vect)
    r1 = np.ascontiguousarray(restrict1, dtype=np.int32)
    r2 = np.ascontiguousarray(restrict2, dtype=np.int32)
    out = np.zeros(data2.shape[1], dtype=np.float64)
    _lib.check(_lib.lib().bmx_adjust_shift_variance(_lib.f64p(data1), data1.shape[0], data1.shape[1], _lib.f64p(data2),
                                                    data2.shape[0], data2.shape[1], _lib.f64p(vect), vect.shape[0],
                                                    vect.shape[1], ctypes.c_double(sigma2), _lib.i32p(r1), r1.size,
                                                    _lib.i32p(r2), r2.size, _lib.f64p(out)))
    return out


def adjust_shift_variance_form(n2, nr1, nr2):
    """Which form a call of these sizes takes: "exact" (bit-equal to the CPU restatement) or "tiled" (FP64-MFMA tiles +
    histogram quantile; may pick a neighbouring quantile in ill-conditioned cells) -- bmx_adjust_shift_variance_form."""
    return {1: "exact", 2: "tiled", 3: "bisect"}[int(_lib.lib().bmx_adjust_shift_variance_form(int(n2), int(nr1), int(nr2)))]


def find_mutual_nn(data1, data2, k1, k2):
    """findMutualNN(data1, data2, k1, k2) -> (first, second), 1-based."""
    _lib.require_gpu()
    data1, data2 = _lib.as_f(data1), _lib.as_f(data2)
    pf, ps, n = _lib.c_i32p(), _lib.c_i32p(), ctypes.c_int64(0)
    _lib.check(_lib.lib().bmx_find_mutual_nn(_lib.f64p(data1), data1.shape[0], _lib.f64p(data2), data2.shape[0],
                                             data1.shape[1], int(k1), int(k2), ctypes.byref(pf), ctypes.byref(ps),
                                             ctypes.byref(n)))
    return _lib.take_i32(pf, n.value), _lib.take_i32(ps, n.value)


def center_along_batch_vector(mat, batch_vec, restrict=None):
    """.center_along_batch_vector (R/fastMNN.R:626-640)."""
    _lib.require_gpu()
    out = np.array(mat, dtype=np.float64, order="F", copy=True)
    v = np.ascontiguousarray(batch_vec, dtype=np.float64)
    r = None if restrict is None else np.ascontiguousarray(restrict, dtype=np.int32)
    _lib.check(_lib.lib().bmx_center_along_batch_vector(_lib.f64p(out), out.shape[0], out.shape[1], _lib.f64p(v),
                                                        None if r is None else _lib.i32p(r),
                                                        -1 if r is None else r.size))
    return np.ascontiguousarray(out)


def tricube_weighted_correction(curdata, correction, in_mnn, k=20, ndist=3):
    """.tricube_weighted_correction (R/fastMNN.R:599-608)."""
    _lib.require_gpu()
    out = np.array(curdata, dtype=np.float64, order="F", copy=True)
    corr = _lib.as_f(correction)
    im = np.ascontiguousarray(in_mnn, dtype=np.int32)
    _lib.check(_lib.lib().bmx_tricube_weighted_correction(_lib.f64p(out), out.shape[0], out.shape[1], _lib.f64p(corr),
                                                          _lib.i32p(im), im.size, int(k), ctypes.c_double(ndist)))
    return np.ascontiguousarray(out)


def mnn_average_correction(refdata, curdata, k):
    """findMutualNN + .average_correction (R/fastMNN.R:567-580): (first, second, averaged [U x d], second_u)."""
    _lib.require_gpu()
    refdata, curdata = _lib.as_f(refdata), _lib.as_f(curdata)
    pf, ps, n = _lib.c_i32p(), _lib.c_i32p(), ctypes.c_int64(0)
    pa, pu, U = _lib.c_f64p(), _lib.c_i32p(), ctypes.c_int32(0)
    d = refdata.shape[1]
    _lib.check(_lib.lib().bmx_mnn_average_correction(_lib.f64p(refdata), refdata.shape[0], _lib.f64p(curdata),
                                                     curdata.shape[0], d, int(k), int(k), ctypes.byref(pf),
                                                     ctypes.byref(ps), ctypes.byref(n), ctypes.byref(pa),
                                                     ctypes.byref(pu), ctypes.byref(U)))
    first, second = _lib.take_i32(pf, n.value), _lib.take_i32(ps, n.value)
    if U.value:
        avg = np.ctypeslib.as_array(pa, shape=(U.value * d,)).copy().reshape((U.value, d), order="F")
    else:
        avg = np.zeros((0, d))
    _lib.lib().bmx_free(pa)
    return first, second, np.ascontiguousarray(avg), _lib.take_i32(pu, U.value)


def total_variance(data):
    """One term of .compute_perbatch_var (R/fastMNN.R:651-658)."""
    _lib.require_gpu()
    data = _lib.as_f(data)
    out = ctypes.c_double(0)
    _lib.check(_lib.lib().bmx_total_variance(_lib.f64p(data), data.shape[0], data.shape[1], ctypes.byref(out)))
    return out.value
