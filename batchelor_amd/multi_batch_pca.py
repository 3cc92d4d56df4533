"""multiBatchPCA on the device (default) or on the host (BASELINE.json north_star: "multiBatchPCA stays on the
reference CPU path"; kept as `multiBatchPCA_host`), and the device cosineNorm / projection steps on either side of it.

multiBatchPCA follows R/multiBatchPCA.R:211-322: grand mean of the batch means (weighted), each centred batch scaled
by 1/sqrt(n_b / w_b), left singular vectors u of the scaled genes x cells matrix, projection of the UNSCALED centred
batches on u.  Instead of an SVD of the genes x N matrix this host implementation accumulates the genes x genes Gram
matrix batch by batch (N only enters through blocked GEMMs, so 10^5..10^6 cells never have to be held scaled) and
takes its top eigenvectors -- the same subspace and, up to sign, the same vectors.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def cosineNorm(x, mode="matrix"):
    """cosineNorm(x, mode=c("matrix", "all", "l2norm")) (R/cosineNorm.R:53-82) on the GPU; x is genes x cells."""
    _lib.require_gpu()
    if mode not in ("matrix", "all", "l2norm"):
        raise ValueError("'arg' should be one of 'matrix', 'all', 'l2norm'")
    x = _lib.as_f(x)
    G, n = x.shape
    l2 = np.zeros(n, dtype=np.float64)
    mat = None if mode == "l2norm" else np.zeros((G, n), dtype=np.float64, order="F")
    _lib.check(_lib.lib().bmx_cosine_norm(_lib.f64p(x), G, n, _lib.f64p(l2), None if mat is None else _lib.f64p(mat)))
    if mode == "l2norm":
        return l2
    return mat if mode == "matrix" else {"matrix": mat, "l2norm": l2}


def _list_weights(tree, current=1.0):
    out = []
    share = current / len(tree)
    for item in tree:
        if isinstance(item, (list, tuple)):
            out.extend(_list_weights(item, share))
        else:
            out.append((item, share))
    return out


def _weight_vector(ncells, weights):
    """.construct_weight_vector (R/multiBatchPCA.R:299-334)."""
    n = np.asarray(ncells, dtype=np.float64)
    if weights is None or weights is True:
        return np.ones_like(n)
    if weights is False:
        return n.copy()
    if isinstance(weights, (list, tuple)) and any(isinstance(w, (list, tuple)) for w in weights):
        pairs = _list_weights(weights)
        if sorted(int(i) for i, _ in pairs) != list(range(1, n.size + 1)):
            raise ValueError("invalid integer indices in tree-like 'weights'")
        out = np.zeros_like(n)
        for i, w in pairs:
            out[int(i) - 1] = w
        return out
    w = np.asarray(weights, dtype=np.float64)
    if w.size != n.size:
        raise ValueError("'length(weights)' should be the same as number of entries in '...'")
    return w


class DevicePCA:
    """bmx_pca_t: the batches (genes x cells) stay in HBM; fit() = multiBatchPCA (R/multiBatchPCA.R:211-322) by blocked
    subspace iteration on the FP64 matrix cores; project(b) = crossprod(cosineNorm(x_b) - centers, rotation)."""

    def __init__(self, n_genes, device=0):
        _lib.require_gpu()
        self._h = ctypes.c_void_p()
        lib = _lib.lib()
        lib.bmx_pca_destroy.argtypes = [ctypes.c_void_p]
        lib.bmx_pca_destroy.restype = None
        _lib.check(lib.bmx_pca_create(int(device), int(n_genes), ctypes.byref(self._h)))
        self.G = int(n_genes)
        self.ncells = []
        self.d = 0

    def add_batch(self, x, weight=1.0, cos_norm=False):
        x = _lib.as_f(x)
        if x.ndim != 2 or x.shape[0] != self.G:
            raise ValueError("number of rows is not the same across batches")
        _lib.check(_lib.lib().bmx_pca_add_batch(self._h, _lib.f64p(x), ctypes.c_int64(x.shape[1]),
                                                ctypes.c_double(float(weight)), 1 if cos_norm else 0))
        self.ncells.append(int(x.shape[1]))

    def fit(self, d=50, iters=15):
        centers = np.zeros(self.G)
        rotation = np.zeros((self.G, d), order="F")
        sdev = np.zeros(d)
        _lib.check(_lib.lib().bmx_pca_fit(self._h, int(d), int(iters), _lib.f64p(centers), _lib.f64p(rotation),
                                          _lib.f64p(sdev)))
        self.d = int(d)
        return {"rotation": np.ascontiguousarray(rotation), "centers": centers, "d": sdev}

    def project(self, b):
        out = np.zeros((self.ncells[b], self.d), order="F")
        _lib.check(_lib.lib().bmx_pca_project(self._h, int(b), _lib.f64p(out)))
        return np.ascontiguousarray(out)

    def close(self):
        if self._h:
            _lib.lib().bmx_pca_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multiBatchPCA(*batches, d=50, weights=None, cos_norm=False, iters=15, device=0, return_pcs=True):
    """multiBatchPCA(..., d=, weights=) on the device (R/multiBatchPCA.R:140-258).  Batches are genes x cells; with
    cos_norm the cosine normalisation of fastMNN (R/fastMNN.R:348-351) is applied on the fly.
    Returns {"rotation", "centers", "d", "weights"} plus "pcs": the list of cells x d projections."""
    if len(batches) == 1 and isinstance(batches[0], (list, tuple)):
        batches = tuple(batches[0])
    if len(batches) == 0:
        raise ValueError("at least one batch must be specified")
    G = np.asarray(batches[0]).shape[0]
    if any(np.asarray(m).ndim != 2 or np.asarray(m).shape[0] != G for m in batches):
        raise ValueError("number of rows is not the same across batches")
    w = _weight_vector([np.asarray(m).shape[1] for m in batches], weights)
    pca = DevicePCA(G, device)
    try:
        for m, wi in zip(batches, w):
            pca.add_batch(m, weight=wi, cos_norm=cos_norm)
        out = pca.fit(d=d, iters=iters)
        out["weights"] = w
        if return_pcs:
            out["pcs"] = [pca.project(b) for b in range(len(batches))]
    finally:
        pca.close()
    return out


def multiBatchPCA_host(*batches, d=50, weights=None, l2=None, block=65536):
    """Host PCA across batches (genes x cells each).  `l2` (optional list of per-cell norms) applies the cosine
    normalisation on the fly, so the normalised matrices are never materialised.

    Returns {"rotation": [G x d], "centers": [G], "d": singular values, "weights": w}.  Project with `project()`.
    """
    if len(batches) == 1 and isinstance(batches[0], (list, tuple)):
        batches = tuple(batches[0])
    if len(batches) == 0:
        raise ValueError("at least one batch must be specified")
    mats = [np.asarray(b, dtype=np.float64) for b in batches]
    G = mats[0].shape[0]
    if any(m.ndim != 2 or m.shape[0] != G for m in mats):
        raise ValueError("number of rows is not the same across batches")
    w = _weight_vector([m.shape[1] for m in mats], weights)
    inv = [None if l2 is None else 1.0 / np.maximum(1e-8, np.asarray(l2[i], dtype=np.float64)) for i in range(len(mats))]

    def cols(i, lo, hi):
        blk = mats[i][:, lo:hi]
        return blk if inv[i] is None else blk * inv[i][None, lo:hi]

    # pass 1: grand centre = weighted mean of the batch means (R/multiBatchPCA.R:268-281)
    grand = np.zeros(G)
    for i, m in enumerate(mats):
        s = np.zeros(G)
        for lo in range(0, m.shape[1], block):
            s += cols(i, lo, min(m.shape[1], lo + block)).sum(axis=1)
        grand += (s / m.shape[1]) * w[i]
    grand /= w.sum()
    # pass 2: Gram matrix of the scaled, centred data: sum_b (w_b / n_b) C_b C_b^T   (scaled = C_b / sqrt(n_b / w_b))
    gram = np.zeros((G, G))
    for i, m in enumerate(mats):
        for lo in range(0, m.shape[1], block):
            c = cols(i, lo, min(m.shape[1], lo + block)) - grand[:, None]
            gram += (w[i] / m.shape[1]) * (c @ c.T)
    evals, evecs = np.linalg.eigh(gram)
    order = np.argsort(evals)[::-1][:d]
    return {"rotation": np.ascontiguousarray(evecs[:, order]), "centers": grand,
            "d": np.sqrt(np.maximum(evals[order], 0.0)), "weights": w}


def project(x, rotation, centers, cos_norm=True):
    """crossprod(cosineNorm(x) - centers, rotation) in one GPU pass over x (R/fastMNN.R:348-354 +
    R/multiBatchPCA.R:236-239).  x: genes x cells; returns cells x d."""
    _lib.require_gpu()
    x = _lib.as_f(x)
    rot = _lib.as_f(rotation)
    cen = np.ascontiguousarray(centers, dtype=np.float64)
    G, n = x.shape
    if rot.shape[0] != G or cen.size != G:
        raise ValueError("number of rows is not the same across batches")
    d = rot.shape[1]
    out = np.zeros((n, d), dtype=np.float64, order="F")
    _lib.check(_lib.lib().bmx_cosnorm_project(_lib.f64p(x), G, n, _lib.f64p(rot), d, _lib.f64p(cen),
                                              ctypes.c_int32(1 if cos_norm else 0), _lib.f64p(out)))
    return np.ascontiguousarray(out)
