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
    """bmx_pca_t: the batches (genes x cells) stay in HBM; fit() = multiBatchPCA (R/multiBatchPCA.R:211-322) by
    Chebyshev-filtered subspace iteration on the FP64 matrix cores, run until the Ritz residual is below `tol`;
    project(b) = crossprod(cosineNorm(x_b) - centers, rotation)."""

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
        self.iters_used = 0
        self.residual = float("nan")

    def add_batch(self, x, weight=1.0, cos_norm=False):
        x = _lib.as_f(x)
        if x.ndim != 2 or x.shape[0] != self.G:
            raise ValueError("number of rows is not the same across batches")
        _lib.check(_lib.lib().bmx_pca_add_batch(self._h, _lib.f64p(x), ctypes.c_int64(x.shape[1]),
                                                ctypes.c_double(float(weight)), 1 if cos_norm else 0))
        self.ncells.append(int(x.shape[1]))

    def begin_batch(self, n, weight=1.0, cos_norm=False):
        """Announce a batch of n cells whose columns follow in blocks (add_block), so that it never has to exist on the
        host in one piece."""
        _lib.check(_lib.lib().bmx_pca_begin_batch(self._h, ctypes.c_int64(int(n)), ctypes.c_double(float(weight)),
                                                  1 if cos_norm else 0))
        self.ncells.append(int(n))

    def add_block(self, x_block):
        x_block = _lib.as_f(x_block)
        if x_block.ndim != 2 or x_block.shape[0] != self.G:
            raise ValueError("number of rows is not the same across batches")
        _lib.check(_lib.lib().bmx_pca_add_block(self._h, _lib.f64p(x_block), ctypes.c_int64(x_block.shape[1])))

    def fit(self, d=50, tol=1e-9, max_iters=500, iters=None):
        """tol: relative Ritz residual at which the iteration stops (raises if max_iters applications of the operator do
        not reach it).  iters=N: the fixed-count form of round 2 (N plain subspace iterations, no test)."""
        centers = np.zeros(self.G)
        rotation = np.zeros((self.G, d), order="F")
        sdev = np.zeros(d)
        if iters is not None:
            _lib.check(_lib.lib().bmx_pca_fit(self._h, int(d), int(iters), _lib.f64p(centers), _lib.f64p(rotation),
                                              _lib.f64p(sdev)))
            self.iters_used, self.residual = int(iters), float("nan")
        else:
            used, res = ctypes.c_int32(0), ctypes.c_double(0.0)
            rc = _lib.lib().bmx_pca_fit_tol(self._h, int(d), ctypes.c_double(float(tol)), int(max_iters),
                                            _lib.f64p(centers), _lib.f64p(rotation), _lib.f64p(sdev), ctypes.byref(used),
                                            ctypes.byref(res))
            self.iters_used, self.residual = used.value, res.value
            _lib.check(rc)
        self.d = int(d)
        return {"rotation": np.ascontiguousarray(rotation), "centers": centers, "d": sdev,
                "iters_used": self.iters_used, "residual": self.residual}

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


def multiBatchPCA(*batches, d=50, weights=None, cos_norm=False, tol=1e-9, max_iters=500, iters=None, device=0,
                  return_pcs=True, l2=None, block=65536):
    """multiBatchPCA(..., d=, weights=) (R/multiBatchPCA.R:140-258) on the device.  Batches are genes x cells; with
    cos_norm the cosine normalisation of fastMNN (R/fastMNN.R:348-351) is applied on the fly.
    Returns {"rotation", "centers", "d", "weights", "iters_used", "residual", "path"} plus "pcs": the list of cells x d
    projections.

    The device path iterates until the relative Ritz residual is <= tol (the reference's irlba stops at 1e-5 on the
    singular triplets, R/multiBatchPCA.R:386-393) and raises when max_iters passes do not get there.  Inputs the blocked
    iteration cannot take -- fewer genes or cells than its block of 64 / 128 vectors, d > 120, data of rank below the
    block -- go to multiBatchPCA_host (north_star keeps multiBatchPCA on the host path anyway), as does a call in
    round 1's form with per-cell norms `l2=` (and its `block=`)."""
    if len(batches) == 1 and isinstance(batches[0], (list, tuple)):
        batches = tuple(batches[0])
    if len(batches) == 0:
        raise ValueError("at least one batch must be specified")
    G = np.asarray(batches[0]).shape[0]
    if any(np.asarray(m).ndim != 2 or np.asarray(m).shape[0] != G for m in batches):
        raise ValueError("number of rows is not the same across batches")
    ncells = [np.asarray(m).shape[1] for m in batches]
    w = _weight_vector(ncells, weights)
    width = 64 if d <= 56 else 128

    def host(reason):
        if l2 is not None:
            norms = l2
        else:
            norms = [cosineNorm(m, mode="l2norm") for m in batches] if cos_norm else None
        out = multiBatchPCA_host(*batches, d=d, weights=weights, l2=norms, block=block)
        out.update(iters_used=0, residual=0.0, path="host: " + reason)
        if return_pcs:
            out["pcs"] = [project(m, out["rotation"], out["centers"], cos_norm=norms is not None) for m in batches]
        return out

    if l2 is not None:
        return host("per-cell norms given (round-1 signature)")
    if d > 120 or G < width or sum(ncells) <= width:
        return host("fewer genes / cells than the device block, or d > 120")
    pca = DevicePCA(G, device)
    try:
        for m, wi in zip(batches, w):
            pca.add_batch(m, weight=wi, cos_norm=cos_norm)
        try:
            out = pca.fit(d=d, tol=tol, max_iters=max_iters, iters=iters)
        except _lib.BatchelorMI355XError as exc:
            if "rank below the subspace width" in str(exc):
                pca.close()
                return host("data of rank below the device block")
            raise
        out["weights"] = w
        out["path"] = "device"
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
    if G > 4096:
        return _host_pca_lanczos(mats, inv, grand, w, d, block)
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


def _host_pca_lanczos(mats, inv, grand, w, d, block):
    """Many genes: the genes x genes Gram matrix is not formed (20 000 genes: 3.2 GB and a dense eigensolver of 1e13
    flops); its top d eigenpairs come from implicitly restarted Lanczos on the operator v -> sum_b (w_b/n_b) C_b (C_b^T v),
    run to machine precision -- the host analogue of the reference's default BSPARAM=IrlbaParam() (R/fastMNN.R:287).
    C_b = X_b diag(inv_b) - grand 1^T is never formed: the per-cell factors and the centring are applied to the vectors."""
    from scipy.sparse.linalg import LinearOperator, eigsh
    G = mats[0].shape[0]

    def matvec(v):
        v = np.asarray(v, dtype=np.float64).reshape(G)
        out = np.zeros(G)
        gv = float(grand @ v)
        for i, m in enumerate(mats):
            acc = np.zeros(G)
            zsum = 0.0
            for lo in range(0, m.shape[1], block):
                hi = min(m.shape[1], lo + block)
                blk = m[:, lo:hi]
                z = blk.T @ v                 # C_b^T v for these cells ...
                if inv[i] is not None:
                    z *= inv[i][lo:hi]
                z -= gv
                zsum += float(z.sum())
                acc += blk @ (z if inv[i] is None else z * inv[i][lo:hi])
            out += (w[i] / m.shape[1]) * (acc - grand * zsum)
        return out

    op = LinearOperator((G, G), matvec=matvec, dtype=np.float64)
    k = min(d, G - 1)
    evals, evecs = eigsh(op, k=k, which="LA", tol=0, ncv=min(G, max(3 * k, k + 32)),
                         v0=np.random.default_rng(0).standard_normal(G))
    order = np.argsort(evals)[::-1]
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
