"""CPU restatement (numpy) of the steps directly upstream of the merge engine in fastMNN():
cosineNorm (R/cosineNorm.R:53-82) and multiBatchPCA (R/multiBatchPCA.R:140-557), plus fastMNN's list front-end
(R/fastMNN.R:339-358).

TEST INFRASTRUCTURE ONLY -- never imported by batchelor_amd/.  Matrices here are genes x cells, as the reference has
them at this level.  The SVD is numpy's exact LAPACK SVD (BSPARAM=ExactParam() in the reference's own tests); singular
vectors are defined up to sign, which the tests account for exactly as tests/testthat/test-multi-pca.R:6-10 does.

Pinned by the reference's tests re-expressed in tests/test_oracle_pca.py: test-cos-norm.R:5-47, test-multi-pca.R:13-56
(duplicated-batch invariance, prcomp equivalence at equal sizes, distance preservation at full rank), :97-105
(projection identity), :237-265 (variance explained).
"""
from __future__ import annotations

import numpy as np

from . import fastmnn_oracle as engine


def cosine_norm(x, mode="matrix"):
    """R/cosineNorm.R:53-82: l2 = sqrt(colSums(x^2)); columns divided by pmax(1e-8, l2)."""
    x = np.asarray(x, dtype=np.float64)
    l2 = np.sqrt((x ** 2).sum(axis=0))
    if mode == "l2norm":
        return l2
    mat = x / np.maximum(1e-8, l2)[None, :]
    return mat if mode == "matrix" else {"matrix": mat, "l2norm": l2}


def _list_weights(tree, current=1.0):
    """R/multiBatchPCA.R:338-352 (.get_list_weights): equal weight at each split of a nested list."""
    out = []
    reweight = current / len(tree)
    for item in tree:
        if isinstance(item, (list, tuple)):
            out.extend(_list_weights(item, reweight))
        else:
            out.append((item, reweight))
    return out


def construct_weight_vector(ncells, weights):
    """R/multiBatchPCA.R:299-334."""
    n = np.asarray(ncells, dtype=np.float64)
    if weights is None or weights is True:
        return np.ones_like(n)
    if weights is False:
        return n.copy()
    if isinstance(weights, (list, tuple)) and any(isinstance(w, (list, tuple)) for w in weights):
        pairs = _list_weights(weights)
        ids = sorted(int(i) for i, _ in pairs)
        if ids != list(range(1, n.size + 1)):
            raise ValueError("invalid integer indices in tree-like 'weights'")
        out = np.zeros_like(n)
        for i, w in pairs:
            out[int(i) - 1] = w
        return out
    w = np.asarray(weights, dtype=np.float64)
    if w.size != n.size:
        raise ValueError("'length(weights)' should be the same as number of entries in '...'")
    return w


def process_listed_matrices_for_pca(mat_list, weights=None):
    """R/multiBatchPCA.R:265-322 (non-deferred arithmetic; deferred centring is algebraically the same)."""
    mats = [np.asarray(m, dtype=np.float64) for m in mat_list]
    w = construct_weight_vector([m.shape[1] for m in mats], weights)
    grand = 0
    for m, wi in zip(mats, w):
        grand = grand + m.mean(axis=1) * wi
    grand = grand / w.sum()
    centered = [m - grand[:, None] for m in mats]
    scaled = np.hstack([c / np.sqrt(c.shape[1] / wi) for c, wi in zip(centered, w)])
    return centered, scaled, grand


def multi_batch_pca(mat_list, d=50, weights=None, get_variance=False, method="svd"):
    """R/multiBatchPCA.R:211-258 (.multi_pca_list): SVD of the scaled matrix, projection of the UNSCALED centred
    batches on u.  Returns (list of cells x d matrices, metadata dict).
    method="gram": the same left singular vectors from the dense eigendecomposition of scaled^T scaled (cells x cells),
    u = scaled v / s -- for shapes with far more genes than cells, where LAPACK's SVD of the tall matrix takes minutes."""
    if len(mat_list) == 0:
        raise ValueError("at least one batch must be specified")
    g = np.asarray(mat_list[0]).shape[0]
    if any(np.asarray(m).shape[0] != g for m in mat_list):
        raise ValueError("number of rows is not the same across batches")
    centered, scaled, centers = process_listed_matrices_for_pca(mat_list, weights)
    if method == "gram":
        ev, v = np.linalg.eigh(scaled.T @ scaled)
        top = np.argsort(ev)[::-1][:d]
        s = np.sqrt(np.maximum(ev[np.argsort(ev)[::-1]], 0.0))
        u = (scaled @ v[:, top]) / s[:d][None, :]
    else:
        u, s, _ = np.linalg.svd(scaled, full_matrices=False)
        u = u[:, :d]
    out = [c.T @ u for c in centered]
    meta = {"rotation": u, "centers": centers}
    if get_variance:
        nb = len(mat_list)
        meta["var.explained"] = s[:d] ** 2 / nb
        meta["var.total"] = float((scaled ** 2).sum() / nb)
    return out, meta


def fast_mnn_single(x, batch, k=20, prop_k=None, restrict=None, cos_norm=True, ndist=3, d=50, weights=None,
                    merge_order=None, auto_merge=False, min_batch_skip=0.0, nthreads=0):
    """R/fastMNN.R:364-388 (.fast_mnn_single): one genes x cells object, `batch` names each cell's batch.  The PCA takes
    the levels of factor(batch) as the batches (.multi_pca_single, R/multiBatchPCA.R:241-258); the PCs are divided into
    batches (R/fastMNN.R:379), corrected, and rows / pairs are put back in the caller's order (:383-385)."""
    x = np.asarray(x, dtype=np.float64)
    batch = np.asarray(batch)
    if cos_norm:
        x = cosine_norm(x)
    levels = sorted(set(batch.tolist()))
    pcs, meta = multi_batch_pca([x[:, batch == lev] for lev in levels], d=d, weights=weights)
    allpcs = np.empty((x.shape[1], pcs[0].shape[1]))
    for lev, pc in zip(levels, pcs):
        allpcs[batch == lev] = pc
    out = engine.reduced_mnn(allpcs, batch=batch, k=k, prop_k=prop_k, restrict=None if restrict is None else [restrict],
                             ndist=ndist, merge_order=merge_order, auto_merge=auto_merge, min_batch_skip=min_batch_skip,
                             nthreads=nthreads)
    return out, meta


def fast_mnn(*batches, k=20, prop_k=None, restrict=None, cos_norm=True, ndist=3, d=50, weights=None,
             merge_order=None, auto_merge=False, min_batch_skip=0.0, nthreads=0, pca_method="svd"):
    """R/fastMNN.R:339-358 (.fast_mnn_list): cosine normalisation, multi-batch PCA, merge engine.
    Batches are genes x cells.  Returns (engine result, pca metadata)."""
    if len(batches) < 2:
        raise ValueError("at least two batches must be specified")
    mats = [np.asarray(b, dtype=np.float64) for b in batches]
    if cos_norm:
        mats = [cosine_norm(m) for m in mats]
    pcs, meta = multi_batch_pca(mats, d=d, weights=weights, method=pca_method)
    out = engine.fast_mnn(pcs, k=k, prop_k=prop_k, restrict=restrict, ndist=ndist, merge_order=merge_order,
                          auto_merge=auto_merge, min_batch_skip=min_batch_skip, nthreads=nthreads)
    return out, meta
