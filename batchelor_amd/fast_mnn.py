"""fastMNN() front-end for a list of batches (R/fastMNN.R:283-358, `.fast_mnn_list`): cosine normalisation,
multiBatchPCA and the projection on the GPU (`pca="host"` keeps multiBatchPCA on the host, as BASELINE.json's north_star
allows), then the MI355X merge engine.  Batches are genes x cells, as in the reference."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .multi_batch_pca import cosineNorm, multiBatchPCA, multiBatchPCA_host, project
from .reduced_mnn import MnnResult, _fast_mnn, _reindex_pairings, divideIntoBatches


@dataclass
class FastMnnResult:
    """What convertPCsToSCE (R/convertPCsToSCE.R:50-72) is built from: corrected PCs, batch, rotation, merge.info."""
    corrected: np.ndarray
    batch: np.ndarray
    rotation: np.ndarray
    centers: np.ndarray
    merge_info: object
    stats: object = None


def _pca_step(mats, d, weights, cos_norm, device, pca, pca_tol, pca_maxit):
    """cosineNorm + multiBatchPCA + projection (R/fastMNN.R:348-354): (pca record, list of cells x d matrices)."""
    if pca not in ("device", "host"):
        raise ValueError("'pca' should be one of 'device', 'host'")
    if pca == "device":
        rec = multiBatchPCA(*mats, d=d, weights=weights, cos_norm=cos_norm, tol=pca_tol, max_iters=pca_maxit, device=device)
        return rec, rec["pcs"]
    l2 = [cosineNorm(m, mode="l2norm") for m in mats] if cos_norm else None  # R/fastMNN.R:348-351
    rec = multiBatchPCA_host(*mats, d=d, weights=weights, l2=l2)             # R/fastMNN.R:353-354 (host)
    return rec, [project(m, rec["rotation"], rec["centers"], cos_norm=cos_norm) for m in mats]


def fastMNN(*batches, batch=None, k=20, prop_k=None, restrict=None, cos_norm=True, ndist=3, d=50, weights=None,
            merge_order=None, auto_merge=False, min_batch_skip=0.0, names=None, device=0, pca="device",
            pca_tol=1e-9, pca_maxit=500) -> FastMnnResult:
    """fastMNN(..., batch=, k=, prop.k=, restrict=, cos.norm=, ndist=, d=, weights=, merge.order=, auto.merge=,
    min.batch.skip=) (R/fastMNN.R:283-331): several batches (`.fast_mnn_list`, :339-358) or ONE genes x cells object with
    `batch=` naming each cell's batch (`.fast_mnn_single`, :364-388)."""
    if len(batches) == 1 and isinstance(batches[0], (list, tuple)):
        batches = tuple(batches[0])
    if len(batches) == 1:
        return _fast_mnn_single(np.asarray(batches[0], dtype=np.float64), batch, k, prop_k, restrict, cos_norm, ndist, d,
                                weights, merge_order, auto_merge, min_batch_skip, device, pca, pca_tol, pca_maxit)
    if len(batches) < 2:
        raise ValueError("at least two batches must be specified")  # R/fastMNN.R:345
    mats = [np.asarray(b, dtype=np.float64) for b in batches]
    G = mats[0].shape[0]
    if any(m.ndim != 2 or m.shape[0] != G for m in mats):
        raise ValueError("number of rows is not the same across batches")  # R/checkInputs.R:64-71
    rec, pcs = _pca_step(mats, d, weights, cos_norm, device, pca, pca_tol, pca_maxit)
    out: MnnResult = _fast_mnn(pcs, k, prop_k, restrict, ndist, merge_order, auto_merge, min_batch_skip, names, device)
    return FastMnnResult(corrected=out.corrected, batch=out.batch, rotation=rec["rotation"], centers=rec["centers"],
                         merge_info=out.merge_info, stats=out.stats)


def _fast_mnn_single(x, batch, k, prop_k, restrict, cos_norm, ndist, d, weights, merge_order, auto_merge,
                     min_batch_skip, device, pca, pca_tol, pca_maxit):
    """.fast_mnn_single (R/fastMNN.R:364-388): the batches are the levels of factor(batch) in sorted order; the PCA sees
    them as separate batches (`.multi_pca_single`, R/multiBatchPCA.R:241-258), the merge engine too
    (divideIntoBatches), and rows and pairs come back in the caller's cell order."""
    if batch is None:
        raise ValueError("'batch' must be specified if '...' has only one object")  # R/checkInputs.R:128
    batch = np.asarray(batch)
    if x.ndim != 2 or batch.shape[0] != x.shape[1]:
        raise ValueError("'length(batch)' and 'ncol(x)' are not the same")  # R/checkInputs.R:131
    levels = sorted(set(batch.tolist()))
    if len(levels) < 2:
        raise ValueError("at least two batches must be specified")
    mats = [x[:, batch == lev] for lev in levels]
    rec, pcs = _pca_step(mats, d, weights, cos_norm, device, pca, pca_tol, pca_maxit)
    # divideIntoBatches(mat, batch, restrict, byrow=TRUE) on the PCs (R/fastMNN.R:379): restrict is ONE subsetting vector
    # over the cells of x (1-based positions or a logical mask)
    r = restrict[0] if isinstance(restrict, (list, tuple)) and len(restrict) == 1 else restrict
    allpcs = np.empty((x.shape[1], pcs[0].shape[1]))
    for lev, pc in zip(levels, pcs):
        allpcs[batch == lev] = pc
    div = divideIntoBatches(allpcs, batch, r)
    out = _fast_mnn(div["batches"], k, prop_k, div["restricted"], ndist, merge_order, auto_merge, min_batch_skip,
                    [str(lev) for lev in div["levels"]], device)
    reo = div["reorder"]                                         # R/fastMNN.R:383-385
    out.merge_info.pairs = _reindex_pairings(out.merge_info.pairs, reo)
    return FastMnnResult(corrected=out.corrected[reo - 1], batch=out.batch[reo - 1], rotation=rec["rotation"],
                         centers=rec["centers"], merge_info=out.merge_info, stats=out.stats)
