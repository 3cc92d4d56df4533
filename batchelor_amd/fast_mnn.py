"""fastMNN() front-end for a list of batches (R/fastMNN.R:283-358, `.fast_mnn_list`): cosine normalisation,
multiBatchPCA and the projection on the GPU (`pca="host"` keeps multiBatchPCA on the host, as BASELINE.json's north_star
allows), then the MI355X merge engine.  Batches are genes x cells, as in the reference."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .multi_batch_pca import cosineNorm, multiBatchPCA, multiBatchPCA_host, project
from .reduced_mnn import MnnResult, _fast_mnn


@dataclass
class FastMnnResult:
    """What convertPCsToSCE (R/convertPCsToSCE.R:50-72) is built from: corrected PCs, batch, rotation, merge.info."""
    corrected: np.ndarray
    batch: np.ndarray
    rotation: np.ndarray
    centers: np.ndarray
    merge_info: object
    stats: object = None


def fastMNN(*batches, k=20, prop_k=None, restrict=None, cos_norm=True, ndist=3, d=50, weights=None,
            merge_order=None, auto_merge=False, min_batch_skip=0.0, names=None, device=0, pca="device",
            pca_iters=15) -> FastMnnResult:
    """fastMNN(..., k=, prop.k=, restrict=, cos.norm=, ndist=, d=, weights=, merge.order=, auto.merge=,
    min.batch.skip=) for >= 2 batches (R/fastMNN.R:339-358)."""
    if len(batches) == 1 and isinstance(batches[0], (list, tuple)):
        batches = tuple(batches[0])
    if len(batches) < 2:
        raise ValueError("at least two batches must be specified")  # R/fastMNN.R:345
    mats = [np.asarray(b, dtype=np.float64) for b in batches]
    G = mats[0].shape[0]
    if any(m.ndim != 2 or m.shape[0] != G for m in mats):
        raise ValueError("number of rows is not the same across batches")  # R/checkInputs.R:64-71
    if pca == "device":
        pca = multiBatchPCA(*mats, d=d, weights=weights, cos_norm=cos_norm, iters=pca_iters, device=device)
        pcs = pca["pcs"]                                                          # R/fastMNN.R:348-354 on the device
    else:
        l2 = [cosineNorm(m, mode="l2norm") for m in mats] if cos_norm else None  # R/fastMNN.R:348-351
        pca = multiBatchPCA_host(*mats, d=d, weights=weights, l2=l2)             # R/fastMNN.R:353-354 (host)
        pcs = [project(m, pca["rotation"], pca["centers"], cos_norm=cos_norm) for m in mats]
    out: MnnResult = _fast_mnn(pcs, k, prop_k, restrict, ndist, merge_order, auto_merge, min_batch_skip, names, device)
    return FastMnnResult(corrected=out.corrected, batch=out.batch, rotation=pca["rotation"], centers=pca["centers"],
                         merge_info=out.merge_info, stats=out.stats)
