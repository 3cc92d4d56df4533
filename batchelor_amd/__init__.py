"""batchelor_amd: MI355X-native fastMNN / reducedMNN hot path behind batchelor's interface.

Host side mirrors the reference's R functions (same names, arguments, 1-based conventions and error text); the
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of include/batchelor_mi355x.h.
"""
from ._lib import BatchelorMI355XError, device_count  # noqa: F401
from .neighbors import query_knn  # noqa: F401
from .reduced_mnn import MnnEngine, MnnResult, divideIntoBatches, reducedMNN  # noqa: F401
from .natives import (adjust_shift_variance, find_mutual_nn, find_mutual_nns, smooth_gaussian_kernel)  # noqa: F401
from .multi_batch_pca import DevicePCA, cosineNorm, multiBatchPCA, multiBatchPCA_host, project  # noqa: F401
from .fast_mnn import fastMNN  # noqa: F401
