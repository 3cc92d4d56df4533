"""batchelor_amd: MI355X-native fastMNN / reducedMNN hot path behind batchelor's interface.

Host side mirrors the reference's R functions (same names, arguments, 1-based conventions and error text); the
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of include/batchelor_mi355x.h.
"""
from ._lib import BatchelorMI355XError, device_count  # noqa: F401
from .neighbors import query_knn  # noqa: F401
