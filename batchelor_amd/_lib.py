"""ctypes loader for libbatchelor_mi355x.so (the C ABI of include/batchelor_mi355x.h).

There is no CPU fallback: if the HIP library is missing this module raises, and every op raises if no GPU answers.
torch is imported first on purpose: it brings its own libamdhip64, and loading ours afterwards makes both share that
one HIP runtime (a process with two HIP runtimes cannot share device pointers or streams).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BMX_LIB: developer switch, another build of the same library (e.g. `make STAMPS=1`: per-wave cycle accounting)
LIB_PATH = os.environ.get("BMX_LIB") or os.path.join(_HERE, "libbatchelor_mi355x.so")

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f64p = ctypes.POINTER(ctypes.c_double)

_lib = None


class BatchelorMI355XError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C batchelor_amd/csrc`). There is no CPU fallback for the MI355X hot path.")
        try:
            import torch  # noqa: F401  (one HIP runtime per process: see module docstring)
        except Exception:  # pragma: no cover - torch is optional for single-GPU use
            pass
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _lib.bmx_last_error.restype = ctypes.c_char_p
        _lib.bmx_device_count.restype = ctypes.c_int32
        _lib.bmx_last_knn_exact_fallbacks.restype = ctypes.c_int64
        _lib.bmx_free.argtypes = [ctypes.c_void_p]
        _lib.bmx_free.restype = None
    return _lib


def check(rc):
    if rc != 0:
        raise BatchelorMI355XError(rc, lib().bmx_last_error().decode("utf-8", "replace"))


def dev_set(name, value):
    """Testing hook (bmx_dev_set): process-wide knobs of the library that tests and developer scripts set explicitly --
    the environment of the host process never changes what the library computes.  dev_set("reset", 0) restores all."""
    check(lib().bmx_dev_set(name.encode(), ctypes.c_int32(int(value))))


def dev_get(name):
    """Counters for tests and bench.py (bmx_dev_get): e.g. "asv_literal_cells", "asv_fallback_cells", "asv_tiled_cells",
    "asv_tally_reset"."""
    v = ctypes.c_int64(0)
    check(lib().bmx_dev_get(name.encode(), ctypes.byref(v)))
    return int(v.value)


def dev_get_bytes(name, n):
    """Testing hook (bmx_dev_get_bytes): e.g. "asv_modes" after dev_set("asv_modes", n)."""
    out = np.zeros(int(n), dtype=np.uint8)
    check(lib().bmx_dev_get_bytes(name.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(int(n))))
    return out


def device_count():
    return int(lib().bmx_device_count())


def require_gpu():
    if device_count() < 1:
        raise BatchelorMI355XError(-1, "no MI355X / HIP device visible: the batchelor_amd hot path has no CPU fallback")


def f64p(a):
    return a.ctypes.data_as(c_f64p)


def i32p(a):
    return a.ctypes.data_as(c_i32p)


def as_f(a):
    """Column-major float64 copy/view: R's matrix layout at the C ABI."""
    return np.asfortranarray(a, dtype=np.float64)


def take_i32(ptr, n):
    """Copy an engine-allocated int32 array of length n and free it."""
    if n == 0:
        out = np.zeros(0, dtype=np.int32)
    else:
        out = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
    lib().bmx_free(ptr)
    return out
