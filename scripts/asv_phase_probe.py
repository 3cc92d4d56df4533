"""Developer helper (GPU box): where the tiled adjust_shift_variance spends its time -- the stream, the round barrier, the
per-cell phase (100 MHz ticks added up over the workgroups: bmx_dev_get "asv_ticks_*") -- on one mid-size call.
   python scripts/asv_phase_probe.py [n2 nr1 g sigma [knob=value ...]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batchelor_amd import _lib, natives as nat  # noqa: E402

a = sys.argv[1:]
n2, nr1, g, sigma = (int(a[0]), int(a[1]), int(a[2]), float(a[3])) if len(a) >= 4 else (100000, 400000, 100, 1.0)
rng = np.random.default_rng(7)
s = 1.0 / np.sqrt(1.0 + np.arange(g) / 5.0)
d1 = np.asfortranarray((rng.standard_normal((nr1, g)) * s).T)
d2 = np.asfortranarray((rng.standard_normal((n2, g)) * s + 0.3).T)
cv = rng.standard_normal((n2, g)) * 0.2
_lib.dev_set("asv_fast", 1)
for kv in a[4:]:
    k, v = kv.split("=")
    _lib.dev_set(k, int(v))
r1, r2 = np.arange(nr1), np.arange(n2)
for rep in range(2):
    _lib.dev_get("asv_tally_reset")
    t = time.perf_counter()
    out = nat.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
    dt = time.perf_counter() - t
    lib = _lib.lib()
    import ctypes
    lib.bmx_last_native_kernel_ms.restype = ctypes.c_double
    kms = lib.bmx_last_native_kernel_ms()
    tk = [_lib.dev_get(n) for n in ("asv_ticks_stream", "asv_ticks_wait", "asv_ticks_cells")]
    wg = 256
    print(f"{a[4:]} asv {n2} x ({nr1} + {n2}), g={g}, sigma={sigma}: call {dt:.3f} s, kernels {kms:.1f} ms; per workgroup (of {wg}): "
          f"stream {tk[0] / wg / 1e5:.1f} ms, barrier {tk[1] / wg / 1e5:.1f} ms, cells {tk[2] / wg / 1e5:.1f} ms; "
          f"tally {[_lib.dev_get(n) for n in ('asv_literal_cells', 'asv_fallback_cells', 'asv_tiled_cells')]}", flush=True)
