"""Developer helper (GPU box): the tiled adjust_shift_variance alone, .Call level, on a mid-size problem; prints seconds and
ns per (cell, restricted cell) pair.   [BMX_LIB=<variant build>] python scripts/asv_probe.py [n2 nr1 nr2 g sigma]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batchelor_amd import _lib, natives as nat  # noqa: E402

n2, nr1, nr2, g, sigma = tuple(int(x) for x in sys.argv[1:5]) + (float(sys.argv[5]),) if len(sys.argv) > 5 else (100000, 200000, 100000, 100, 1.0)
rng = np.random.default_rng(7)
s = 1.0 / np.sqrt(1.0 + np.arange(g) / 5.0)
d1 = np.asfortranarray((rng.standard_normal((nr1, g)) * s).T)
d2 = np.asfortranarray((rng.standard_normal((max(n2, nr2), g)) * s + 0.3).T)
cv = rng.standard_normal((d2.shape[1], g)) * 0.2
_lib.dev_set("asv_fast", 1)
if os.environ.get("BMX_ASV_CAP") and hasattr(_lib.lib(), "bmx_dev_get"):  # (a developer script's own switch, not the library's)
    _lib.dev_set("asv_cap", int(os.environ["BMX_ASV_CAP"]))
r1, r2 = np.arange(nr1), np.arange(nr2)
for rep in range(3):
    t = time.perf_counter()
    out = nat.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
    dt = time.perf_counter() - t
    print(f"asv {d2.shape[1]} cells x ({nr1} + {nr2}) restricted, g={g}, sigma={sigma}: {dt:.3f} s, "
          f"{1e9 * dt / (d2.shape[1] * (nr1 + nr2)):.4f} ns per pair, finite {np.isfinite(out).mean():.3f}", flush=True)
if hasattr(_lib.lib(), "bmx_dev_get"):
    print("tally (literal, flagged beyond, tiled) over the 3 calls:", [_lib.dev_get(n) for n in ("asv_literal_cells", "asv_fallback_cells", "asv_tiled_cells")])
