"""Developer helper (GPU box): share of cells on which the tiled adjust_shift_variance equals the CPU oracle, by sigma, on the
reference's test shapes and on a 100-dimension shape.   python scripts/asv_parity_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batchelor_amd import _lib, natives as nat  # noqa: E402
from oracle import fastmnn_oracle as orc  # noqa: E402

_lib.dev_set("asv_fast", 1)
rng = np.random.default_rng(100032)
data1 = rng.standard_normal((25, 400)) * 0.1
data2 = rng.standard_normal((25, 1000)) * 0.1
corvect = rng.random((1000, 25))
d1 = rng.standard_normal((100, 1237)) / np.sqrt(1.0 + np.arange(100) / 5.0)[:, None]
d2 = rng.standard_normal((100, 1003)) / np.sqrt(1.0 + np.arange(100) / 5.0)[:, None] + 0.3
cv = rng.standard_normal((1003, 100)) * 0.2
for name, (a, b, v) in {"25 x (400, 1000), scale 0.1": (data1, data2, corvect), "100 x (1237, 1003)": (d1, d2, cv)}.items():
    for sigma in (10.0, 1.0, 0.3, 0.1, 0.03, 0.01):
        out = nat.adjust_shift_variance(a, b, v, sigma, np.arange(a.shape[1]), np.arange(b.shape[1]))
        ref = orc.adjust_shift_variance(a, b, v, sigma, np.arange(a.shape[1]), np.arange(b.shape[1]))
        close = np.isclose(out, ref, rtol=1e-8, atol=1e-12, equal_nan=True)
        bad = np.flatnonzero(~close)
        print(f"{name} sigma {sigma}: equal on {close.mean():.4f} of {close.size} cells; first differing: "
              f"{[(int(i), float(out[i]), float(ref[i])) for i in bad[:3]]}", flush=True)
