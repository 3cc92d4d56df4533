"""Developer helper (GPU box): which cells of the edge-case test differ from the oracle, and what they have in common."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batchelor_amd import _lib, natives as nat  # noqa: E402
from oracle import fastmnn_oracle as orc  # noqa: E402

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
rng = np.random.default_rng(100033)
d1 = rng.standard_normal((100, 1237)) / np.sqrt(1.0 + np.arange(100) / 5.0)[:, None]
d2 = rng.standard_normal((100, 1003)) / np.sqrt(1.0 + np.arange(100) / 5.0)[:, None] + 0.3
cv = rng.standard_normal((1003, 100)) * 0.2
cv[17] = 0.0
r1 = np.concatenate([rng.permutation(1237)[:901], rng.integers(0, 1237, 40)])
r2 = np.concatenate([rng.permutation(1003)[:777], rng.integers(0, 1003, 60)])
ref = orc.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
exact = nat.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
print("exact form equal to oracle:", np.array_equal(exact, ref, equal_nan=True))
_lib.dev_set("asv_fast", 1)
mult2 = np.bincount(r2, minlength=1003)
for cap in (-1, 0):
    _lib.dev_set("asv_cap", cap)
    _lib.dev_get("asv_tally_reset")
    out = nat.adjust_shift_variance(d1, d2, cv, sigma, r1, r2)
    t = [_lib.dev_get(n) for n in ("asv_literal_cells", "asv_fallback_cells", "asv_tiled_cells")]
    bad = np.flatnonzero(~np.isclose(out, ref, rtol=1e-8, atol=1e-12, equal_nan=True))
    print(f"cap {cap}: tally {t}; {bad.size} cells differ; multiplicity of the differing cells in restrict2: "
          f"{np.bincount(mult2[bad], minlength=4)} (all cells: {np.bincount(mult2, minlength=4)})")
    for c in bad[:12]:
        print(f"   cell {c} x{mult2[c]} in restrict2: out {out[c]!r} ref {ref[c]!r}")
