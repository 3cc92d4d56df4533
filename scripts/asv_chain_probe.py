"""Developer probe (CPU, see asv_chain_probe.c): python scripts/asv_chain_probe.py [n1 n2 d ncells]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libasvprobe.so"
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", os.path.join(here, "asv_chain_probe.c"),
                       "-o", so, "-lm"])
L = ctypes.CDLL(so)
n1, n2, d, nc = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1237, 1003, 100, 1003)))
rng = np.random.default_rng(100032)
spec = 1.0 / np.sqrt(1.0 + np.arange(d) / 5.0)
d1 = rng.standard_normal((n1, d)) * spec
d2 = rng.standard_normal((n2, d)) * spec + 0.3
cv = rng.standard_normal((n2, d)) * 0.2 - 0.3
cells = np.sort(rng.choice(n2, size=min(nc, n2), replace=False)).astype(np.int32)
r1 = np.arange(n1, dtype=np.int32)
r2 = np.arange(n2, dtype=np.int32)
f64p = ctypes.POINTER(ctypes.c_double)
i32p = ctypes.POINTER(ctypes.c_int32)
i64p = ctypes.POINTER(ctypes.c_int64)
cvf = np.asfortranarray(cv)
for sigma in (float(x) for x in (sys.argv[5:] or (10, 3, 1, 0.5, 0.3, 0.1, 0.03, 0.01))):
    full = np.zeros(cells.size)
    sub = np.zeros(cells.size)
    K = np.zeros((cells.size, 4), dtype=np.int64)
    ul = np.zeros(cells.size)
    L.probe_cells(d1.ctypes.data_as(f64p), d2.ctypes.data_as(f64p), d, n1, n2, cvf.ctypes.data_as(f64p), ctypes.c_double(sigma),
                  r1.ctypes.data_as(i32p), n1, r2.ctypes.data_as(i32p), n2, cells.ctypes.data_as(i32p), cells.size,
                  full.ctypes.data_as(f64p), sub.ctypes.data_as(f64p), K.ctypes.data_as(i64p), ul.ctypes.data_as(f64p))
    same = np.array_equal(full, sub, equal_nan=True)
    q = lambda a: [int(x) for x in np.percentile(a, [50, 90, 99, 100])]
    print(f"sigma {sigma}: subset chains bitwise equal to full: {same} ({(full == sub).mean():.4f}); "
          f"K own {q(K[:, 0])} ref restrict-order {q(K[:, 1])} ref sorted-order {q(K[:, 2])} within 38 of max {q(K[:, 3])}; "
          f"(tot1 - target)/ulp p10/50/90 {[float(f'{x:.3g}') for x in np.percentile(ul, [10, 50, 90])]}", flush=True)
