"""Developer helper (GPU box): what the host gives a process -- affinity, cgroup CPU quota -- and how the CPU baseline T
(oracle/tiled_knn_baseline.c) scales with its thread count.   python scripts/host_cpu_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_baselines as cb  # noqa: E402

print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "physical", cb.physical_cores())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
rng = np.random.default_rng(0)
X = rng.standard_normal((100000, 50))
Q = rng.standard_normal((50000, 50))
cb.tiled_knn(X[:5000], Q[:500], 20, 8)
for nt in (1, 8, 16, 32, 64, 128, 256):
    t = time.perf_counter()
    cb.tiled_knn(X, Q if nt > 1 else Q[:4000], 20, nt)
    dt = time.perf_counter() - t
    nq = Q.shape[0] if nt > 1 else 4000
    print(f"threads {nt}: {2 * 50 * 1e5 * nq / dt / 1e9:.0f} GFLOP/s ({dt:.2f} s)", flush=True)
