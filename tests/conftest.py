import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # The oracle is test infrastructure: build it on demand so the CPU suite is self-contained.
    so = os.path.join(ROOT, "oracle", "libmnn_oracle.so")
    src = os.path.join(ROOT, "oracle", "mnn_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def _gpu_count():
    try:
        from batchelor_amd import _lib
        return _lib.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of erroring them.  On the GPU box
    nothing is skipped: a missing extension or device there must fail loudly, which test_gpu_* do on import."""
    # /dev/kfd is the ROCm compute device node: where it exists a GPU is expected and nothing may be skipped
    if any("gpu" in item.keywords for item in items) and not os.path.exists("/dev/kfd") and _gpu_count() == 0:
        skip = pytest.mark.skip(reason="no MI355X / HIP device visible (gpu-marked test)")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture
def dev():
    """The library's testing hooks (bmx_dev_set), restored to their defaults when the test is through."""
    from batchelor_amd import _lib
    try:
        yield _lib.dev_set
    finally:
        _lib.dev_set("reset", 0)


@pytest.fixture(scope="session")
def oracle():
    from oracle import fastmnn_oracle
    return fastmnn_oracle


def synth_batches(config, sizes, d, shift=1.0):
    """Synthetic Gaussian batches, SURVEY.md 8(d): X_b = Z diag(s) + mu_b, seed = 20250314 + 1000*config + b."""
    import numpy as np
    out = []
    s = 1.0 / np.sqrt(1.0 + np.arange(d) / 5.0)
    for b, n in enumerate(sizes):
        rng = np.random.Generator(np.random.PCG64(20250314 + 1000 * config + b))
        mu = np.zeros(d)
        mu[b % d] += shift
        mu += 0.5 * b / np.sqrt(d)
        out.append(rng.standard_normal((n, d)) * s + mu)
    return out
