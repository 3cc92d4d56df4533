"""CPU check of the lemma the tiled adjust_shift_variance's literal re-run rests on (batchelor_amd/csrc/legacy.hip, in front of
asv_noop_below): an addend far enough below a logspace_add chain's running value is an exact no-op, so
the chain over the addends that are NOT provably no-ops -- bounds taken from the largest addend so far, the count, and, once
the cell's own weight 1 has gone by, the largest other weight -- has the bits of the full chain
(src/adjust_shift_variance.cpp:96-109, :127-131, :147-151).  scripts/asv_chain_probe.c runs the oracle's chains and the
restricted chains side by side; here: every cell of random shapes at bandwidths across the regimes must come out bit-equal,
and at small bandwidths the restricted chains must be SHORT (that is what makes the re-run affordable at scale)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("asvprobe") / "libasvprobe.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC",
                           os.path.join(ROOT, "scripts", "asv_chain_probe.c"), "-o", so, "-lm"])
    return ctypes.CDLL(so)


@pytest.mark.parametrize("d,n1,n2,scale", [(100, 900, 400, 1.0), (25, 400, 300, 0.1), (12, 1500, 200, 1.0)])
def test_restricted_chains_have_the_full_chains_bits(probe, d, n1, n2, scale):
    rng = np.random.default_rng(100 + d)
    spec = scale / np.sqrt(1.0 + np.arange(d) / 5.0)
    d1 = rng.standard_normal((n1, d)) * spec
    d2 = rng.standard_normal((n2, d)) * spec + 0.3 * scale
    cv = np.asfortranarray(rng.standard_normal((n2, d)) * 0.2 - 0.3)
    # restrict vectors in arbitrary order, with omissions and repeats (a cell named twice: its later occurrences are addends)
    r1 = np.concatenate([rng.permutation(n1)[:(2 * n1) // 3], rng.integers(0, n1, 9)]).astype(np.int32)
    r2 = np.concatenate([rng.permutation(n2)[:(2 * n2) // 3], rng.integers(0, n2, 9)]).astype(np.int32)
    cells = np.arange(n2, dtype=np.int32)
    f64p, i32p, i64p = (ctypes.POINTER(t) for t in (ctypes.c_double, ctypes.c_int32, ctypes.c_int64))
    kept = {}
    for sigma in (10.0, 1.0, 0.3, 0.1, 0.03, 0.003):
        s2 = sigma * scale * scale
        full, sub = np.zeros(n2), np.zeros(n2)
        K = np.zeros((n2, 4), dtype=np.int64)
        ul = np.zeros(n2)
        probe.probe_cells(d1.ctypes.data_as(f64p), d2.ctypes.data_as(f64p), d, n1, n2, cv.ctypes.data_as(f64p), ctypes.c_double(s2),
                          r1.ctypes.data_as(i32p), r1.size, r2.ctypes.data_as(i32p), r2.size, cells.ctypes.data_as(i32p), n2,
                          full.ctypes.data_as(f64p), sub.ctypes.data_as(f64p), K.ctypes.data_as(i64p), ul.ctypes.data_as(f64p))
        assert np.array_equal(full, sub, equal_nan=True), (sigma, int((full != sub).sum()))
        kept[sigma] = np.median(K[:, :3], axis=0)
    # where the bandwidth is small against the squared distances only a handful of addends can change a chain
    assert kept[0.003].max() <= 0.1 * r1.size, kept
    assert kept[10.0].max() >= 0.5 * r2.size      # ... and where it is large, nearly every addend does
