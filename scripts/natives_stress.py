"""Developer helper (GPU box): the single-step primitives and the three legacy natives on random shapes against the
oracle.   python scripts/natives_stress.py <cases> <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batchelor_amd import natives as nat  # noqa: E402
from oracle import fastmnn_oracle as oracle  # noqa: E402

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0


def check(name, fn):
    global bad
    try:
        fn()
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print("MISMATCH", name, repr(exc)[:300], flush=True)


for case in range(cases):
    n1, n2 = int(rng.choice([40, 333, 1000, 2500])), int(rng.choice([50, 400, 1500, 3000]))
    d = int(rng.choice([2, 7, 25, 50, 64, 90]))
    k = int(rng.choice([1, 5, 13, 20]))
    k = min(k, n1, n2)
    print("case", case, n1, n2, d, k, flush=True)
    t1 = rng.standard_normal((n1, d))
    t2 = rng.standard_normal((n2, d)) + 0.3

    def mutual():
        L = np.vstack([rng.permutation(n2)[:k] + 1 for _ in range(n1)])
        R = np.vstack([rng.permutation(n1)[:min(k + 3, n1)] + 1 for _ in range(n2)])
        f, s = nat.find_mutual_nns(L, R)
        of, os_ = oracle.find_mutual_nns(L, R)
        assert np.array_equal(f, of) and np.array_equal(s, os_)
    check("find_mutual_nns", mutual)

    def mnn_avg():
        f, s = nat.find_mutual_nn(t1, t2, k, k)
        of, os_ = oracle.find_mutual_nn(t1, t2, k, k)
        assert np.array_equal(f, of) and np.array_equal(s, os_)
        if len(of):
            f2, s2, avg, su = nat.mnn_average_correction(t1, t2, k)
            oavg, osu = oracle.average_correction(t1, of, t2, os_)
            assert np.array_equal(su, osu)
            np.testing.assert_allclose(avg, oavg, rtol=1e-12, atol=1e-14)
    check("find_mutual_nn / average_correction", mnn_avg)

    def center():
        b = rng.standard_normal(d)
        np.testing.assert_allclose(nat.center_along_batch_vector(t1, b), oracle.center_along_batch_vector(t1, b),
                                   rtol=1e-11, atol=1e-12)
    check("center_along_batch_vector", center)

    def tricube():
        m = max(k, n2 // 3)
        involved = np.sort(rng.permutation(n2)[:m]) + 1
        corr = rng.standard_normal((m, d))
        kk = min(k, m)
        out = nat.tricube_weighted_correction(t2, corr, involved, k=kk, ndist=3)
        ref = oracle.tricube_weighted_correction(t2, corr, involved, k=kk, ndist=3)
        np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-12, equal_nan=True)
    check("tricube_weighted_correction", tricube)

    def sgk():
        U = max(3, n2 // 7)
        gd, g = d, int(rng.choice([d, 3, 130]))
        mat = rng.standard_normal((gd, n2)) * 0.3
        index = rng.permutation(n2)[:U]
        averaged = rng.standard_normal((g, U))
        s2 = float(rng.choice([0.5, 0.05]))
        np.testing.assert_allclose(nat.smooth_gaussian_kernel(averaged, index, mat, s2),
                                   oracle.smooth_gaussian_kernel(averaged, index, mat, s2), rtol=1e-9, atol=1e-13)
    check("smooth_gaussian_kernel", sgk)

    def asv():
        m1, m2 = min(n1, 300), min(n2, 400)
        g = min(d, 30)
        data1 = rng.standard_normal((g, m1)) * 0.1
        data2 = rng.standard_normal((g, m2)) * 0.1
        cv = rng.random((m2, g))
        r1 = rng.permutation(m1)[:max(5, m1 // 2)]
        r2 = rng.permutation(m2)[:max(5, m2 // 2)]
        sig = float(rng.choice([1.0, 0.1]))
        assert np.array_equal(nat.adjust_shift_variance(data1, data2, cv, sig, r1, r2),
                              oracle.adjust_shift_variance(data1, data2, cv, sig, r1, r2))
    check("adjust_shift_variance", asv)

    def asv_tiled():
        # the form taken beyond 4e7 pairs (testing hook "asv_fast"): random shapes, bandwidths from 10 to 0.01 times the data's
        # scale, restrict vectors with repeats and omissions; >= 0.999 of the cells to 1e-8, and bit for bit when every cell
        # took the re-run
        from batchelor_amd import _lib
        _lib.dev_set("asv_fast", 1)
        try:
            m1, m2 = int(rng.choice([150, 700, 2500, 6000])), int(rng.choice([100, 500, 1500]))
            g = int(rng.choice([3, 12, 25, 50, 100, 130]))
            scale = float(rng.choice([0.1, 1.0]))
            spec = 1.0 / np.sqrt(1.0 + np.arange(g) / 5.0)
            data1 = (rng.standard_normal((m1, g)) * spec * scale).T
            data2 = (rng.standard_normal((m2, g)) * spec * scale + 0.3 * scale).T
            cv = rng.standard_normal((m2, g)) * 0.2 - 0.3
            r1 = np.concatenate([rng.permutation(m1)[:max(5, (2 * m1) // 3)], rng.integers(0, m1, 7)])
            r2 = np.concatenate([rng.permutation(m2)[:max(5, (2 * m2) // 3)], rng.integers(0, m2, 7)])
            sig = float(rng.choice([10.0, 1.0, 0.5, 0.3, 0.1, 0.03, 0.01])) * scale * scale
            _lib.dev_get("asv_tally_reset")
            out = nat.adjust_shift_variance(data1, data2, cv, sig, r1, r2)
            lit, back, tiled = (_lib.dev_get(n) for n in ("asv_literal_cells", "asv_fallback_cells", "asv_tiled_cells"))
            ref = oracle.adjust_shift_variance(data1, data2, cv, sig, r1, r2)
            close = np.isclose(out, ref, rtol=1e-8, atol=1e-12, equal_nan=True)
            print("   asv tiled", m1, m2, g, sig, "literal", lit, "beyond", back, "of", tiled, "equal", round(float(close.mean()), 4), flush=True)
            assert close.mean() >= 0.999, close.mean()
            if lit == tiled:
                assert np.array_equal(out, ref, equal_nan=True)
        finally:
            _lib.dev_set("asv_fast", 0)
    check("adjust_shift_variance (tiled form)", asv_tiled)
print("cases", cases, "mismatches", bad, flush=True)
