"""Developer helper (GPU box): random reducedMNN configurations through the C ABI against the CPU oracle (pairs bit for
bit, coordinates to 1e-5 relative).   python scripts/engine_stress.py <cases> <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import synth_batches  # noqa: E402
from tests.test_gpu_engine import assert_same_result  # noqa: E402
from tests.test_gpu_degenerate_k import assert_same_up_to_twins  # noqa: E402
import batchelor_amd as bx  # noqa: E402
from oracle import fastmnn_oracle as oracle  # noqa: E402

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    nb = int(rng.integers(2, 6))
    sizes = [int(rng.choice([60, 150, 400, 900, 2000, 4500])) for _ in range(nb)]
    d = int(rng.choice([2, 5, 10, 20, 30, 50, 64, 80, 100]))
    kw = {}
    mode = int(rng.integers(0, 7))
    if mode == 1:
        # (k <= 2 with more than two batches is degenerate -- a cell with one pair lands on its partner up to an ulp and
        # the next merge has to tell the two apart: checked up to such twins, tests/test_gpu_degenerate_k.py)
        kw["k"] = int(rng.choice([1, 2, 5, 10, 25, 30, 40, 70, 120]))
    elif mode == 2:
        kw["prop_k"] = float(rng.choice([0.01, 0.05, 0.1]))
    elif mode == 3:
        kw["merge_order"] = [int(x) for x in rng.permutation(nb) + 1]
    elif mode == 4:
        kw["auto_merge"] = True
    elif mode == 5:
        kw["restrict"] = [np.sort(rng.choice(n, size=max(30, n // 2), replace=False)) + 1 for n in sizes]
    elif mode == 6:  # any R subsetting vector: unsorted, cells named more than once
        kw["restrict"] = [rng.choice(n, size=max(30, n // 2), replace=True) + 1 for n in sizes]
    print("case", case, sizes, d, {k: (v if k != "restrict" else "...") for k, v in kw.items()}, flush=True)
    B = synth_batches(2000 + seed * 1000 + case, sizes, d)
    try:
        ref = oracle.reduced_mnn(*B, **kw)
    except Exception as exc:  # noqa: BLE001
        try:
            bx.reducedMNN(*B, **kw)
            bad += 1
            print("MISMATCH: the oracle raised", repr(exc), "the engine did not", flush=True)
        except Exception as exc2:  # noqa: BLE001
            print("  both raised:", str(exc)[:60], "|", str(exc2)[:60], flush=True)
        continue
    try:
        out = bx.reducedMNN(*B, **kw)
        if nb > 2 and kw.get("k", 20) <= 2:
            assert_same_up_to_twins(out, ref)
        else:
            assert_same_result(out, ref)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print("MISMATCH", repr(exc)[:300], flush=True)
print("cases", cases, "mismatches", bad, flush=True)
