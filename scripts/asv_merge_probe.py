"""Developer helper (GPU box): merge-level agreement of var_adj runs with the oracle (share of right cells within 1e-5)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import batchelor_amd as bx  # noqa: E402
from batchelor_amd import _lib  # noqa: E402
from oracle import fastmnn_oracle as orc  # noqa: E402
from tests.conftest import synth_batches  # noqa: E402
from tests.test_gpu_config5 import balanced_tree  # noqa: E402

for d in (12, 100):
    B = synth_batches(5, [1200, 900], d)
    for sigma in (1.0, 0.3, 0.1):
        ref = orc.reduced_mnn(*B, var_adj=True, sigma=sigma)
        for fast in (0, 1):
            _lib.dev_set("asv_fast", fast)
            out = bx.reducedMNN(*B, var_adj=True, sigma=sigma)
            close = np.isclose(out.corrected[1200:], ref.corrected[1200:], rtol=1e-5, atol=1e-9).all(axis=1)
            print(f"2 batches d={d} sigma {sigma} asv_fast {fast}: {close.mean():.4f} of right cells agree", flush=True)
rng = np.random.Generator(np.random.PCG64(20250314 + 5001))
sizes = [int(x) for x in np.exp(rng.uniform(np.log(150), np.log(2500), 16))]
tree = balanced_tree([int(i) + 1 for i in np.argsort(sizes)[::-1]])
B = synth_batches(5, sizes, 100)
for sigma in (1.0, 0.1):
    ref = orc.reduced_mnn(*B, merge_order=tree, var_adj=True, sigma=sigma)
    for fast in (0, 1):
        _lib.dev_set("asv_fast", fast)
        out = bx.reducedMNN(*B, merge_order=tree, var_adj=True, sigma=sigma)
        close = np.isclose(out.corrected, ref.corrected, rtol=1e-5, atol=1e-9).all(axis=1)
        same_pairs = [np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(out.merge_info.pairs, ref.merge_info.pairs)]
        print(f"16-batch tree sigma {sigma} asv_fast {fast}: {close.mean():.4f} of cells agree; merges with identical pairs {sum(same_pairs)}/15", flush=True)
