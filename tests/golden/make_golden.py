"""Generates tests/golden/config1_reduced_mnn.npz from the CPU oracle (run in the build container:
`python tests/golden/make_golden.py`).  The reference itself (R) cannot run here, so this fixture pins the GPU path
to the oracle, which in turn is pinned to the reference's known-answer tests (tests/test_oracle_kat.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fastmnn_oracle as orc  # noqa: E402
from tests.conftest import synth_batches  # noqa: E402

B = synth_batches(1, [2000, 2000], 50)
ref = orc.reduced_mnn(*B)
rows = np.arange(0, 4000, 8)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config1_reduced_mnn.npz"), rows=rows,
                    corrected_rows=ref.corrected[rows], pairs_left=ref.merge_info.pairs[0][0].astype(np.int32),
                    pairs_right=ref.merge_info.pairs[0][1].astype(np.int32), lost_var=ref.merge_info.lost_var,
                    batch_size=ref.merge_info.batch_size)
print("pairs", ref.merge_info.pairs[0][0].size)
