"""Generates the fixtures under tests/golden/ from the CPU oracle, in the build container:

    python tests/golden/make_golden.py config1          (seconds)   -> config1_reduced_mnn.npz
    python tests/golden/make_golden.py config2 [thr]    (~15 min on 8 cores)  -> config2_full_reduced_mnn.npz
    python tests/golden/make_golden.py config3 [thr]    (hours on 8 cores, once) -> config3_full_reduced_mnn.npz

The reference itself (R) cannot run here, so these fixtures pin the GPU path to the oracle, which in turn is pinned to the
reference's known-answer tests (tests/test_oracle_kat.py).  config2 / config3 are BASELINE.json's configurations at FULL
size (2 and 8 batches of 100 000 cells x 50 PCs, k = 20): the fixture holds, per merge, the number of MNN pairs and the
sha256 of the ordered pair arrays (int32, 1-based output rows: bit-exactness of every pair and of their order), every 64th
corrected row (float64: the 1e-5 bar of the north star), lost.var and batch.size -- so that the -m gpu tests can hold the
engine's WHOLE result at the sizes the metric is quoted on without the oracle's hours on the GPU box."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fastmnn_oracle as orc  # noqa: E402
from tests.conftest import synth_batches  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def pair_digest(left, right):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(left, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(right, dtype=np.int32).tobytes())
    return h.hexdigest()


def full_size(config, nbatches, nthreads):
    B = synth_batches(config, [100_000] * nbatches, 50)
    t0 = time.time()
    ref = orc.reduced_mnn(*B, k=20, nthreads=nthreads)
    rows = np.arange(0, 100_000 * nbatches, 64)
    np.savez_compressed(
        os.path.join(GOLD, f"config{config}_full_reduced_mnn.npz"), rows=rows, corrected_rows=ref.corrected[rows],
        npairs=np.asarray([p[0].size for p in ref.merge_info.pairs], dtype=np.int64),
        pair_sha256=np.asarray([pair_digest(*p) for p in ref.merge_info.pairs]),
        # (a few pairs of every merge in the clear, for a readable failure)
        pairs_head=np.asarray([np.stack([p[0][:16], p[1][:16]]) for p in ref.merge_info.pairs], dtype=np.int32),
        lost_var=ref.merge_info.lost_var, batch_size=ref.merge_info.batch_size,
        corrected_abs_max=np.abs(ref.corrected).max(axis=0))
    print(f"config {config}: {[p[0].size for p in ref.merge_info.pairs]} pairs, {time.time() - t0:.0f} s", flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "config1"
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if what == "config1":
        B = synth_batches(1, [2000, 2000], 50)
        ref = orc.reduced_mnn(*B)
        rows = np.arange(0, 4000, 8)
        np.savez_compressed(os.path.join(GOLD, "config1_reduced_mnn.npz"), rows=rows, corrected_rows=ref.corrected[rows],
                            pairs_left=ref.merge_info.pairs[0][0].astype(np.int32),
                            pairs_right=ref.merge_info.pairs[0][1].astype(np.int32), lost_var=ref.merge_info.lost_var,
                            batch_size=ref.merge_info.batch_size)
        print("pairs", ref.merge_info.pairs[0][0].size)
    elif what == "config2":
        full_size(2, 2, nthreads)
    elif what == "config3":
        full_size(3, 8, nthreads)
    else:
        raise SystemExit("config1 | config2 | config3")


if __name__ == "__main__":
    main()
