"""Developer helper (GPU box): config 3's batches with auto.merge = TRUE (R/MNN_tree.R:154-226: B(B-1)/2 pair counts, then a
recount of the new node against every remaining batch after each merge) against the predefined progressive order.
   python scripts/auto_merge_probe.py [n] [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import batchelor_amd as bx  # noqa: E402
from tests.conftest import synth_batches  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
batches = synth_batches(3, [n] * B, 50)
eng = bx.MnnEngine()
eng.upload(batches)
for auto in (False, True, False, True):
    eng.run(k=20, auto_merge=auto)
    t = time.perf_counter()
    eng.run(k=20, auto_merge=auto)
    dt = time.perf_counter() - t
    st = eng.merge_stats()
    print(f"B={B} n={n} auto_merge={auto}: {1e3 * dt:.1f} ms per step; merges {[ (s.get('left'), s.get('right')) for s in st][:3]}...", flush=True)
eng.close()
