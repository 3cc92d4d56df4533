"""Developer helper (GPU box): the one-shot call (bmx_fast_mnn + pairs) on config 3, host to host, six times; with
BMX_DEBUG=t the library prints its own split (upload / run / download); the Python side's steps after the call are timed
here.   BMX_DEBUG=t python scripts/h2h_probe.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import synth_batches
from batchelor_amd import reduced_mnn as rm
acc = {}
def timed(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + 1e3 * (time.perf_counter() - t)
    setattr(cls, name, g)
for n in ("_pairs", "merge_stats", "close"):
    timed(rm.MnnEngine, n)
B = [np.asfortranarray(b) for b in synth_batches(3, [100000]*8, 50)]
for i in range(6):
    acc.clear()
    t = time.perf_counter(); r = rm.fast_mnn_one_shot(B, k=20, c_order=False)
    print("one-shot %.2f ms  (python side: %s)" % (1e3 * (time.perf_counter() - t), ", ".join("%s %.2f" % kv for kv in acc.items())), flush=True)
    del r
