"""Developer helper (GPU box): the staged calls on config 3 (create, upload, run, run again, download), each timed, next to the
one-shot call: what the lazy upload of the one-shot call leaves exposed inside its run.   python scripts/staged_probe.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import synth_batches
from batchelor_amd import reduced_mnn as rm
B = [np.asfortranarray(b) for b in synth_batches(3, [100000]*8, 50)]
ms = lambda t: 1e3 * (time.perf_counter() - t)
for i in range(4):
    t = time.perf_counter(); e = rm.MnnEngine(); a = ms(t)
    t = time.perf_counter(); e.upload(B); b = ms(t)
    t = time.perf_counter(); e.run(k=20); c = ms(t)
    t = time.perf_counter(); e.run(k=20); c2 = ms(t)
    t = time.perf_counter(); r = e.download(c_order=False); d = ms(t)
    t = time.perf_counter(); e.close(); f = ms(t)
    print("staged: create %.2f upload %.2f run %.2f run-again %.2f download(+pairs) %.2f close %.2f" % (a, b, c, c2, d, f), flush=True)
    del r
    t = time.perf_counter(); r = rm.fast_mnn_one_shot(B, k=20, c_order=False); print("one-shot %.2f" % ms(t), flush=True); del r
