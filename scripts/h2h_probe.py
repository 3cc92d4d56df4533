"""Developer helper (GPU box): the one-shot call (bmx_fast_mnn + pairs) on config 3, host to host, four times; with
BMX_DEBUG=t the library prints its own split (upload / run / download).   BMX_DEBUG=t python scripts/h2h_probe.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import synth_batches
from batchelor_amd.reduced_mnn import fast_mnn_one_shot
B = [np.asfortranarray(b) for b in synth_batches(3, [100000]*8, 50)]
for i in range(4):
    t=time.perf_counter(); r=fast_mnn_one_shot(B, k=20, c_order=False); print("one-shot %.2f ms" % (1e3*(time.perf_counter()-t)), flush=True); del r
