import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import batchelor_amd as bx
from batchelor_amd import _lib, natives as nat
from oracle import fastmnn_oracle as orc
from tests.conftest import synth_batches
B = synth_batches(5, [1200, 900], 12)
keep = [None, np.arange(1, 801)]
if len(sys.argv) > 1:
    r0 = orc.reduced_mnn(*B, var_adj=True, sigma=1.0, restrict=keep)
for fast in (0, 1):
    _lib.dev_set("asv_fast", fast)
    eng = bx.MnnEngine(); eng.upload(B, restrict=keep); eng.set_snapshot(0); eng.run(var_adj=True, sigma=1.0)
    out, snap = eng.download(), eng.snapshot_var_adj(); eng.close()
    ref = orc.adjust_shift_variance(snap["left"].T, snap["right"].T, snap["correction"], 1.0, snap["restrict1"], snap["restrict2"])
    ref2 = orc.adjust_shift_variance(snap["left"].T, snap["right"].T, snap["correction"], 1.0, snap["restrict1"], snap["restrict2"])
    abi = nat.adjust_shift_variance(snap["left"].T, snap["right"].T, snap["correction"], 1.0, snap["restrict1"], snap["restrict2"])
    got = snap["scaling"]
    print("fast", fast, "engine vs oracle differ:", int((got != ref).sum()), "abi vs oracle differ:", int((abi != ref).sum()), "oracle twice differ:", int((ref != ref2).sum()))
    bad = np.flatnonzero(got != ref)
    if bad.size:
        print("max rel diff", np.max(np.abs(got[bad] - ref[bad]) / np.abs(ref[bad])))
        print("bad cells", bad[:20], "in restrict2 (<800):", int((bad < 800).sum()), "of", bad.size)
        for c in bad[:5]:
            print(c, repr(got[c]), repr(ref[c]), repr(abi[c]))
