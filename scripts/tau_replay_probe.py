"""Developer experiment (GPU box; EXPERIMENTS.md round 6): the candidate full pass of a config-3 step when every query STARTS
from the threshold it ends with (testing hook "tau_replay": a first run records every search's final thresholds, the next
runs start their full passes from them) against the normal step -- the bound on what ANY better source of starting
thresholds can win.   python scripts/tau_replay_probe.py [workload] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import batchelor_amd as bx  # noqa: E402
from batchelor_amd import _lib  # noqa: E402
from bench import WORKLOADS, synth_batches  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg, sizes, d, k, tree = WORKLOADS[wl]
B = synth_batches(cfg, sizes, d)
if tree is not None:
    from batchelor_amd.merge_tree import resolve_merge_order
    tree = resolve_merge_order(len(sizes), tree)
eng = bx.MnnEngine(0)
eng.upload(B)


def timed(tag):
    for _ in range(2):
        eng.run(k=k, merge_tree=tree)
    eng.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f = s = 0.0
    for _ in range(steps):
        eng.run(k=k, merge_tree=tree)
        p = eng.profile_detail()
        f += p["f16_ms"]
        s += p["sample_ms"]
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    eng.set_profiling(False)
    print(f"{tag}: {ms:.2f} ms per step, full passes {f / steps:.2f} ms, sample passes {s / steps:.2f} ms", flush=True)
    return eng.download(with_pairs=True)


base = timed("normal")
_lib.dev_set("tau_replay", 1)
eng.run(k=k, merge_tree=tree)
_lib.dev_set("tau_replay", 2)
rep = timed("every full pass started from its final thresholds")
_lib.dev_set("tau_replay", 0)
again = timed("normal again")
assert np.array_equal(rep.corrected, base.corrected)
for (a, b), (c, e) in zip(rep.merge_info.pairs, base.merge_info.pairs):
    assert np.array_equal(a, c) and np.array_equal(b, e)
print("results identical")
eng.close()
