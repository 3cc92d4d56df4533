import sys, time, os, ctypes
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import synth_batches
from batchelor_amd import reduced_mnn as rm, _lib
B = [np.asfortranarray(b) for b in synth_batches(3, [100000]*8, 50)]
orig = rm.MnnEngine._pairs
def probe(self):
    L = _lib.lib()
    t0 = time.perf_counter()
    ns = []
    for m in range(self.nbatches - 1):
        n = ctypes.c_int64(0)
        _lib.check(L.bmx_engine_pairs_into(self._h, m, None, None, ctypes.c_int64(0), ctypes.byref(n)))
        ns.append(n.value)
    t1 = time.perf_counter()
    arrs = [(np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)) for n in ns]
    t2 = time.perf_counter()
    per = []
    for m, (pl, pr) in enumerate(arrs):
        n = ctypes.c_int64(0)
        ta = time.perf_counter()
        _lib.check(L.bmx_engine_pairs_into(self._h, m, _lib.i32p(pl), _lib.i32p(pr), ctypes.c_int64(ns[m]), ctypes.byref(n)))
        per.append(1e3 * (time.perf_counter() - ta))
    t3 = time.perf_counter()
    # again into the now-resident arrays
    for m, (pl, pr) in enumerate(arrs):
        n = ctypes.c_int64(0)
        _lib.check(L.bmx_engine_pairs_into(self._h, m, _lib.i32p(pl), _lib.i32p(pr), ctypes.c_int64(ns[m]), ctypes.byref(n)))
    t4 = time.perf_counter()
    print("   counts %.3f  empty %.3f  fill %.3f (%s)  refill %.3f ms; sizes %s" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), " ".join("%.2f" % x for x in per), 1e3*(t4-t3), ns), flush=True)
    return arrs
rm.MnnEngine._pairs = probe
for i in range(4):
    t = time.perf_counter(); r = rm.fast_mnn_one_shot(B, k=20, c_order=False)
    print("one-shot %.2f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
    del r
