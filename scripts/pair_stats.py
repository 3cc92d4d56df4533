"""Developer helper: per merge of config 3, how many distinct left / right cells take part in a pair."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.conftest import synth_batches
import batchelor_amd as bx
B = synth_batches(3, [100000] * 8, 50)
out = bx.reducedMNN(*B, k=20)
for m, (l, r) in enumerate(out.merge_info.pairs):
    print("merge", m + 1, "pairs", l.size, "distinct left", np.unique(l).size, "distinct right", np.unique(r).size)
