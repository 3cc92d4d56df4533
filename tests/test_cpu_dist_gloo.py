"""The N > 1 path on CPU: two gloo ranks shard a kNN search by query rows exactly as the engine does
(bmx_shard_range + padded contiguous slices), all-gather the neighbour lists through batchelor_amd.dist, and must
both end with the unsharded answer.  The arithmetic inside each rank is the CPU oracle here (no GPU on this box);
what is under test is the product's partition + exchange code."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharded_knn_equals_single_rank(tmp_path, oracle):
    import __graft_entry__ as g
    g.build()
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), str(r), "2", str(port),
                               str(tmp_path)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    from tests.conftest import synth_batches
    X, Q = synth_batches(5, [1500, 1001], 20)
    ref_i, ref_d = oracle.query_knn(X, Q, 20)
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["idx"] + 1, ref_i)
        assert np.array_equal(got["dist"], ref_d)
        assert int(got["calls"]) == 2
