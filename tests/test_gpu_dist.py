"""Sharded engine on real hardware: two ranks (sharing the box's one GPU, gloo exchange) must both produce exactly
the single-rank result -- pairs bit-identical, coordinates bit-identical (the exchange is the only difference)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,var_adj", [(2, False), (4, False), (2, True), (4, True)])
def test_n_rank_engine_equals_single_rank(tmp_path, world, var_adj):
    import batchelor_amd as bx
    from tests.conftest import synth_batches
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_shard_worker.py"), str(r), str(world),
                               str(port), str(tmp_path)] + (["var_adj"] if var_adj else []), env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    B = synth_batches(13, [3001, 2500, 1777], 50)
    # var_adj: every right cell's scaling (adjust_shift_variance, independent per cell) is computed by the rank that owns
    # the cell and all-gathered -- one more exchange per merge
    ref = bx.reducedMNN(*B, var_adj=var_adj, sigma=1.0)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["corrected"], ref.corrected)
        for m in range(2):
            assert np.array_equal(got[f"pl{m}"], ref.merge_info.pairs[m][0])
            assert np.array_equal(got[f"pr{m}"], ref.merge_info.pairs[m][1])
        assert np.array_equal(got["lost_var"], ref.merge_info.lost_var)
        # per merge: index + distance gathers of the first search, index + k-th distance of the second, with each of them the
        # ranks' "could not be completed on the device" flags; the tricube search exchanges its flags only -- every rank
        # corrects its own slice of the right cells and the corrected ROWS are gathered (with var_adj the lists are, as every
        # rank needs all correction vectors, + the scalings)
        # (round 6: + the column sums of the first averaging, whose workgroups are dealt over the ranks)
        assert int(got["calls"]) == 2 * (3 + 3 + 1 + (3 + 1 if var_adj else 2))
        assert int(got["retries"]) == 0


def retry_batches():
    """Three batches whose FIRST half is 30 clusters of 100 near-duplicates (thousands of references inside the fp16 pass's
    error margin of a query's k-th neighbour: the search cannot be completed on the device) and whose second half is a
    diffuse cloud far away (certified at once): with two ranks the flag goes up on rank 0's query rows only."""
    rng = np.random.default_rng(4242)
    centres = rng.standard_normal((30, 50)) * 2.0
    out = []
    for b in range(3):
        clus = np.repeat(centres, 100, axis=0) + 1e-4 * rng.standard_normal((3000, 50)) + 0.3 * b
        diffuse = rng.standard_normal((3000, 50)) + 0.3 * b
        diffuse[:, 0] += 60.0
        out.append(np.vstack([clus, diffuse]))
    return out


@pytest.mark.parametrize("mode", ["retry", "retry_auto"])
def test_ranks_restart_an_optimistic_run_together(tmp_path, mode):
    """Several ranks run optimistically too: a search one rank cannot complete on the device raises a flag that travels with
    the search's lists (or, for the searches auto-merge deals out whole, with their counts), every rank sees it at the same
    wait and all of them repeat the run with host-checked searches.  A rank restarting alone would hang the others in a
    collective: the workers have a timeout.  Same result as one rank, bit for bit; the restart did happen, on both."""
    import batchelor_amd as bx
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_shard_worker.py"), str(r), "2", str(port),
                               str(tmp_path), mode], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    B = retry_batches()
    ref = bx.reducedMNN(*B, auto_merge=mode == "retry_auto")
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert int(got["retries"]) >= 1
        assert np.array_equal(got["corrected"], ref.corrected)
        for m in range(2):
            assert np.array_equal(got[f"pl{m}"], ref.merge_info.pairs[m][0])
            assert np.array_equal(got[f"pr{m}"], ref.merge_info.pairs[m][1])


@pytest.mark.parametrize("world", [2, 3])
def test_auto_merge_counts_are_dealt_over_the_ranks(tmp_path, world):
    """auto.merge = TRUE (R/MNN_tree.R:154-226): the B (B - 1) / 2 initial MNN-pair counts and the recounts after every merge are
    independent searches -- with several ranks each count runs whole on one rank (round robin) and the numbers are
    all-gathered; the merges themselves are row-sharded as ever.  Same merge order, same pairs, same coordinates as one rank."""
    import batchelor_amd as bx
    from tests.conftest import synth_batches
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_shard_worker.py"), str(r), str(world),
                               str(port), str(tmp_path), "auto"], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    B = synth_batches(14, [900, 1400, 700, 1100, 800], 20)
    ref = bx.reducedMNN(*B, auto_merge=True)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert list(got["left"]) == [sum(1 << (b - 1) for b in s_) for s_ in ref.merge_info.left]
        assert list(got["right"]) == [sum(1 << (b - 1) for b in s_) for s_ in ref.merge_info.right]
        assert np.array_equal(got["corrected"], ref.corrected)
        for m in range(2):
            assert np.array_equal(got[f"pl{m}"], ref.merge_info.pairs[m][0])
            assert np.array_equal(got[f"pr{m}"], ref.merge_info.pairs[m][1])


def test_nccl_exchange_aliases_raw_device_pointer():
    """World-size-1 RCCL group: the production transport (all_gather_into_tensor on a tensor aliasing a raw device
    pointer handed over by the engine) runs and really aliases the buffer."""
    code = r'''
import os, sys
sys.path.insert(0, os.environ["BMX_ROOT"])
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["BMX_PORT"], rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
from batchelor_amd.dist import TorchExchange, _RawDeviceBuffer
ex = TorchExchange(0)
t = torch.arange(4096, dtype=torch.uint8, device="cuda") % 251
ref = t.clone()
alias = torch.as_tensor(_RawDeviceBuffer(t.data_ptr(), t.numel()), device=torch.device("cuda", 0))
alias[7] = 200
assert int(t[7]) == 200
ref[7] = 200
ex(t.data_ptr(), t.numel())
assert torch.equal(t, ref) and ex.calls == 1
dist.destroy_process_group()
print("nccl-ok")
'''
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BMX_ROOT=ROOT, BMX_PORT=str(port))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "nccl-ok" in out.stdout, out.stderr[-2000:]


def test_engine_owned_rccl_communicator_world_size_one():
    """The production exchange: RCCL called from inside the engine, in place on its stream.  One GPU on the test box, so
    the communicator has one rank; the testing hook "exchange_always" makes that rank go through ncclAllGather for every
    list anyway.
    The result must be the plain engine's, bit for bit, and the gathers must have happened."""
    code = r'''
import os, sys
sys.path.insert(0, os.environ["BMX_ROOT"])
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["BMX_PORT"], rank=0, world_size=1)
import batchelor_amd as bx
from batchelor_amd import _lib
from batchelor_amd.dist import init_engine_rccl
from tests.conftest import synth_batches
B = synth_batches(13, [3001, 2500, 1777], 50)
ref = bx.reducedMNN(*B)
_lib.dev_set("exchange_always", 1)
eng = bx.MnnEngine(0)
assert init_engine_rccl(eng) == (0, 1)
eng.upload(B)
eng.run()
got = eng.download()
st = eng.exchange_stats()
assert st["calls"] == 2 * (2 + 2 + 1 + 1), st  # per merge: index + distance, index + k-th distance, the first averaging's column sums, the tricube-corrected rows
assert np.array_equal(got.corrected, ref.corrected)
for (a, b), (c, d) in zip(got.merge_info.pairs, ref.merge_info.pairs):
    assert np.array_equal(a, c) and np.array_equal(b, d)
eng.close()
dist.destroy_process_group()
print("rccl-engine-ok")
'''
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BMX_ROOT=ROOT, BMX_PORT=str(port), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl-engine-ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_emulated_rank_of_n_equals_the_single_rank_run():
    """bmx_engine_emulate (bench.py --emulate-world): one rank's share of an N-rank run on one GPU -- its slice of every search,
    the other ranks' slices of every exchange replayed from a recorded single-rank run.  Whatever the rank and the world size,
    the emulated rank must end with the recorded run's result bit for bit (else what bench.py times is not the job)."""
    import batchelor_amd as bx
    from tests.conftest import synth_batches
    B = synth_batches(13, [3001, 2500, 1777], 50)
    eng = bx.MnnEngine(0)
    eng.upload(B)
    eng.run()
    base = eng.download()
    with pytest.raises(bx.BatchelorMI355XError, match="more exchanges than the recorded run"):
        eng.emulate(2, 1, 4)          # nothing recorded yet
        eng.run()
    eng.emulate(1)
    eng.run()
    for world, rank in [(2, 0), (2, 1), (3, 2), (4, 1), (8, 7)]:
        eng.emulate(2, rank, world)
        eng.run()
        got = eng.download()
        assert np.array_equal(got.corrected, base.corrected), (world, rank)
        for (a, b), (c, d) in zip(got.merge_info.pairs, base.merge_info.pairs):
            assert np.array_equal(a, c) and np.array_equal(b, d)
        assert eng.exchange_stats()["calls"] == 2 * (3 + 3 + 1 + 2)   # (the same exchanges a real rank makes)
    with pytest.raises(bx.BatchelorMI355XError, match="auto-merge"):
        eng.run(auto_merge=True)
    with pytest.raises(bx.BatchelorMI355XError, match="invalid rank"):
        eng.emulate(2, 4, 4)
    eng.emulate(0)
    eng.run()
    assert np.array_equal(eng.download().corrected, base.corrected)
    eng.close()
