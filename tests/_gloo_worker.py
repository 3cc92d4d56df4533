"""Worker for tests/test_cpu_dist_gloo.py: one rank of a world_size-2 gloo group on CPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sharded_knn(knn_fn, X, Q, k, exchange_tensor_fn, rank, world, shard_range):
    """Host-side statement of the engine's sharded search: rank-local kNN on the rank's query slice (the product's
    bmx_shard_range), then the in-place all-gather of the padded per-rank slices through the product's exchange."""
    import torch
    nq = Q.shape[0]
    per = (nq + world - 1) // world
    b, e = shard_range(nq, rank, world)
    idx = np.zeros((per * world, k), dtype=np.int32)
    dist = np.zeros((per * world, k), dtype=np.float64)
    if e > b:
        i, dd = knn_fn(X, Q[b:e], k)
        idx[b:e], dist[b:e] = i, dd
    for arr in (idx, dist):
        t = torch.from_numpy(arr.reshape(-1).view(np.uint8))
        exchange_tensor_fn(t, per * k * arr.itemsize)
    return idx[:nq], dist[:nq]


def main():
    import torch
    import torch.distributed as dist
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from batchelor_amd.dist import TorchExchange, shard_range
    from oracle import fastmnn_oracle as orc
    from tests.conftest import synth_batches

    ex = TorchExchange.__new__(TorchExchange)           # CPU box: no cuda device to bind
    ex.torch, ex.dist, ex.group = torch, dist, None
    ex.rank, ex.world, ex.backend = rank, world, "gloo"
    ex.calls = ex.bytes = 0

    X, Q = synth_batches(5, [1500, 1001], 20)            # 1001: the last rank's slice is shorter (padding path)

    def knn_fn(Xr, Qs, k):
        i, d = orc.query_knn(Xr, Qs, k)
        return i - 1, d

    idx, dd = sharded_knn(knn_fn, X, Q, 20, ex.allgather_tensor_, rank, world, shard_range)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), idx=idx, dist=dd, calls=ex.calls)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
