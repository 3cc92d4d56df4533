"""Worker for tests/test_cpu_dist_gloo.py: one rank of a world_size-2 gloo group on CPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from batchelor_amd.dist import TorchExchange, sharded_rows
    from oracle import fastmnn_oracle as orc
    from tests.conftest import synth_batches

    ex = TorchExchange.__new__(TorchExchange)           # CPU box: no cuda device to bind
    ex.torch, ex.dist, ex.group = torch, dist, None
    ex.rank, ex.world, ex.backend = rank, world, "gloo"
    ex.calls = ex.bytes = 0

    X, Q = synth_batches(5, [1500, 1001], 20)            # 1001: the last rank's slice is shorter (padding path)

    # the product's partition + exchange (batchelor_amd.dist.sharded_rows: bmx_shard_range, bmx_shard_gather_bytes,
    # TorchExchange); only the arithmetic inside a rank's slice is the CPU oracle here (no GPU on this box)
    k = 20
    idx = sharded_rows(lambda b, e: orc.query_knn(X, Q[b:e], k)[0] - 1, Q.shape[0], (k,), np.int32, ex, rank, world)
    dd = sharded_rows(lambda b, e: orc.query_knn(X, Q[b:e], k)[1], Q.shape[0], (k,), np.float64, ex, rank, world)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), idx=idx, dist=dd, calls=ex.calls)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
