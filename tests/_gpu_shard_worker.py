"""Worker for tests/test_gpu_dist.py: one rank of a 2-rank sharded engine run.  Both ranks share GPU 0 on the 1-GPU
test box, so the exchange goes through gloo (host staging); on a multi-GPU node the same code path runs over RCCL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else ""
    var_adj, auto = mode == "var_adj", mode == "auto"
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import batchelor_amd as bx
    from batchelor_amd.dist import TorchExchange
    from tests.conftest import synth_batches

    B = synth_batches(13, [3001, 2500, 1777], 50) if not auto else synth_batches(14, [900, 1400, 700, 1100, 800], 20)
    if mode.startswith("retry"):
        from tests.test_gpu_dist import retry_batches
        B = retry_batches()
        auto = mode == "retry_auto"
    eng = bx.MnnEngine(0)
    ex = TorchExchange(0)
    eng.set_shard(rank, world, ex)
    eng.upload(B)
    eng.set_profiling(True)
    eng.run(var_adj=var_adj, sigma=1.0, auto_merge=auto)
    out = eng.download()
    retries = eng.profile_detail()["optimistic_retries"]
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), corrected=out.corrected,
             pl0=out.merge_info.pairs[0][0], pr0=out.merge_info.pairs[0][1],
             pl1=out.merge_info.pairs[1][0], pr1=out.merge_info.pairs[1][1],
             lost_var=out.merge_info.lost_var, calls=ex.calls, retries=retries,
             left=np.asarray([sum(1 << (b - 1) for b in s_) for s_ in out.merge_info.left]),
             right=np.asarray([sum(1 << (b - 1) for b in s_) for s_ in out.merge_info.right]))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
