#!/usr/bin/env python3
"""Headline benchmark: cells/sec corrected by the MI355X reducedMNN engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one complete engine run -- every merge of the workload: exact kNN both ways, mutual pairs, averaged
correction vectors, centring, variance bookkeeping, tricube-smoothed correction -- on inputs already resident in
HBM (bmx_engine_upload happens before the timed region; the PCIe-inclusive rate is quoted in DESIGN.md).
Workload (all N): the configuration BASELINE.json's metric and target are quoted on and that fits one GPU --
configs[2] = 8 synthetic Gaussian batches x 100 000 cells x 50 PCs, k = 20, progressive merge 1..8 (7 merges; the
m-th new batch is orthogonalised against m-1 earlier batch vectors).  `--workload config2` (2 x 100k) and `config5`
(16 unequal batches, 100 PCs, balanced tree) are available for comparison runs.
With N > 1 the same job is split by kNN query rows over the ranks (strong scaling) and the per-rank neighbour lists
are all-gathered with RCCL; every rank ends with the full result.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel knn_topk_mfma: algorithmic
FLOPs / HIP-event time vs the f32-input MFMA peak) and, at N = 1, `cpu_baseline` (the CPU oracle timed on a bounded
sample of the same workload on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Dense MFMA peaks from /opt/skills/guides/MI355X_MICROARCH.md ("Chip-level parameters" / "Matrix cores")
KERNELS = {
    3: {"kernel": "knn_topk_f16", "dtype": "fp16 MFMA candidate pass (f32 accumulate) + FP64 exact re-rank", "peak": 2500.0},
    2: {"kernel": "knn_topk_bf16", "dtype": "bf16 MFMA (f32 operands split into 3 bf16 products) + FP64 exact re-rank",
        "peak": 2500.0},
}

def _config5_sizes():
    """16 batch sizes, log-uniform in [5 000, 500 000], fixed by the seed."""
    rng = np.random.Generator(np.random.PCG64(20250314 + 5000))
    return [int(x) for x in np.exp(rng.uniform(np.log(5e3), np.log(5e5), 16))]


def _balanced_tree(ids):
    """Balanced nested merge.order over the given (1-based) batch ids."""
    if len(ids) == 1:
        return ids[0]
    h = len(ids) // 2
    return [_balanced_tree(ids[:h]), _balanced_tree(ids[h:])]


WORKLOADS = {
    # name: (config id for the seeds, batch sizes, d, k, merge tree or None for 1..B)
    "config2": (2, [100000, 100000], 50, 20, None),
    "config3": (3, [100000] * 8, 50, 20, None),
    "config1": (1, [2000, 2000], 50, 20, None),
}
_s5 = _config5_sizes()
# config 5: balanced tree over the size-sorted batches (largest first so the biggest batch is a reference leaf)
WORKLOADS["config5"] = (5, _s5, 100, 20, _balanced_tree([i + 1 for i in np.argsort(_s5)[::-1]]))


def synth_batches(config, sizes, d, shift=1.0):
    """X_b = Z diag(s) + mu_b, Z ~ N(0, I), s_j = 1/sqrt(1 + j/5), PCG64(20250314 + 1000 config + b)."""
    out = []
    s = 1.0 / np.sqrt(1.0 + np.arange(d) / 5.0)
    for b, n in enumerate(sizes):
        rng = np.random.Generator(np.random.PCG64(20250314 + 1000 * config + b))
        mu = np.zeros(d)
        mu[b % d] += shift
        mu += 0.5 * b / np.sqrt(d)
        out.append(rng.standard_normal((n, d)) * s + mu)
    return out


def measured_traffic(workload, variant):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/): FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this very command, corrected as MI355X_MICROARCH.md prescribes
    for gfx950 (FETCH_SIZE x 2).  PMC counters cannot be read from inside the timed process, so the number is the
    last committed measurement for the same workload and kernel; None when there is none."""
    path = os.path.join(ROOT, "profiles", f"r01_traffic_{workload}_bf16.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    return rec["bytes_per_launch"] if rec.get("variant") == variant and rec.get("workload") == workload else None


def algorithmic_flops(stats, d):
    """F_merge = 2 d (nL nR + nR U): one shared distance block for both kNN directions plus the
    tricube search of every right cell against the U MNN-involved right cells."""
    return sum(2.0 * d * (m["nL"] * m["nR"] + m["nR_all"] * m["U"]) for m in stats)


def cpu_baseline(batches, stats, d, k):
    """Times the CPU oracle (oracle/, "port" of the reference algorithm; the reference R/Rcpp path itself cannot run
    on this box) on a bounded sample: the three exact searches of the merge with 2048 sampled query rows each
    against the FULL reference sets, all host cores; extrapolated by pair evaluations to the whole job."""
    from oracle import fastmnn_oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    L, R = batches[0], batches[1]
    rng = np.random.default_rng(0)
    U = max(k, int(stats[0]["U"]))
    sub = R[rng.choice(R.shape[0], min(U, R.shape[0]), replace=False)]

    def timed(ns):
        ql = L[rng.choice(L.shape[0], min(ns, L.shape[0]), replace=False)]
        qr = R[rng.choice(R.shape[0], min(ns, R.shape[0]), replace=False)]
        t0 = time.perf_counter()
        orc.query_knn(R, ql, k, nthreads=cores)
        orc.query_knn(L, qr, k, nthreads=cores)
        orc.query_knn(sub, qr, k, nthreads=cores)
        dt = time.perf_counter() - t0
        return ql, qr, dt, ql.shape[0] * R.shape[0] + qr.shape[0] * L.shape[0] + qr.shape[0] * sub.shape[0]

    # calibrate with a small sample, then size the timed sample for ~15 s of CPU work (bounded by the full job)
    _, _, dt0, ev0 = timed(max(1024, 8 * cores))
    ns = int(min(L.shape[0], max(2048, 8 * cores) * max(1.0, 15.0 / max(dt0, 1e-3))))
    ql, qr, dt, sampled = timed(ns)
    rate = sampled / dt
    total = sum(2.0 * m["nL"] * m["nR"] + m["nR_all"] * m["U"] for m in stats)  # two searches per block on the CPU
    n_cells = sum(b.shape[0] for b in batches)
    return {
        "value": n_cells / (total / rate), "unit": "cells/s", "cores": cores, "kind": "port",
        "sample": (f"oracle exact FP64 brute-force kNN (OpenMP, {cores} threads): {ql.shape[0]} sampled query rows x "
                   f"full reference set for each of the 3 searches of the merge, {dt:.1f} s; extrapolated by pair "
                   f"evaluations ({rate:.3g}/s) to the whole job (kNN is >98% of the CPU time)"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    import torch
    torch.cuda.set_device(local_rank)
    import batchelor_amd as bx
    from batchelor_amd.dist import TorchExchange

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg, sizes, d, k, tree = WORKLOADS[args.workload]
    batches = synth_batches(cfg, sizes, d)
    n_cells = int(sum(sizes))

    eng = bx.MnnEngine(local_rank)
    exchange = "none"
    if world > 1:
        # production: RCCL inside the engine (in place, on its stream); if the communicator cannot be made, the
        # torch.distributed all-gather through the callback
        from batchelor_amd.dist import init_engine_rccl
        try:
            init_engine_rccl(eng)
            exchange = "rccl in engine"
        except Exception as exc:  # noqa: BLE001
            if rank == 0:
                print(f"bench.py: engine-owned RCCL unavailable ({exc}); using torch.distributed", file=sys.stderr)
            eng.set_shard(rank, world, TorchExchange(local_rank))
            exchange = "torch.distributed all_gather_into_tensor"
    eng.upload(batches)  # inputs resident in HBM from here on

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if tree is not None:
        from batchelor_amd.merge_tree import resolve_merge_order
        tree = resolve_merge_order(len(sizes), tree)
    for _ in range(args.warmup):
        eng.run(k=k, merge_tree=tree)
    eng.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    topk_ms, topk_launches = 0.0, 0
    for _ in range(args.steps):
        eng.run(k=k, merge_tree=tree)  # returns after the engine's stream has drained
        p = eng.profile()
        topk_ms += p["topk_ms"]
        topk_launches += p["topk_launches"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = eng.merge_stats()
    prof = eng.profile()
    fallbacks = prof["exact_fallbacks"]
    kern = KERNELS.get(prof["variant"], KERNELS[3])
    if rank == 0:
        flops = algorithmic_flops(stats, d) / world  # this rank's share of the query rows
        achieved = flops * args.steps / (topk_ms * 1e-3) / 1e12 if topk_ms > 0 else 0.0
        line = {
            "metric": "cells/sec corrected (reducedMNN engine, 50 PCs)",
            "value": n_cells * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": kern["dtype"],
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {len(sizes)} synthetic Gaussian batches x "
                                   f"{sizes[0] if len(set(sizes)) == 1 else str(min(sizes)) + '..' + str(max(sizes))} "
                                   f"cells x {d} PCs, k={k}, "
                                   f"merge.order={'1..' + str(len(sizes)) if tree is None else 'balanced tree'}, "
                                   "inputs resident in HBM",
                       "parallelism": "kNN query rows sharded over ranks, RCCL all-gather of neighbour lists"
                                      if world > 1 else "single GPU",
                       "mnn_pairs": [m["P"] for m in stats], "exact_fallback_queries": fallbacks},
            "roofline": {
                "bound": "mfma", "kernel": kern["kernel"], "achieved": achieved, "peak": kern["peak"],
                "unit": "TFLOP/s", "frac": achieved / kern["peak"],
                "traffic": measured_traffic(args.workload, prof["variant"]) if world == 1 else None,
                "launches_per_step": topk_launches / max(1, args.steps),
                "avg_launch_ms": topk_ms / max(1, topk_launches),
                "algorithmic_flops_per_step": flops,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(batches, stats, d, k)
            except Exception as exc:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {exc}"}
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
