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

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel knn_topk_f16: algorithmic
FLOPs / HIP-event time of its launches vs the dense fp16 MFMA peak), `streaming` (the HBM-bound rest of the merges
against the HBM peak), `value_host_to_host` (upload, run, download and pairs included) and, at N = 1, `cpu_baseline`
(two CPU statements of the dominant step timed on a bounded sample of the same workload on this box's host cores).
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


def measured_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/): FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this very command, corrected as MI355X_MICROARCH.md prescribes
    for gfx950 (FETCH_SIZE x 2).  PMC counters cannot be read from inside the timed process, so the number is the
    last committed measurement -- and it is only reported when it was taken on the very kernel (template arguments
    included) this run has just launched; otherwise None plus the reason."""
    rec = None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):  # the latest committed measurement
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_traffic_{workload}.json")))
            break
        except (OSError, ValueError):
            continue
    if rec is None:
        return None, f"no committed PMC measurement for {workload}"
    if rec.get("workload") != workload or not str(rec.get("kernel", "")).startswith(kernel):
        return None, f"committed PMC measurement is for {rec.get('kernel')!r}, this run launched {kernel!r}"
    return rec["bytes_per_launch"], None


def algorithmic_flops(stats, d):
    """F_merge = 2 d (nL nR + nR U): one shared distance block for both kNN directions plus the
    tricube search of every right cell against the U MNN-involved right cells."""
    return sum(2.0 * d * (m["nL"] * m["nR"] + m["nR_all"] * m["U"]) for m in stats)


def streaming_bytes(stats, d, k):
    """B_merge = 8 d [(nL + nR)(1 + 2 + 1) + 2 nR E + 4 P + 2 nR] + 12 nR k -- variances before and
    after, the centring pass (read + write), orthogonalisation of the right batch against E earlier batch vectors, the
    two gathers of the two averaging passes, the tricube apply (read + write) and its index / distance lists."""
    tot = 0.0
    for e, m in enumerate(stats):
        n_l, n_r = m["nL_all"], m["nR_all"]
        tot += 8.0 * d * ((n_l + n_r) * 4 + 2 * n_r * e + 4 * m["P"] + 2 * n_r) + 12.0 * n_r * k
    return tot


def host_fp64_peak_gflops():
    """The host's FP64 peak as cores x flops per cycle x clock, from what /proc and /sys say (stated, not measured): physical
    cores = distinct (physical id, core id) pairs; 32 flops per cycle and core with AVX-512 (two 512-bit FMA pipes), else 16;
    clock = cpuinfo_max_freq, else the largest current 'cpu MHz'."""
    try:
        cores, flags, mhz = set(), "", 0.0
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                    cores.add((phys, core))
                elif ln.startswith("flags") and not flags:
                    flags = ln
                elif ln.startswith("cpu MHz"):
                    mhz = max(mhz, float(ln.split(":")[1]))
        try:
            with open("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq") as f:
                mhz = float(f.read()) / 1e3
        except OSError:
            pass
        ncores = len(cores) or (os.cpu_count() or 1)
        per = 32 if " avx512f" in flags else 16
        return ncores * per * mhz / 1e3, f"{ncores} physical cores x {per} FP64 flops/cycle x {mhz / 1e3:.2f} GHz"
    except Exception as exc:  # noqa: BLE001
        return None, f"unknown ({exc})"


def cpu_baselines(batches, stats, d, k):
    """The reference R/Rcpp path cannot run on this box (no R), so two CPU statements of its dominant step -- the
    exact kNN searches, > 98 % of the CPU time -- are timed on a bounded sample of the same workload and extrapolated by
    pair evaluations to the whole job (oracle/cpu_baselines.py):
      A  one thread, pruned exact search in the manner of KmknnParam() / SerialParam(), fastMNN()'s defaults;
      B  all host threads, blocked brute force on the host BLAS with a fused running-threshold filter;
      C  all host cores, the oracle's OpenMP brute force (no BLAS);
      T  all physical cores, cache-blocked brute force with a hand-written AVX2 + FMA micro-kernel and the filter fused in.
    Returns (main, variants): main = the faster one in the contract's cpu_baseline form."""
    from oracle import cpu_baselines as cb
    hw_threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # (round 6: the GPU boxes of this pool show 256 hardware threads and grant the container 16 CPUs' worth of time
    # (cgroup cpu.max): every "all cores" baseline of rounds 1-5 ran 256 threads on that quota.  The baselines now start as
    # many threads as the quota and the physical cores allow, and say so.)
    quota = cb.cpu_quota()
    cores = cb.usable_cores() if quota else hw_threads
    quota_note = (f"; the container's cgroup grants {quota:g} CPUs of the host's {hw_threads} hardware threads" if quota else "")
    L, R = batches[0], batches[1]
    rng = np.random.default_rng(0)
    total_pairs = sum(2.0 * m["nL"] * m["nR"] + m["nR_all"] * m["U"] for m in stats)  # two searches per block on a CPU
    n_cells = sum(b.shape[0] for b in batches)
    out = {}
    # A: build on one batch (the k-means cost scales with the cells, like the search), a few hundred queries
    t0 = time.perf_counter()
    nq_a = 256
    qa = R[rng.choice(R.shape[0], nq_a, replace=False)]
    _, _, st = cb.kmknn_knn(L, qa, k, iters=3)
    rate_a = nq_a * L.shape[0] / st["query_s"]
    build_total = st["build_s"] * sum((m["nL"] + m["nR"]) for m in stats) / L.shape[0]  # an index per side per merge
    out["A"] = {"value": n_cells / (total_pairs / rate_a + build_total), "unit": "cells/s", "cores": 1, "kind": "port",
                "sample": (f"KMKNN-style pruned exact search (oracle/kmknn_baseline.c), 1 thread: index over "
                           f"{L.shape[0]} cells in {st['build_s']:.1f} s, {nq_a} queries in {st['query_s']:.2f} s, "
                           f"{100 * st['visited']:.1f}% of the reference visited per query ({rate_a:.3g} pair "
                           f"evaluations/s); extrapolated by pair evaluations to the whole job, "
                           f"{time.perf_counter() - t0:.0f} s of CPU time spent")}
    # B: BLAS brute force over worker threads.  One WHOLE block of the job (every right cell against the first batch) when a
    # short trial says it fits the budget of ~25 s, else as many queries as do
    flop_per_pair = 2.0 * d
    t0 = time.perf_counter()
    nq_b = min(R.shape[0], 8192)
    _, _, info_b = cb.blas_knn(L, R[:nq_b], k, workers=cores)
    dt0 = time.perf_counter() - t0
    nq_b = int(min(R.shape[0], max(8192, 8192 * 25.0 / max(dt0, 1e-3))))
    t0 = time.perf_counter()
    cb.blas_knn(L, R[:nq_b], k, workers=cores)
    dt = time.perf_counter() - t0
    rate_b = nq_b * L.shape[0] / dt
    whole_b = "one whole block of the job" if nq_b == R.shape[0] else "a sample of the block"
    peak_gf, peak_how = host_fp64_peak_gflops()
    if quota and peak_gf:  # the peak this process can reach: its quota's share of the physical cores
        phys = cb.physical_cores()
        peak_gf = peak_gf * min(1.0, quota / max(phys, 1))
        peak_how += f", of which the cgroup's quota of {quota:g} CPUs can reach {peak_gf:.0f} GFLOP/s"
    out["B"] = {"value": n_cells / (total_pairs / rate_b), "unit": "cells/s", "cores": info_b["workers"], "kind": "port",
                "gflops": rate_b * flop_per_pair / 1e9,
                "host_fp64_peak_gflops": peak_gf, "host_fp64_peak_how": peak_how,
                "frac_of_host_fp64_peak": (rate_b * flop_per_pair / 1e9 / peak_gf) if peak_gf else None,
                "sample": (f"blocked brute force on the host BLAS ({info_b['blas']}; FP64 DGEMM tiles of 256 queries x 4096 "
                           f"reference cells with a fused running-threshold filter + exact "
                           f"re-evaluation of the kept), {info_b['workers']} worker threads x {info_b['blas_threads_per_worker']} "
                           f"BLAS thread(s){quota_note}: {whole_b}, {nq_b} queries x {L.shape[0]} reference cells in "
                           f"{dt:.1f} s ({rate_b:.3g} pair evaluations/s = {rate_b * flop_per_pair / 1e9:.0f} GFLOP/s); scaled by "
                           f"pair evaluations to the whole job")}
    # C: the oracle's own OpenMP brute force (FP64, no BLAS), all cores -- round 1's baseline, kept for continuity
    from oracle import fastmnn_oracle as orc
    t0 = time.perf_counter()
    orc.query_knn(L, R[:4096], k, nthreads=cores)
    dt0 = time.perf_counter() - t0
    nq_c = int(min(R.shape[0], max(4096, 4096 * 15.0 / max(dt0, 1e-3))))
    t0 = time.perf_counter()
    orc.query_knn(L, R[:nq_c], k, nthreads=cores)
    dt = time.perf_counter() - t0
    rate_c = nq_c * L.shape[0] / dt
    whole_c = "one whole block of the job" if nq_c == R.shape[0] else "a sample of the block"
    out["C"] = {"value": n_cells / (total_pairs / rate_c), "unit": "cells/s", "cores": cores, "kind": "port",
                "gflops": rate_c * flop_per_pair / 1e9,
                "sample": (f"oracle exact FP64 brute force (oracle/mnn_oracle.c, OpenMP, {cores} threads{quota_note}): {whole_c}, {nq_c} "
                           f"queries x {L.shape[0]} reference cells in {dt:.1f} s ({rate_c:.3g} pair evaluations/s = "
                           f"{rate_c * flop_per_pair / 1e9:.0f} GFLOP/s); scaled by pair evaluations to the whole job")}
    # T: the same search the way a CPU wants it (oracle/tiled_knn_baseline.c): packed operands, a register-blocked AVX2 + FMA
    # micro-kernel, the reference block in L2, the filter on the tile in registers; one thread per physical core; whole blocks
    # of the job (every right cell against a batch) until ~10 s are spent
    try:
        nt = cb.usable_cores()
        cb.tiled_knn(L[:4096], R[:512], k, nthreads=nt)
        t0 = time.perf_counter()
        pairs_t, nblk = 0.0, 0
        while time.perf_counter() - t0 < 10.0 and nblk < 64:
            cb.tiled_knn(L, R, k, nthreads=nt)
            pairs_t += float(L.shape[0]) * R.shape[0]
            nblk += 1
        dt = time.perf_counter() - t0
        rate_t = pairs_t / dt
        out["T"] = {"value": n_cells / (total_pairs / rate_t), "unit": "cells/s", "cores": nt, "kind": "port",
                    "gflops": rate_t * flop_per_pair / 1e9, "host_fp64_peak_gflops": peak_gf,
                    "frac_of_host_fp64_peak": (rate_t * flop_per_pair / 1e9 / peak_gf) if peak_gf else None,
                    "sample": (f"cache-blocked FP64 brute force with a 4 x 8 AVX2 + FMA micro-kernel and the threshold filter fused "
                               f"into it (oracle/tiled_knn_baseline.c, OpenMP, {nt} threads{quota_note}): "
                               f"{nblk} whole block(s) of the job, {R.shape[0]} queries x {L.shape[0]} reference cells each, in {dt:.1f} s "
                               f"({rate_t:.3g} pair evaluations/s = {rate_t * flop_per_pair / 1e9:.0f} GFLOP/s); scaled by pair "
                               f"evaluations to the whole job")}
    except Exception as exc:  # noqa: BLE001
        out["T"] = {"value": 0.0, "unit": "cells/s", "cores": 0, "kind": "port", "sample": f"failed: {exc}"}
    main = dict(max(out.values(), key=lambda r: r["value"]))
    return main, out


def cpu_baseline_var_adj(batches, d, sigma, asv_pairs_per_step):
    """adjust_shift_variance on the host cores: the oracle's restatement of src/adjust_shift_variance.cpp:51-161 (OpenMP over
    the cells, which the reference's loop treats independently) on a bounded sample -- 8 cells per host thread of one batch against
    ~200 000 reference cells + its own batch -- scaled by (cell, restricted cell) pairs to the step's calls."""
    from oracle import fastmnn_oracle as orc
    from oracle import cpu_baselines as cb
    cores = cb.usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # (the oracle's loop takes OpenMP's default: set before its first parallel region)
    order = np.argsort([-b.shape[0] for b in batches])
    ref = batches[order[0]][:200000]
    own = batches[order[1]][:60000]
    rng = np.random.default_rng(3)
    cells = np.sort(rng.choice(own.shape[0], min(8 * cores, own.shape[0]), replace=False)).astype(np.int32)
    vect = rng.standard_normal((own.shape[0], d)) * 0.2
    r1, r2 = np.arange(ref.shape[0]), np.arange(own.shape[0])
    t0 = time.perf_counter()
    orc.adjust_shift_variance(ref.T, own.T, vect, sigma, r1, r2, cells=cells)
    dt = time.perf_counter() - t0
    rate = cells.size * float(ref.shape[0] + own.shape[0]) / dt
    return {"seconds_per_step": asv_pairs_per_step / rate, "pairs_per_s": rate, "cores": cores, "kind": "port",
            "sample": (f"oracle/mnn_oracle.c orc_adjust_shift_variance_cells (OpenMP, {cores} threads): {cells.size} cells x "
                       f"({ref.shape[0]} + {own.shape[0]}) restricted cells, {d} dims, sigma {sigma}, in {dt:.1f} s = {rate:.3g} "
                       f"(cell, restricted cell) pairs/s; scaled by pairs to the step's {asv_pairs_per_step:.3g}")}


def run_config4(args):
    """BASELINE.json configs[3]: fastMNN end to end -- cosineNorm + multiBatchPCA(d = 50) over 20 000 genes + reducedMNN,
    4 batches x --cells cells (200 000 = the configuration as named: 128 GB of FP64 input).  The input never exists in one
    piece: blocks of 2 048 cells are generated by a pool of host threads (50 shared expression programmes with
    log-normal-ish loadings + noise + a per-batch offset) and handed to the boundary's blocked ingest
    (bmx_pca_begin_batch / bmx_pca_add_block: pinned double-buffered upload, cosine norms on the fly), so host memory holds
    a few blocks and HBM holds the batches.  PCA (run to a relative Ritz residual of --pca-tol), projection and merge engine
    are timed separately; `ingest_ms` is the wall time of the add_block calls (generation overlaps it where it can and is
    reported beside it)."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    torch.cuda.set_device(0)
    import batchelor_amd as bx
    G, d, nb, n = 20000, 50, 4, args.cells
    blk = 2048
    load = np.abs(np.random.Generator(np.random.PCG64(20250314 + 4000)).standard_normal((G, d))) \
        * (1.0 / np.sqrt(1.0 + np.arange(d) / 5.0))

    def make_block(b, c0):
        return make_block_n(b, c0, min(blk, n - c0))

    def make_block_n(b, c0, m):
        rng = np.random.Generator(np.random.PCG64([20250314 + 4000, b, c0]))
        z = rng.standard_normal((d, m))
        x = np.empty((G, m), order="F")
        x[:] = load @ z
        x += (rng.random((G, m)) - 0.5) * (0.5 * 3.4641)
        x += 4.0 + 0.3 * b
        return x

    # accuracy first, at a size the dense CPU decomposition takes seconds for (4 x 2 000 cells of the same generator):
    # rotation, corrected coordinates and MNN pairs of the whole device pipeline against oracle/pca_oracle.py
    from oracle import pca_oracle
    ns = 2000
    small = [np.hstack([make_block_n(b, c0, min(blk, ns - c0)) for c0 in range(0, ns, blk)]) for b in range(nb)]
    dev = bx.fastMNN(*small, d=d)
    ref, meta = pca_oracle.fast_mnn(*small, d=d, pca_method="gram")
    sgn = np.sign((dev.rotation * meta["rotation"]).sum(axis=0))
    rot_err = float(np.abs(dev.rotation * sgn[None, :] - meta["rotation"]).max() / np.abs(meta["rotation"]).max())
    cor_err = float(np.abs(dev.corrected * sgn[None, :] - ref.corrected).max() / np.abs(ref.corrected).max())
    pairs_equal = all(np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1])
                      for a, b_ in zip(dev.merge_info.pairs, ref.merge_info.pairs))
    accuracy = {"cells_per_batch": ns, "rotation_max_rel_err": rot_err, "corrected_max_rel_err": cor_err,
                "mnn_pairs_bit_exact": bool(pairs_equal), "checker": "oracle/pca_oracle.py (dense Gram decomposition, numpy)"}
    assert rot_err < 1e-5 and cor_err < 1e-5 and pairs_equal, accuracy
    del small, dev, ref, meta

    times = {"generation_wait_ms": 0.0, "ingest_ms": 0.0}
    # the generators run numpy with the BLAS pinned to one thread each (24 multi-threaded matmuls at once starved the
    # library's copy threads in round 3: 8.6 GB/s of ingest at 200 000 cells against 19.8 at 25 000)
    workers = max(2, min(args.gen_threads, (os.cpu_count() or 4) // 2))
    try:
        from threadpoolctl import threadpool_limits
        blas_limit = threadpool_limits(limits=1, user_api="blas")
    except Exception:
        blas_limit = None
    jobs = [(b, c0) for b in range(nb) for c0 in range(0, n, blk)]
    pca = bx.DevicePCA(G, 0)
    t_all = time.perf_counter()
    with ThreadPoolExecutor(workers) as pool:
        ahead = 2 * workers
        futs = [pool.submit(make_block, *jobs[i]) for i in range(min(ahead, len(jobs)))]
        nxt = len(futs)
        cur_b = -1
        for i, (b, c0) in enumerate(jobs):
            t0 = time.perf_counter()
            x = futs[i].result()
            futs[i] = None
            times["generation_wait_ms"] += 1e3 * (time.perf_counter() - t0)
            if nxt < len(jobs):
                futs.append(pool.submit(make_block, *jobs[nxt]))
                nxt += 1
            t0 = time.perf_counter()
            if b != cur_b:
                pca.begin_batch(n, weight=1.0, cos_norm=True)
                cur_b = b
            pca.add_block(x)
            times["ingest_ms"] += 1e3 * (time.perf_counter() - t0)
            del x
    torch.cuda.synchronize()
    if blas_limit is not None:
        blas_limit.restore_original_limits()
    times["generation_and_ingest_wall_ms"] = 1e3 * (time.perf_counter() - t_all)
    t0 = time.perf_counter()
    fit = pca.fit(d=d, tol=args.pca_tol, max_iters=500)
    times["pca_fit_ms"] = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    pcs = [pca.project(b) for b in range(nb)]
    times["projection_ms"] = 1e3 * (time.perf_counter() - t0)
    pca.close()
    t0 = time.perf_counter()
    out = bx.reducedMNN(*pcs, k=20)
    times["merge_engine_ms"] = 1e3 * (time.perf_counter() - t0)
    total = times["ingest_ms"] + times["pca_fit_ms"] + times["projection_ms"] + times["merge_engine_ms"]
    in_bytes = 8.0 * G * nb * n
    flops_pca = fit["iters_used"] * 2 * 2.0 * G * 64 * nb * n  # two 64-wide products per batch and application of the operator
    line = {"metric": "cells/sec corrected (fastMNN end to end: cosineNorm + multiBatchPCA + reducedMNN)",
            "value": nb * n / (total * 1e-3), "unit": "cells/s", "n_gpus": 1, "steps": 1, "warmup": 0,
            "ms_per_step": total, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64 MFMA (PCA) + fp16 MFMA candidate pass / FP64 exact re-rank (merge engine)", "data": "synthetic",
            "config": {"workload": f"config4: {nb} batches x {n} cells x {G} genes -> {d} PCs, host blocks of {blk} cells in "
                                   f"({in_bytes / 1e9:.0f} GB), host result out" + ("" if n == 200000 else
                                   "; the configuration as named has 200000 cells per batch"),
                       "generator_threads": workers, "mnn_pairs": [int(p[0].size) for p in out.merge_info.pairs]},
            "stages_ms": times, "accuracy": accuracy,
            "ingest_GBps": in_bytes / 1e9 / (times["ingest_ms"] * 1e-3),
            "pca": {"operator_applications": fit["iters_used"], "relative_ritz_residual": fit["residual"],
                    "tolerance": args.pca_tol, "algorithmic_flops": flops_pca,
                    "achieved_TFLOPs": flops_pca / (times["pca_fit_ms"] * 1e-3) / 1e12,
                    "singular_values_head": [float(v) for v in fit["d"][:3]]}}
    print(json.dumps(line), flush=True)


def run_sgk(args):
    """smooth_gaussian_kernel (K8) at n = 1e5 cells, U = 1e4 MNN cells, 50 dimensions, through the .Call-level entry
    (host matrices in and out); a sample of cells is checked against the dense specification of
    tests/testthat/test-mnn-correct.R:36-65 computed by numpy."""
    import torch
    torch.cuda.set_device(0)
    from batchelor_amd import natives as nat
    n, U, gd = 100000, 10000, 50
    rng = np.random.Generator(np.random.PCG64(20250314 + 8000))
    mat = rng.standard_normal((gd, n)) / np.sqrt(1.0 + np.arange(gd) / 5.0)[:, None]
    index = np.sort(rng.choice(n, U, replace=False))
    averaged = rng.standard_normal((gd, U)) * 0.1
    s2 = 1.0
    nat.smooth_gaussian_kernel(averaged, index, mat, s2)
    t0 = time.perf_counter()
    out = nat.smooth_gaussian_kernel(averaged, index, mat, s2)
    dt = time.perf_counter() - t0
    import ctypes
    from batchelor_amd import _lib as _bl
    _bl.lib().bmx_last_native_kernel_ms.restype = ctypes.c_double
    kern_ms = float(_bl.lib().bmx_last_native_kernel_ms())
    cells = rng.choice(n, 64, replace=False)
    m = mat[:, index]
    d_mm = ((m[:, :, None] - m[:, None, :]) ** 2).sum(axis=0) if U <= 2000 else None
    # dense REF on the sample (chunked over MNN cells to bound memory)
    dens = np.zeros(U)
    for i0 in range(0, U, 500):
        dd = ((m[:, i0:i0 + 500, None] - m[:, None, :]) ** 2).sum(axis=0)
        dens[i0:i0 + 500] = np.log(np.exp(-dd / s2).sum(axis=1))
    dc = ((m[:, :, None] - mat[:, None, cells]) ** 2).sum(axis=0)          # U x 64
    lw = -dc / s2 - dens[:, None]
    w = np.exp(lw - lw.max(axis=0))
    ref = (averaged @ w) / w.sum(axis=0)
    err = float(np.abs(out[:, cells] - ref).max() / np.abs(ref).max())
    flops = 2.0 * gd * U * (n + U) + 2.0 * gd * U * n
    cpu = None
    if not args.no_cpu_baseline:
        # the reference's loop (src/smooth_gaussian_kernel.cpp:32-99: serial over the MNN cells, no threads) as the oracle
        # restates it, on a bounded sample -- 400 of the MNN cells against themselves + 20 000 other cells -- scaled by
        # (MNN cell, cell) pairs to the whole call
        from oracle import fastmnn_oracle as orc
        Us, ns_ = 400, 20000
        sel = np.concatenate([index[:Us], np.setdiff1d(np.arange(n), index[:Us])[:ns_]])
        t1 = time.perf_counter()
        orc.smooth_gaussian_kernel(averaged[:, :Us], np.arange(Us), mat[:, sel], s2)
        dtc = time.perf_counter() - t1
        rate = Us * float(sel.size) / dtc
        cpu = {"value": n / (U * float(n) / rate), "unit": "cells/s", "cores": 1, "kind": "port",
               "sample": (f"oracle/mnn_oracle.c orc_smooth_gaussian_kernel (the reference's serial loop, 1 thread): {Us} MNN cells x "
                          f"{sel.size} cells in {dtc:.1f} s = {rate:.3g} (MNN cell, cell) pairs/s; scaled by pairs to U x n = "
                          f"{U * float(n):.3g}")}
    if kern_ms != kern_ms:  # (NaN: the call's events could not be read)
        kern_ms = 0.0
    print(json.dumps({"metric": "cells/sec smoothed (smooth_gaussian_kernel, .Call level, host in / host out)",
                      "value": n / dt, "unit": "cells/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": 1e3 * dt,
                      "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 MFMA",
                      "data": "synthetic", "config": {"workload": f"sgk: n={n} cells, U={U} MNN cells, {gd} dims, sigma2={s2}",
                                                      "max_rel_err_vs_dense_spec_on_64_cells": err},
                      # src/smooth_gaussian_kernel.cpp:36-98 in GEMM form: the distance products of every (MNN cell, cell)
                      # and (MNN cell, MNN cell) pair + the weighted sum of the averaged vectors, 2 flops per multiply-add;
                      # peak: the FP64 matrix rate of the MI355X data sheet (not in the in-container guide)
                      "roofline": {"bound": "mfma", "kernel": "sgk_flash<true> (+ sgk_flash<false>, row_norms2)",
                                   "achieved": flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None, "peak": 78.6,
                                   "unit": "TFLOP/s", "frac": flops / (kern_ms * 1e-3) / 1e12 / 78.6 if kern_ms > 0 else None,
                                   "kernel_ms": kern_ms, "traffic": None,
                                   "peak_note": "FP64 matrix peak from AMD's MI355X data sheet; not in the in-container guide"},
                      "cpu_baseline": cpu, "host_cores": os.cpu_count(),
                      "algorithmic_flops": flops}), flush=True)
    assert err < 1e-9, err


def measure_exchange_call_overhead(batches, k, tree, run_kw, steps, warmup):
    """What ONE exchange costs beside its bytes, measured: an engine with its own RCCL communicator of ONE rank (the only
    world this pool offers) runs the job with and without the testing hook "exchange_always" -- with it every list goes
    through ncclAllGather on the engine's stream (in place, one rank: a launch and its latency, no link traffic) and the
    sharded forms of the tricube apply and of the first averaging run with their copy kernels.  (ms per step without, with,
    calls per step) -> the per-call term of the exchange model."""
    import socket
    import torch
    import torch.distributed as dist
    import batchelor_amd as bx
    from batchelor_amd import _lib as _bl
    from batchelor_amd.dist import init_engine_rccl
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    eng = bx.MnnEngine(0)
    try:
        init_engine_rccl(eng)
        eng.upload(batches)
        out = {}
        for rep in range(2):
            for flag in (0, 1):
                _bl.dev_set("exchange_always", flag)
                for _ in range(warmup):
                    eng.run(k=k, merge_tree=tree, **run_kw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    eng.run(k=k, merge_tree=tree, **run_kw)
                torch.cuda.synchronize()
                out.setdefault(flag, []).append(1e3 * (time.perf_counter() - t0) / steps)
                if flag:
                    calls = eng.exchange_stats()["calls"]
        return min(out[0]), min(out[1]), calls
    finally:
        _bl.dev_set("exchange_always", 0)
        eng.close()
        dist.destroy_process_group()


def run_emulation(args):
    """One rank's share of an N-rank run, measured on ONE GPU: `bmx_engine_emulate` makes the engine rank r
    of N -- its slice of every search's query rows, every replicated kernel in full -- with the other ranks' slices of each
    exchange replayed from a recorded single-rank run.  The collectives themselves are not run: their count and bytes are
    reported with a stated model of their time beside the measured per-rank step.  One JSON line per world size."""
    import torch
    import batchelor_amd as bx
    torch.cuda.set_device(0)
    cfg, sizes, d, k, tree = WORKLOADS[args.workload]
    batches = synth_batches(cfg, sizes, d)
    n_cells = int(sum(sizes))
    if tree is not None:
        from batchelor_amd.merge_tree import resolve_merge_order
        tree = resolve_merge_order(len(sizes), tree)
    run_kw = {"var_adj": True, "sigma": args.sigma} if args.var_adj else {}
    eng = bx.MnnEngine(0)
    eng.upload(batches)

    def timed(steps, warmup):
        for _ in range(warmup):
            eng.run(k=k, merge_tree=tree, **run_kw)
        eng.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = None
        for _ in range(steps):
            eng.run(k=k, merge_tree=tree, **run_kw)
            p = eng.profile_detail()
            acc = dict(p) if acc is None else {key: (acc[key] + v if isinstance(v, (int, float)) else v) for key, v in p.items()}
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        eng.set_profiling(False)
        cand = (acc["f16_ms"] + acc["bf16_ms"] + acc["sample_ms"]) / steps
        return ms, cand, acc["streaming_ms"] / steps

    t1, cand1, stream1 = timed(args.steps, args.warmup)
    base = eng.download(with_pairs=True)
    per_call_us, per_call_note = 15.0, "assumed"
    if args.measure_exchange:
        try:
            ms0, ms1, calls1 = measure_exchange_call_overhead(batches, k, tree, run_kw, args.steps, args.warmup)
            per_call_us = max(0.0, 1e3 * (ms1 - ms0) / max(calls1, 1))
            per_call_note = (f"measured: one-rank ncclAllGather on the engine's stream for every exchange, {ms0:.2f} -> {ms1:.2f} ms "
                             f"per step over {calls1} calls (the sharded forms' copy kernels included)")
        except Exception as exc:  # noqa: BLE001
            per_call_note = f"assumed (measurement failed: {exc})"
    eng.emulate(1)
    eng.run(k=k, merge_tree=tree, **run_kw)   # the recorded run
    # xGMI: 7 links x ~153 GB/s per GPU (task statement); a ring all-gather moves (N-1)/N of the gathered bytes through every
    # rank at the rate of its slowest hop; modelled at 60 % of one direction of the links a rank can drive + 15 us per launch
    for world in [int(x) for x in str(args.emulate_world).split(",")]:
        ranks = list(range(world)) if args.emulate_ranks == "all" else [int(x) for x in args.emulate_ranks.split(",") if int(x) < world]
        per_rank, detail = [], []
        xst = None
        for r in ranks:
            eng.emulate(2, r, world)
            # (the faster of two measurements: one host-side hiccup in five steps -- seen twice on the last rank of a long
            # series, not reproducible on that rank alone -- would otherwise become the "slowest rank")
            ms, cand, stream = min(timed(args.steps, args.warmup), timed(args.steps, 0))
            per_rank.append(ms)
            detail.append({"rank": r, "ms_per_step": ms, "candidate_pass_ms": cand, "streaming_ms": stream})
            xst = eng.exchange_stats()
            if r == ranks[0]:  # the emulated rank ends with the recorded run's result, bit for bit
                got = eng.download(with_pairs=True)
                assert np.array_equal(got.corrected, base.corrected)
                for (a0, a1), (b0, b1) in zip(got.merge_info.pairs, base.merge_info.pairs):
                    assert np.array_equal(a0, b0) and np.array_equal(a1, b1)
        links = min(world - 1, 7)
        model_ms = 1e3 * (xst["bytes"] * (world - 1) / world) / (0.6 * 153e9 * links) + 1e-3 * per_call_us * xst["calls"]
        worst = max(per_rank)
        print(json.dumps({
            "metric": "one rank's ms per step, emulated on one GPU (rank r of N: its share of every search, every replicated kernel)",
            "workload": args.workload, "n_gpus_emulated": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step_one_gpu": t1, "candidate_pass_ms_one_gpu": cand1, "streaming_ms_one_gpu": stream1,
            "per_rank_ms_per_step": per_rank, "max_rank_ms_per_step": worst, "per_rank": detail,
            "ideal_ms_per_step": t1 / world, "sharding_efficiency": (t1 / world) / worst,
            "exchange_calls_per_step": xst["calls"], "exchange_bytes_per_step": xst["bytes"],
            "exchange_model_ms_per_step": model_ms,
            "exchange_model": f"bytes (N-1)/N over min(N-1, 7) xGMI links at 60 % of 153 GB/s + {per_call_us:.1f} us per call ({per_call_note}), serial",
            "projected_speedup_compute_only": t1 / worst, "projected_speedup_with_modelled_exchange": t1 / (worst + model_ms),
            "projected_cells_per_s": n_cells / ((worst + model_ms) * 1e-3),
            "results_identical_to_one_gpu": True}), flush=True)
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS) + ["config4", "sgk"])
    ap.add_argument("--cells", type=int, default=25000, help="config4: cells per batch (200000 = the configuration as named)")
    ap.add_argument("--pca-tol", type=float, default=1e-9, help="config4: relative Ritz residual the PCA iterates to")
    ap.add_argument("--gen-threads", type=int, default=16, help="config4: host threads generating the input blocks")
    ap.add_argument("--var-adj", action="store_true", help="config5: mnnCorrect-style variance adjustment in the merges")
    ap.add_argument("--sigma", type=float, default=1.0, help="--var-adj: the bandwidth handed to adjust_shift_variance "
                    "(1.0 relative to the synthetic spectrum; 0.1 is mnnCorrect's default, R/mnnCorrect.R:125-130)")
    ap.add_argument("--dev", action="append", default=[], metavar="KNOB=VALUE",
                    help="developer A/B runs: a testing hook of the library (bmx_dev_set), e.g. --dev asv_cap=0")
    ap.add_argument("--emulate-world", default=None, metavar="N[,N...]",
                    help="measure one rank's share of an N-rank run on this one GPU (bmx_engine_emulate); e.g. 2,4,8")
    ap.add_argument("--emulate-ranks", default="all", help="--emulate-world: which ranks to measure (all, or e.g. 0,3,7)")
    ap.add_argument("--measure-exchange", action="store_true",
                    help="--emulate-world: measure the per-call term of the exchange model with a one-rank RCCL communicator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true")
    args = ap.parse_args()

    if args.dev:
        from batchelor_amd import _lib as _bl
        for kv in args.dev:
            name, val = kv.split("=")
            _bl.dev_set(name, int(val))
    if args.emulate_world:
        return run_emulation(args)
    if args.workload == "config4":
        return run_config4(args)
    if args.workload == "sgk":
        return run_sgk(args)
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
    run_kw = {"var_adj": True, "sigma": args.sigma} if args.var_adj else {}
    for _ in range(args.warmup):
        eng.run(k=k, merge_tree=tree, **run_kw)
    eng.set_profiling(True)
    if args.var_adj:
        from batchelor_amd import _lib as _bl
        if not hasattr(_bl.lib(), "bmx_engine_profile_var_adj"):  # (BMX_LIB = a build of an earlier round, A/B runs)
            args.var_adj_old_lib = True
        else:
            _bl.dev_get("asv_tally_reset")
    barrier()
    t0 = time.perf_counter()
    acc = None
    asv = {"asv_ms": 0.0, "asv_launches": 0, "asv_pairs": 0.0}
    for _ in range(args.steps):
        eng.run(k=k, merge_tree=tree, **run_kw)  # returns after the engine's stream has drained
        p = eng.profile_detail()
        if args.var_adj and not getattr(args, "var_adj_old_lib", False):
            for key, v in eng.profile_var_adj().items():
                asv[key] += v
        if acc is None:
            acc = dict(p)
        else:
            for key, v in p.items():
                if isinstance(v, (int, float)):
                    acc[key] += v
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = eng.merge_stats()
    variant = eng.profile()["variant"]
    kern = KERNELS.get(variant, KERNELS[3])
    dom = "f16" if variant == 3 else "bf16"
    dom_ms, dom_launches = acc[dom + "_ms"], acc[dom + "_launches"]
    xst = eng.exchange_stats() if world > 1 else {"calls": 0, "bytes": 0}

    # host to host: column-major host inputs -> host outputs, upload, run, download and pairs included
    h2h = None
    if world == 1 and not args.no_host_to_host:
        eng.set_profiling(False)
        reps = []
        fbatches = [np.asfortranarray(b) for b in batches]  # what the boundary is handed: column-major matrices
        stages = []
        for _ in range(3):
            t1 = time.perf_counter()
            e2 = bx.MnnEngine(local_rank)
            ta = time.perf_counter()
            e2.upload(fbatches)
            tb = time.perf_counter()
            e2.run(k=k, merge_tree=tree, **run_kw)
            tc = time.perf_counter()
            res = e2.download(with_pairs=True, c_order=False)
            td = time.perf_counter()
            e2.close()
            reps.append(time.perf_counter() - t1)
            stages.append({"create_ms": 1e3 * (ta - t1), "upload_ms": 1e3 * (tb - ta), "run_ms": 1e3 * (tc - tb),
                           "download_ms": 1e3 * (td - tc), "close_ms": 1e3 * (reps[-1] - (td - t1))})
            del res
        # ... and the same through the ONE call the .Call shim makes (bmx_fast_mnn + bmx_engine_pairs_into): the library
        # pulls the batches itself, the upload of batches 3.. hides behind the first merges
        from batchelor_amd.reduced_mnn import fast_mnn_one_shot
        one = []
        for _ in range(3):
            t1 = time.perf_counter()
            res = fast_mnn_one_shot(fbatches, k=k, merge_tree=tree, c_order=False, **run_kw)
            one.append(time.perf_counter() - t1)
            del res
        h2h_stages = stages[int(np.argmin(reps))]
        h2h_stages["staged_calls_total_ms"] = 1e3 * min(reps)
        h2h_stages["one_shot_call_ms"] = 1e3 * min(one)
        h2h = min(min(reps), min(one))

    if rank == 0:
        flops = algorithmic_flops(stats, d) / world  # this rank's share of the query rows
        achieved = flops * args.steps / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic, why = measured_traffic(args.workload, acc["kernel"]) if world == 1 else (None, "multi-GPU run")
        sbytes = streaming_bytes(stats, d, k)
        stream_s = acc["streaming_ms"] * 1e-3 / args.steps
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
                       "parallelism": f"kNN query rows sharded over ranks, all-gather of neighbour lists: {exchange}"
                                      if world > 1 else "single GPU",
                       "mnn_pairs": [m["P"] for m in stats], "exact_fallback_queries": acc["exact_fallbacks"],
                       "second_tier_queries": acc["tier2_queries"]},
            "roofline": {
                "bound": "mfma", "kernel": acc["kernel"], "achieved": achieved, "peak": kern["peak"],
                "unit": "TFLOP/s", "frac": achieved / kern["peak"],
                "traffic": traffic, "traffic_note": why,
                "launches_per_step": dom_launches / max(1, args.steps),
                "avg_launch_ms": dom_ms / max(1, dom_launches),
                "algorithmic_flops_per_step": flops,
                "other_candidate_passes_ms_per_step": {"sample": acc["sample_ms"] / args.steps,
                                                       "second_tier": (acc["bf16_ms"] if dom == "f16" else 0.0) / args.steps},
            },
            # the HBM-bound rest of a merge (SURVEY.md 8d, K3-K6 + the apply half of K7): algorithmic bytes over the
            # HIP-event time of the merges' streaming sections, against the 8 TB/s HBM peak
            "streaming": {"bound": "hbm", "algorithmic_bytes_per_step": sbytes, "ms_per_step": 1e3 * stream_s,
                          "achieved": sbytes / stream_s / 1e9 if stream_s > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                          "frac": sbytes / stream_s / 8e12 if stream_s > 0 else 0.0},
            "per_rank": {"candidate_pass_ms_per_step": (dom_ms + acc["sample_ms"] + acc["bf16_ms" if dom == "f16" else "f16_ms"])
                                                       / args.steps,
                         "streaming_ms_per_step": 1e3 * stream_s,
                         "exchange_calls_per_step": xst["calls"], "exchange_bytes_per_step": xst["bytes"]},
        }
        if args.var_adj and asv["asv_ms"] > 0:
            # configs[4] "with adjust_shift_variance on": the step is that kernel.  Algorithmic work per (cell, restricted cell)
            # pair: the two inner products x_c . x_o and g_c . x_o (src/adjust_shift_variance.cpp:9-27, :88-89 in GEMM form) =
            # 4 d flops.  Peak: the FP64 matrix rate of the MI355X data sheet, 78.6 TFLOP/s -- the in-container guide
            # (MI355X_MICROARCH.md) lists no FP64 MFMA figure, so this one is NOT from it.
            from batchelor_amd import _lib as _bl
            a_tf = 4.0 * d * asv["asv_pairs"] / (asv["asv_ms"] * 1e-3) / 1e12
            # HBM bytes per launch from the committed --pmc passes of this very command at sigma 1 (2 x FETCH_SIZE + WRITE_SIZE
            # over the 15 launches of a step); another bandwidth runs other code paths: no number then
            asv_traffic, asv_traffic_note = None, "profiles/r0N_asv_tile_pmc.json holds the counters of the sigma = 1 run only"
            if args.sigma == 1.0 and args.workload == "config5":
                for rnd in ("r06", "r05"):
                    try:
                        rec = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_asv_tile_pmc.json")))["all_launches_total"]
                        asv_traffic = (rec["hbm_read_bytes"] + rec["hbm_write_bytes"]) / 15.0
                        asv_traffic_note = (f"profiles/{rnd}_asv_tile_pmc.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 over a step's 15 "
                                            "launches / 15")
                        break
                    except (OSError, ValueError, KeyError):
                        continue
            line["metric"] = "cells/sec corrected (reducedMNN engine + adjust_shift_variance, 100 PCs)"
            line["dtype"] = "f64 (FP64 MFMA) for adjust_shift_variance; " + kern["dtype"] + " for the searches"
            line["config"]["var_adj_sigma"] = args.sigma
            line["roofline_searches"] = line["roofline"]
            line["roofline"] = {
                "bound": "mfma", "kernel": "asv_tile_kernel<13> (adjust_shift_variance, tiled FP64-MFMA form)",
                "achieved": a_tf, "peak": 78.6, "unit": "TFLOP/s", "frac": a_tf / 78.6,
                "peak_note": "FP64 matrix peak from AMD's MI355X data sheet; not in the in-container guide",
                "traffic": asv_traffic, "traffic_note": asv_traffic_note,
                "launches_per_step": asv["asv_launches"] / max(1, args.steps),
                "avg_launch_ms": asv["asv_ms"] / max(1, asv["asv_launches"]),
                "pairs_per_step": asv["asv_pairs"] / max(1, args.steps),
                "algorithmic_flops_per_step": 4.0 * d * asv["asv_pairs"] / max(1, args.steps),
                "share_of_step": asv["asv_ms"] / (1e3 * elapsed),
                "cells_tiled": _bl.dev_get("asv_tiled_cells") // max(1, args.steps + 0),
                "cells_rerun_in_reference_order": _bl.dev_get("asv_literal_cells") // max(1, args.steps),
                "cells_flagged_beyond_the_rerun": _bl.dev_get("asv_fallback_cells") // max(1, args.steps),
                # 100 MHz ticks added up over the workgroups (256 per launch at this size) over the timed steps (the tallies
                # and ticks are reset when the timed region starts)
                "phase_ms_per_workgroup_per_step": {ph: _bl.dev_get("asv_ticks_" + ph) / 256 / 1e5 / max(1, args.steps)
                                                    for ph in ("stream", "wait", "cells")},
                # the re-run cells' share of the per-cell phase, added up over their workgroups (NOT divided by 256: a re-run
                # cell holds ONE workgroup, and the launch waits for it)
                "rerun_ms_per_step": {"selection_reevaluation_sort": _bl.dev_get("asv_ticks_literal") / 1e5 / max(1, args.steps),
                                      "chains_and_walk": _bl.dev_get("asv_ticks_chains") / 1e5 / max(1, args.steps),
                                      "kept_addends": _bl.dev_get("asv_literal_addends") // max(1, args.steps),
                                      "tiles_with_chains": _bl.dev_get("asv_chain_tiles") // max(1, args.steps)},
            }
        if h2h is not None:
            line["value_host_to_host"] = n_cells / h2h
            line["host_to_host_ms"] = 1e3 * h2h
            line["host_to_host_stages_ms"] = h2h_stages
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"], line["cpu_baseline_variants"] = cpu_baselines(batches, stats, d, k)
                line["host_cores"] = os.cpu_count()
                if args.var_adj and asv["asv_ms"] > 0:
                    # the step is searches + adjust_shift_variance: both on the host cores, added up
                    va = cpu_baseline_var_adj(batches, d, args.sigma, asv["asv_pairs"] / max(1, args.steps))
                    knn_s = n_cells / line["cpu_baseline"]["value"]
                    line["cpu_baseline_searches_only"] = dict(line["cpu_baseline"])
                    line["cpu_baseline"] = {
                        "value": n_cells / (knn_s + va["seconds_per_step"]), "unit": "cells/s", "cores": va["cores"], "kind": "port",
                        "adjust_shift_variance_s_per_step": va["seconds_per_step"], "searches_s_per_step": knn_s,
                        "sample": "searches: " + line["cpu_baseline"]["sample"] + " | adjust_shift_variance: " + va["sample"]}
                try:  # (the CPU baseline is reported with the host's core count and CPU model)
                    with open("/proc/cpuinfo") as f:
                        line["host_cpu_model"] = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), None)
                except OSError:
                    line["host_cpu_model"] = None
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
