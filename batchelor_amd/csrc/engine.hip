// Merge engine (host orchestration of the HIP kernels): R/fastMNN.R:398-562, R/MNN_tree.R:61-77,113-226.
#include "engine.hpp"
#include "host_xfer.hpp"
#include "rccl_dyn.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <thread>

namespace bmx {
namespace {

__global__ void gather_rows_i32(const int32_t* __restrict__ pos, int n, const int32_t* __restrict__ rows,
                                int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rows ? rows[pos[i]] : pos[i];
}

// testing hook of the watchdog: spins on the 100 MHz real-time counter, then ends
__global__ void stall_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// The run's device words go to the host by a STORE into pinned (host-coherent) memory, the sequence number last: the host
// spins on that word instead of waiting for a copy to be scheduled, signalled and polled through the runtime (measured:
// 100-170 us of idle GPU per wait that way, 14 waits per config-3 step).
__global__ void publish_state(const int32_t* __restrict__ st, int32_t* host, int seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = st[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) __hip_atomic_store(&host[i], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host[8], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Several ranks, optimistic searches: a rank's "this search could not be completed on the device" flag (opt_state[0]) is
// all-gathered with the search's lists and OR-ed into every rank's own flag, so that the next wait of EVERY rank sees it and
// all of them start the run over at the same point (a rank that restarted alone would leave the others in a collective).
__global__ void stage_opt_flag(const int32_t* __restrict__ st, int32_t* __restrict__ slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) slot[0] = st[0];
}
__global__ void merge_opt_flags(const int32_t* __restrict__ flags, int world, int32_t* __restrict__ st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int any = 0;
    for (int r = 0; r < world; ++r) any |= flags[4 * r];
    if (any) st[0] = 1;
}

__global__ void iota_offset(int32_t* __restrict__ out, int n, int off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = off + i;
}

// dup_next / dup_head of a merged node's restrict list = those of its children, the right child's positions shifted
__global__ void combine_dup_chains(const int32_t* __restrict__ lnext, const int32_t* __restrict__ lhead, int nl,
                                   const int32_t* __restrict__ rnext, const int32_t* __restrict__ rhead, int nr,
                                   int32_t* __restrict__ next, int32_t* __restrict__ head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nl + nr) return;
    if (i < nl) {
        next[i] = lnext ? lnext[i] : -1;
        head[i] = lhead ? lhead[i] : 1;
    } else {
        const int32_t t = rnext ? rnext[i - nl] : -1;
        next[i] = t >= 0 ? t + nl : -1;
        head[i] = rhead ? rhead[i - nl] : 1;
    }
}

// right cells named more than once: the pairs of all positions of a cell belong to the cell (rowsum groups by cell,
// R/fastMNN.R:571-579): fold[r] = the cell's pairs at its first position, 0 at the others
__global__ void fold_dup_counts(const int32_t* __restrict__ cnt, const int32_t* __restrict__ next, const int32_t* __restrict__ head,
                                int n, int32_t* __restrict__ fold) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int t = 0;
    if (head[r])
        for (int p = r; p >= 0; p = next[p]) t += cnt[p];
    fold[r] = t;
}

__global__ void add_offset_copy(const int32_t* __restrict__ in, int n, int off, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + off;
}

// .choose_k (R/MNN_tree.R:140-146); R's round() is half-to-even = nearbyint in the default rounding mode
int choose_k(int k, double prop_k, int N) {
    if (std::isnan(prop_k)) return k;
    const double r = std::nearbyint(prop_k * (double)N);
    const double m = std::max((double)k, r);
    return (int)std::min((double)N, m);
}

}  // namespace

// The one place the sharded search's layout is decided (the engine and the host-side tests both go through it): n query
// rows are cut into `world` padded slices of rows_per_rank rows; rank r owns rows [r * rows_per_rank, ...) clipped to n,
// and the list buffers hold world * rows_per_rank rows so that the all-gather is in place with equal counts.
int64_t bmx_shard_rows_per_rank(int64_t n, int world) { return world > 0 ? (n + world - 1) / world : n; }

void bmx_shard_range_impl(int64_t n, int rank, int world, int64_t* begin, int64_t* end) {
    const int64_t per = bmx_shard_rows_per_rank(n, world);
    const int64_t b = std::min<int64_t>(n, per * rank);
    *begin = b;
    *end = std::min<int64_t>(n, b + per);
}

Engine::Engine(int device) : device_(device) {
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    stream_ = StreamPool::take();
    scal_.reserve(4096);
}

void Engine::check_alive() const {
    if (dead_)
        throw Error(BMX_ERR_HIP, "the engine is dead after a watchdog timeout (GPU work that never finished); restart the "
                                 "process to get the GPU back");
}

void Engine::mark_dead() {
    dead_ = true;
    cache_.leak = true;  // hipFree would wait for the device
    knn_ws_.abandon = true;
}

void Engine::wait(double work_s) {
    // the deadline covers everything queued since the last wait that came back (queued_work_s_), e.g. the
    // adjust_shift_variance of the previous merge in front of this merge's first search
    try {
        guarded_stream_sync(stream_, wd_base_s_ > 0.0 ? wd_base_s_ + queued_work_s_ + work_s : 0.0);
    } catch (const WatchdogTimeout&) {
        mark_dead();
        throw;
    }
    queued_work_s_ = 0.0;
}

void Engine::debug_stall(int ms) {
    check_alive();
    BMX_HIP(hipSetDevice(device_));
    hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(64), 0, stream_, (unsigned long long)std::max(0, ms) * 100000ull);
    BMX_LAUNCH_CHECK();
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    if (dead_) {
        // nothing on this stream can be waited for: the stream, the communicator and the device blocks are abandoned
        DevBlockCache::current() = &cache_;
        return;
    }
    if (scal_pin_ && scal_pin_bytes_ > KnnWorkspace::kPinnedSmall) (void)hipHostFree(scal_pin_);
    else KnnWorkspace::pinned_small_give(scal_pin_);
    if (stream_) (void)hipStreamSynchronize(stream_);  // (a pair-list copy may still be on its way into the pinned block)
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
    if (pairs_ev_) (void)hipEventDestroy(pairs_ev_);
    if (pairs_ready_ev_) (void)hipEventDestroy(pairs_ready_ev_);
    PinnedBlocks::give(PinnedBlocks::Blk{pairs_pin_, pairs_pin_bytes_});
    if (comm_ && rccl::api().CommDestroy) {
        if (stream_) (void)hipStreamSynchronize(stream_);
        (void)rccl::api().CommDestroy(comm_);
    }
    for (hipEvent_t ev : up_ev_)
        if (ev) (void)hipEventDestroy(ev);
    // (a stream that reports an error is destroyed, an idle one parked for the next engine on this device)
    for (hipStream_t s : {copy_stream_, stream_}) {
        if (!s) continue;
        if (hipStreamSynchronize(s) == hipSuccess) StreamPool::give(s);
        else (void)hipStreamDestroy(s);
    }
    // the members' DevBufs are released after this body: into this engine's cache, which its own destructor frees
    DevBlockCache::current() = &cache_;
}

void Engine::set_shard(int rank, int world, bmx_allgather_fn fn, void* ctx) {
    if (emu_mode_ != 0) throw Error(BMX_ERR_ARG, "the engine is emulating a rank (bmx_engine_emulate): switch that off first");
    if (world < 1 || rank < 0 || rank >= world) throw Error(BMX_ERR_ARG, "invalid rank / world size");
    if (world > 1 && !fn) throw Error(BMX_ERR_ARG, "a multi-rank engine needs an all-gather callback");
    if (comm_ && rccl::api().CommDestroy) {  // a callback replaces an RCCL communicator
        if (stream_) (void)hipStreamSynchronize(stream_);
        (void)rccl::api().CommDestroy(comm_);
        comm_ = nullptr;
    }
    rank_ = rank;
    world_ = world;
    gather_fn_ = fn;
    gather_ctx_ = ctx;
}

void Engine::init_rccl(int rank, int world, const void* unique_id) {
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    if (emu_mode_ != 0) throw Error(BMX_ERR_ARG, "the engine is emulating a rank (bmx_engine_emulate): switch that off first");
    if (world < 1 || rank < 0 || rank >= world) throw Error(BMX_ERR_ARG, "invalid rank / world size");
    if (!unique_id) throw Error(BMX_ERR_ARG, "null RCCL unique id");
    rccl::Api& a = rccl::api();
    if (!a.ready()) throw Error(BMX_ERR_EXCHANGE, "RCCL is not loaded (bmx_rccl_load)");
    if (comm_) {
        BMX_HIP(hipStreamSynchronize(stream_));
        (void)a.CommDestroy(comm_);
        comm_ = nullptr;
    }
    rccl::UniqueId id;
    std::memcpy(id.internal, unique_id, sizeof(id.internal));
    const int rc = a.CommInitRank(&comm_, world, id, rank);
    if (rc != 0) {
        comm_ = nullptr;
        throw Error(BMX_ERR_EXCHANGE, std::string("ncclCommInitRank failed: ") +
                                          (a.GetErrorString ? a.GetErrorString(rc) : std::to_string(rc).c_str()));
    }
    rank_ = rank;
    world_ = world;
    gather_fn_ = nullptr;
    gather_ctx_ = nullptr;
}

// One rank of an N-rank run measured on ONE GPU (bmx_engine_emulate, bench.py --emulate-world): a single-rank run first
// RECORDS what every exchange would have gathered -- the complete array, which a single rank computes itself --, then the
// engine runs as rank r of N: every search covers its slice of the query rows only, every kernel that a multi-rank run
// replicates runs in full, and an exchange fills the OTHER ranks' slices from the recording (a device copy in place of the
// all-gather: what is timed is the rank's own work; the collective's time is modelled beside it from its byte count).
// The results are bit-identical to the recorded run's, so the sequence and the sizes of the exchanges are the same.
void Engine::emulate(int mode, int rank, int world) {
    if (mode < 0 || mode > 2) throw Error(BMX_ERR_ARG, "emulation mode: 0 off, 1 record, 2 replay");
    if (mode == 2 && (world < 1 || rank < 0 || rank >= world)) throw Error(BMX_ERR_ARG, "invalid rank / world size");
    if (comm_ || gather_fn_) throw Error(BMX_ERR_ARG, "emulation is for an engine without a transport");
    if (stream_) BMX_HIP(hipStreamSynchronize(stream_));
    emu_mode_ = mode;
    if (mode == 1) {
        emu_rec_.clear();
        rank_ = 0;
        world_ = 1;
    } else if (mode == 2) {
        rank_ = rank;
        world_ = world;
    } else {
        emu_rec_.clear();
        rank_ = 0;
        world_ = 1;
    }
}

// n doubles from src to dst (a device copy as a kernel: a hipMemcpyAsync between device buffers costs ~200 us of idle stream
// in front of it on this runtime, a launch ~5)
__global__ void copy_doubles_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
    const int64_t n2 = n >> 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<double2*>(dst)[i] = reinterpret_cast<const double2*>(src)[i];
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
}
static void copy_doubles(hipStream_t stream, double* dst, const double* src, int64_t n) {
    if (n <= 0) return;
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) {  // (never for rows of an even d)
        BMX_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        return;
    }
    hipLaunchKernelGGL(copy_doubles_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (n / 2 + 255) / 256))),
                       dim3(256), 0, stream, dst, src, n);
    BMX_LAUNCH_CHECK();
}

// the other ranks' slices of an emulated exchange in ONE launch: bytes [0, lo) and [hi, total) of the recording (zeros when
// there is none: the optimistic-search flags), 16 bytes per thread where the pieces allow
__global__ void emu_fill_kernel(char* __restrict__ dst, const char* __restrict__ src, int64_t lo, int64_t hi, int64_t total) {
    const int64_t n16 = (lo >> 4) + ((total - hi + 15) >> 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t o = i < (lo >> 4) ? i << 4 : hi + ((i - (lo >> 4)) << 4);
        if (o + 16 <= total && ((reinterpret_cast<uintptr_t>(dst + o) | reinterpret_cast<uintptr_t>(src + o)) & 15) == 0) {
            *reinterpret_cast<int4*>(dst + o) = src ? *reinterpret_cast<const int4*>(src + o) : int4{0, 0, 0, 0};
        } else {
            for (int64_t b = o; b < o + 16 && b < total; ++b) dst[b] = src ? src[b] : 0;
        }
    }
    // (the ragged end of the first piece)
    if (blockIdx.x == 0 && threadIdx.x < (lo & 15)) {
        const int64_t b = (lo & ~(int64_t)15) + threadIdx.x;
        dst[b] = src ? src[b] : 0;
    }
}

void Engine::exchange(void* buf, int64_t bytes_per_rank, bool flags_only) {
    if (emu_mode_ == 1) {  // record: this (single) rank's slice is the whole array
        if (flags_only) return;
        if (emu_next_ >= emu_rec_.size()) emu_rec_.emplace_back();
        auto& r = emu_rec_[emu_next_++];
        r.second = bytes_per_rank;
        BMX_HIP(hipMemcpyAsync(r.first.reserve((size_t)std::max<int64_t>(bytes_per_rank, 1)), buf, (size_t)bytes_per_rank,
                               hipMemcpyDeviceToDevice, stream_));
        return;
    }
    if (emu_mode_ == 2) {
        ++xchg_calls_;
        xchg_bytes_ += bytes_per_rank * world_;
        char* base = static_cast<char*>(buf);
        const int64_t mine = (int64_t)rank_ * bytes_per_rank;
        if (flags_only) {  // the other ranks' optimistic-search flags: all clear (the recorded run went through)
            const int64_t total = (int64_t)world_ * bytes_per_rank;
            hipLaunchKernelGGL(emu_fill_kernel, dim3(1), dim3(256), 0, stream_, base, (const char*)nullptr, mine,
                               mine + bytes_per_rank, total);
            BMX_LAUNCH_CHECK();
            return;
        }
        if (emu_next_ >= emu_rec_.size()) throw Error(BMX_ERR_ARG, "emulated run: more exchanges than the recorded run made");
        const auto& r = emu_rec_[emu_next_++];
        // (the recording must be of THIS job: the same sequence of exchanges, each no larger than this call's padded array)
        if (r.second > (int64_t)world_ * bytes_per_rank)
            throw Error(BMX_ERR_ARG, "emulated run: an exchange of the recorded run is larger than this run's (other inputs, "
                                     "parameters or tree than the recording?)");
        // bytes [0, mine) and [mine + per, total) of the recording, where they exist
        const int64_t total = r.second, lo = std::min(mine, total), hi = std::min(mine + bytes_per_rank, total);
        const int64_t n16 = (lo >> 4) + ((total - hi + 15) >> 4);
        if (n16 > 0 || (lo & 15)) {
            hipLaunchKernelGGL(emu_fill_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (n16 + 255) / 256))),
                               dim3(256), 0, stream_, base, (const char*)r.first.p, lo, hi, total);
            BMX_LAUNCH_CHECK();
        }
        return;
    }
    // testing hook (bmx_dev_set "exchange_always"): a single rank goes through its transport too (an all-gather of one)
    const bool always = dev_knobs().exchange_always != 0;
    if (world_ == 1 && !(always && (comm_ || gather_fn_))) return;
    ++xchg_calls_;
    xchg_bytes_ += bytes_per_rank * world_;
    if (comm_) {
        // in place (send buffer = this rank's slice of the receive buffer), ordered on the engine's stream: no host
        // synchronisation, the next kernel simply queues behind the collective
        char* base = static_cast<char*>(buf);
        const int rc = rccl::api().AllGather(base + (int64_t)rank_ * bytes_per_rank, base, (size_t)bytes_per_rank,
                                             /* ncclUint8 */ 1, comm_, stream_);
        if (rc != 0)
            throw Error(BMX_ERR_EXCHANGE, std::string("ncclAllGather failed: ") +
                                              (rccl::api().GetErrorString ? rccl::api().GetErrorString(rc) : "?"));
        return;
    }
    BMX_HIP(hipStreamSynchronize(stream_));
    const int rc = gather_fn_(gather_ctx_, buf, bytes_per_rank);
    if (rc != 0) throw Error(BMX_ERR_EXCHANGE, "the all-gather callback failed with code " + std::to_string(rc));
}

void Engine::ensure_uploaded(int b) {
    if (uploaded_[b]) return;
    const size_t bytes = (size_t)nrows_[b] * d_ * sizeof(double);
    if (!copy_stream_) copy_stream_ = StreamPool::take();
    if ((int)up_ev_.size() < B_) {
        const size_t old = up_ev_.size();
        up_ev_.resize(B_, nullptr);
        for (size_t i = old; i < up_ev_.size(); ++i) BMX_HIP(hipEventCreateWithFlags(&up_ev_[i], hipEventDisableTiming));
    }
    upload_pageable(inputs_cm_[b].p, host_data_[b], bytes, copy_stream_);
    BMX_HIP(hipEventRecord(up_ev_[b], copy_stream_));
    uploaded_[b] = 1;
}

void Engine::prefetch_one() {
    if (!lazy_) return;
    while (need_pos_ < need_order_.size() && uploaded_[need_order_[need_pos_]]) ++need_pos_;
    if (need_pos_ < need_order_.size()) ensure_uploaded(need_order_[need_pos_++]);
}

void Engine::upload(int nbatches, int d, const double* const* data, const int32_t* nrows,
                    const int32_t* const* restrict_idx, const int32_t* n_restrict, bool lazy) {
    check_alive();
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    root_.reset();  // results of an earlier run describe other inputs
    merges_.clear();
    if (emu_mode_ != 0) {  // a recording describes the exchanges of the inputs it was made on
        emu_rec_.clear();
        emu_mode_ = 0;
        rank_ = 0;
        world_ = 1;
    }
    if (nbatches < 2) throw Error(BMX_ERR_ARG, "at least two batches must be specified");  // R/fastMNN.R:345
    if (d < 1 || d > 256) throw Error(BMX_ERR_ARG, "number of dimensions must be in [1, 256]");
    B_ = nbatches;
    d_ = d;
    N_ = 0;
    nrows_.assign(nrows, nrows + nbatches);
    inputs_cm_.clear();
    inputs_cm_.resize(nbatches);
    inputs_restrict_.clear();
    inputs_restrict_.resize(nbatches);
    inputs_dup_.clear();
    inputs_dup_.resize(nbatches);
    has_dups_.assign(nbatches, 0);
    n_restrict_.assign(nbatches, -1);
    lazy_ = lazy;
    host_data_.assign(data, data + nbatches);
    uploaded_.assign(nbatches, 0);
    need_order_.clear();
    need_pos_ = 0;
    for (int b = 0; b < nbatches; ++b) {
        if (nrows[b] < 1) throw Error(BMX_ERR_ARG, "every batch needs at least one cell");
        N_ += nrows[b];
        double* p = inputs_cm_[b].reserve((size_t)nrows[b] * d);
        // the caller's matrices are pageable (R-owned): through the pinned staging ring at link speed (host_xfer.hpp)
        if (!lazy) {
            upload_pageable(p, data[b], (size_t)nrows[b] * d * sizeof(double), stream_);
            uploaded_[b] = 1;
        }
        const bool has = restrict_idx && restrict_idx[b] && n_restrict && n_restrict[b] >= 0;
        if (has) {
            const int m = n_restrict[b];
            if (m == 0) throw Error(BMX_ERR_ARG, "no cells remaining in a batch after restriction");  // R/checkInputs.R:116
            // an R subsetting vector in the caller's order (R/checkInputs.R:96-120 keeps it as given); a cell may be named
            // more than once (see Node::restrict_dups)
            std::vector<int32_t> z(m);
            std::vector<int32_t> last((size_t)nrows[b], -1), chain((size_t)2 * m, -1);
            bool dups = false;
            for (int i = 0; i < m; ++i) {
                const int32_t v = restrict_idx[b][i];
                if (v < 1 || v > nrows[b]) throw Error(BMX_ERR_SUBSET, "subset indices out of range");
                z[i] = v - 1;
                chain[(size_t)m + i] = last[v - 1] < 0 ? 1 : 0;  // head
                if (last[v - 1] >= 0) {
                    chain[last[v - 1]] = i;  // next
                    dups = true;
                }
                last[v - 1] = i;
            }
            int32_t* rp = inputs_restrict_[b].reserve(m);
            BMX_HIP(hipMemcpyAsync(rp, z.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
            if (dups) {
                int32_t* dp = inputs_dup_[b].reserve((size_t)2 * m);
                BMX_HIP(hipMemcpyAsync(dp, chain.data(), (size_t)2 * m * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
            }
            has_dups_[b] = dups ? 1 : 0;
            BMX_HIP(hipStreamSynchronize(stream_));  // z, chain go out of scope
            n_restrict_[b] = m;
        }
    }
    if (N_ > std::numeric_limits<int32_t>::max() / 2) throw Error(BMX_ERR_ARG, "too many cells for int32 indices");
    BMX_HIP(hipStreamSynchronize(stream_));
    if (!lazy) host_data_.clear();  // the caller's matrices are free again
}

void Engine::knn(const double* X, const int32_t* ref_rows, int nr, const double* Q, const int32_t* q_rows, int nq,
                 int k, int32_t* idx, double* dist, const float* seed_d2, const double* centre, double* kth, bool gather) {
    // query rows are split over ranks; the padded per-rank slices are contiguous, so the all-gather is in place
    int64_t b = 0, e = nq;
    bmx_shard_range_impl(nq, rank_, world_, &b, &e);
    // watchdog budget of this search's waits: ~1e4 times what the candidate pass takes per pair evaluation (1.5e-13 s),
    // and enough for a search that falls through to the FP64 scan (1e-10 s per pair and dimension)
    queued_work_s_ += 2e-10 * (double)nq * (double)nr * (double)d_ / 50.0;
    knn_ws_.wd_budget_s = wd_base_s_ > 0.0 ? wd_base_s_ + queued_work_s_ : 0.0;
    try {
        knn_device(stream_, knn_ws_, X, ref_rows, nr, Q, q_rows, nq, d_, k, idx, dist, (int)b, (int)e, seed_d2, centre, kth);
    } catch (const WatchdogTimeout&) {  // (the search's own waits go through knn_ws_.sync, not through wait())
        mark_dead();
        throw;
    }
    {
        const int64_t per = bmx_shard_rows_per_rank(nq, world_);
        // indices and distances of one search: grouped, RCCL sends them as one launch
        const bool flags = world_ > 1 && knn_ws_.optimistic;  // (see stage_opt_flag)
        int32_t* xf = nullptr;
        if (flags) {
            xf = xflags_.reserve((size_t)4 * world_);
            hipLaunchKernelGGL(stage_opt_flag, dim3(1), dim3(64), 0, stream_, (const int32_t*)knn_ws_.opt_state_ptr(stream_),
                               xf + 4 * rank_);
            BMX_LAUNCH_CHECK();
        }
        const bool group = (dist || kth || flags) && comm_ && world_ > 1 && rccl::api().GroupStart && rccl::api().GroupEnd;
        if (group) (void)rccl::api().GroupStart();
        try {
            // (gather = false: the caller goes on with its own slice of the lists and exchanges what it makes of them)
            if (gather) exchange(idx, per * k * (int64_t)sizeof(int32_t));
            if (gather && dist) exchange(dist, per * k * (int64_t)sizeof(double));
            if (gather && kth) exchange(kth, per * (int64_t)sizeof(double));
            if (flags) exchange(xf, 4 * (int64_t)sizeof(int32_t), /* flags_only */ true);
        } catch (...) {
            if (group) (void)rccl::api().GroupEnd();  // never leave the group open behind an error
            throw;
        }
        if (group && rccl::api().GroupEnd() != 0) throw Error(BMX_ERR_EXCHANGE, "ncclGroupEnd failed");
        if (flags) {
            hipLaunchKernelGGL(merge_opt_flags, dim3(1), dim3(64), 0, stream_, (const int32_t*)xf, world_,
                               knn_ws_.opt_state_ptr(stream_));
            BMX_LAUNCH_CHECK();
        }
    }
}

// The run's small device words (KnnWorkspace::opt_state: [0] optimistic search failed, [1] queries through the bounded exact
// sweep, [2] listed left cells, [3] pairs, [4] MNN-involved right cells) come back in ONE 32-byte copy per wait.
const int32_t* Engine::read_state() {
    int32_t* st = knn_ws_.opt_state_ptr(stream_);
    int32_t* pin = reinterpret_cast<int32_t*>(knn_ws_.pinned_words() + 8);  // [0..7] the words, [8] the sequence number
    if (state_seq_ == 0) pin[8] = 0;  // (a pooled block may hold an earlier engine's numbers)
    const int seq = ++state_seq_;
    hipLaunchKernelGGL(publish_state, dim3(1), dim3(64), 0, stream_, (const int32_t*)st, pin, seq);
    BMX_LAUNCH_CHECK();
    prefetch_one();  // (lazy upload: the host has nothing to do until the GPU is through -- move the next batch)
    // the wait: spin on the sequence word (host memory: no runtime call in the loop), with the watchdog's deadline
    const double budget = wd_base_s_ > 0.0 ? wd_base_s_ + queued_work_s_ : 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    volatile int32_t* vp = pin;
    unsigned spins = 0;
    while (vp[8] != seq) {
        if ((++spins & 0x3FFu) == 0) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (budget > 0.0 && el > budget) {
                mark_dead();
                throw WatchdogTimeout("watchdog: the GPU work queued on the engine's stream did not finish in time; the engine is "
                                      "dead (restart the process)");
            }
            if (el > 0.5) {  // a long wait (adjust_shift_variance at scale): stop burning the core, and look for errors
                const hipError_t e = hipStreamQuery(stream_);
                (void)hipGetLastError();
                if (e != hipSuccess && e != hipErrorNotReady)
                    throw Error(BMX_ERR_HIP, std::string("the engine's stream failed: ") + hipGetErrorString(e));
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    queued_work_s_ = 0.0;
    if (pin[0] != 0) {
        // a search of this rank alone (auto-merge's counts, dealt over the ranks) must not restart this rank alone: note it,
        // lower the device flag, go on; gather_counts tells everybody
        if (!solo_) throw OptimisticRetry();
        solo_flag_ = true;
        BMX_HIP(hipMemsetAsync(st, 0, sizeof(int32_t), stream_));
    }
    return pin;
}

Engine::MnnOut Engine::find_mnn(const Node& left, const Node& right, int k, double prop_k, const double* mu_left,
                                const double* mu_right) {
    // .restricted_mnn (R/MNN_tree.R:113-133): search among the restricted rows only
    const int nL = left.has_restrict ? left.n_restrict : left.n;
    const int nR = right.has_restrict ? right.n_restrict : right.n;
    const int32_t* lrows = left.has_restrict ? left.restrict_rows.p : nullptr;
    const int32_t* rrows = right.has_restrict ? right.restrict_rows.p : nullptr;
    MnnOut o;
    o.k1 = std::min(choose_k(k, prop_k, nL), nL);  // neighbours sought in LEFT for each right cell
    o.k2 = std::min(choose_k(k, prop_k, nR), nR);  // neighbours sought in RIGHT for each left cell
    if (o.k1 < 1 || o.k2 < 1) throw Error(BMX_ERR_ARG, "'k' must be positive");
    int32_t* st = knn_ws_.opt_state_ptr(stream_);
    const int64_t perR = bmx_shard_rows_per_rank(nR, world_) * (int64_t)world_;
    int32_t* idxRL = idxRL_.reserve((size_t)perR * o.k1);
    // 1. every right cell's neighbours in LEFT, with their distances
    double* distRL = distRL_.reserve((size_t)perR * o.k1);
    knn(left.data.p, lrows, nL, right.data.p, rrows, nR, o.k1, idxRL, distRL, nullptr, mu_left);
    // 2. a pair needs its left cell in some right cell's list, so only those left cells are searched in RIGHT (with a
    //    growing merged reference most left cells are in nobody's list).  The result is the same set of pairs.
    const size_t stamp_cap = stampL_.cap;
    int32_t* stamp = stampL_.reserve(nL);
    if (stampL_.cap != stamp_cap) {  // a fresh block: stamps start at zero, the searches' numbers at one
        BMX_HIP(hipMemsetAsync(stamp, 0, stampL_.cap * sizeof(int32_t), stream_));
        if (stamp_gen_ > 0x7FFFFF00) stamp_gen_ = 0;
    }
    if (stamp_gen_ > 0x7FFFFF00) {
        BMX_HIP(hipMemsetAsync(stamp, 0, stampL_.cap * sizeof(int32_t), stream_));
        stamp_gen_ = 0;
    }
    const int gen = ++stamp_gen_;
    int32_t* offSel = offSel_.reserve((size_t)nL + 1);
    int32_t* lsel = lsel_.reserve(nL);
    // (an upper bound sizes what the selected cells accumulate into: their seeds and pair masks are zeroed as they are selected)
    const int64_t perLmax = bmx_shard_rows_per_rank(nL, world_) * (int64_t)world_;
    float* seed = seedL_.reserve((size_t)std::max<int64_t>(1, perLmax));
    unsigned long long* maskL = maskL_.reserve(std::max(1, nL));
    select_listed_rows(stream_, scan_ws_, idxRL, (int64_t)nR * o.k1, nL, stamp, gen, offSel, lsel, st + 2, seed, maskL);
    // A right cell r can only pair with a left cell l if it lists l, at a distance search 1 has just measured: the
    // largest such distance bounds how far search 2 has to look for l -- a tight starting threshold for free.  Rows may
    // come back short, padded with -1.  (The seeds only need the selected cells' positions, not their number.)
    seed_thresholds(stream_, idxRL, distRL, (int64_t)nR * o.k1, offSel, nL, seed, st + 2, rank_, world_);
    const int32_t nsel = read_state()[2];
    o.nsel = nsel;
    if (debug_prints()) fprintf(stderr, "[bmx] find_mnn: %d of %d left cells are in some right cell's list\n", nsel, nL);
    const int32_t* qsel = lsel;  // rows of left.data to query with
    if (lrows) {
        int32_t* q = qsel_.reserve(nsel);
        compose_row_list(stream_, lsel, nsel, lrows, q);
        qsel = q;
    }
    const int64_t perL = bmx_shard_rows_per_rank(nsel, world_) * (int64_t)world_;
    int32_t* idxLR = idxLR_.reserve((size_t)std::max<int64_t>(1, perL) * o.k2);
    double* kthL = kthL_.reserve((size_t)std::max<int64_t>(1, perL));
    knn(right.data.p, rrows, nR, left.data.p, qsel, nsel, o.k2, idxLR, nullptr, seed, mu_right, kthL);
    int32_t* cntL = o.k2 > 64 ? cntL_.reserve(std::max(1, nsel)) : nullptr;
    int32_t* offL = offL_.reserve((size_t)nsel + 1);
    int32_t* partR = partR_.reserve((size_t)nR * o.k1);
    int32_t* cntR = cntR_.reserve(nR);
    int32_t* offR = offR_.reserve((size_t)nR + 1);
    int32_t* second_u = second_u_.reserve(nR);
    mutual_counts(stream_, idxLR, nsel, o.k2, idxRL, nR, o.k1, cntL, partR, cntR, lsel, offSel, maskL, /* mask_is_clear */ true,
                  distRL, kthL, &sorted_);
    const int32_t* cntR_cells = cntR;
    if (right.restrict_dups) {
        int32_t* fold = cntFold_.reserve(nR);
        hipLaunchKernelGGL(fold_dup_counts, dim3(cdiv(nR, 256)), dim3(256), 0, stream_, (const int32_t*)cntR,
                           (const int32_t*)right.dup_next.p, (const int32_t*)right.dup_head.p, nR, fold);
        BMX_LAUNCH_CHECK();
        cntR_cells = fold;
    }
    pair_scans(stream_, scan_ws_, maskL, cntL, nsel, o.k2, offL, st + 3, cntR_cells, nR, offR, second_u, st + 4);
    const int32_t* pin = read_state();
    o.P = pin[3];
    o.U = pin[4];
    return o;
}

namespace {
void segment_layout(const Node& node, std::vector<int>& starts, std::vector<int>& ns) {
    int r0 = 0;
    for (const Segment& s : node.origin) {
        starts.push_back(r0);
        ns.push_back(s.n);
        r0 += s.n;
    }
}
}  // namespace

void Engine::node_mean(const Node& node, double* mu) {
    bool fresh = !node.has_restrict && node.origin.size() <= 16 && node.stat_slot.size() == node.origin.size();
    for (int sl : node.stat_slot) fresh = fresh && sl >= 0;
    if (fresh) {
        std::vector<int> starts, ns;
        segment_layout(node, starts, ns);
        node_mean_from_segments(stream_, means_pool_.p, ns.data(), node.stat_slot.data(), (int)ns.size(), d_, mu);
    } else if (node.has_restrict) {
        col_reduce(stream_, red_ws_, node.data.p, node.restrict_rows.p, 0, node.n_restrict, d_, 0, nullptr,
                   1.0 / (double)node.n_restrict, mu);
    } else {
        col_reduce(stream_, red_ws_, node.data.p, nullptr, 0, node.n, d_, 0, nullptr, 1.0 / (double)node.n, mu);
    }
}

void Engine::row_pass(Node& node, const std::vector<int>& vec_ids, bool with_stats, const double* mu_known) {
    if (vec_ids.empty() && !with_stats) return;
    double* mu_own = vecs_.p + (size_t)(2 * B_ + 5) * d_;
    if (!vec_ids.empty() && !mu_known) node_mean(node, mu_own);
    const double* mu = mu_known ? mu_known : mu_own;
    std::vector<int> starts, ns, slots;
    segment_layout(node, starts, ns);
    if (with_stats) {
        if (n_slots_ + (int)ns.size() > slot_cap_) throw Error(BMX_ERR_ARG, "internal: statistics slots exhausted");
        for (size_t i = 0; i < ns.size(); ++i) slots.push_back(n_slots_++);
    }
    rows_apply_stats(stream_, red_ws_, node.data.p, d_, starts.data(), ns.data(), (int)ns.size(), mu, vecs_.p,
                     vec_ids.data(), (int)vec_ids.size(), with_stats ? slots.data() : nullptr, means_pool_.p, scal_.p);
    if (with_stats)
        node.stat_slot = slots;
    else
        node.stat_slot.assign(node.origin.size(), -1);  // the rows moved: their statistics are stale
}

void Engine::ensure_stats(Node& node) { ensure_stats2(node, nullptr); }

void Engine::ensure_stats2(Node& a, Node* b) {
    // .compute_perbatch_var (R/fastMNN.R:651-658) only for the segments whose rows changed since their last statistics; the
    // stale segments of both nodes of a merge go through ONE pass (+ one small launch that finishes the sums)
    std::vector<RowSeg> segs;
    std::vector<std::pair<Node*, int>> which;
    for (Node* node : {&a, b}) {
        if (!node) continue;
        if (node->stat_slot.size() != node->origin.size()) node->stat_slot.assign(node->origin.size(), -1);
        int r0 = 0;
        for (size_t i = 0; i < node->origin.size(); ++i) {
            if (node->stat_slot[i] < 0) {
                if (n_slots_ + 1 > slot_cap_) throw Error(BMX_ERR_ARG, "internal: statistics slots exhausted");
                segs.push_back(RowSeg{node->data.p, r0, node->origin[i].n, nullptr, nullptr, n_slots_++});
                which.emplace_back(node, (int)i);
            }
            r0 += node->origin[i].n;
        }
    }
    if (segs.empty()) return;
    rows_multi(stream_, red_ws_, d_, segs.data(), (int)segs.size(), vecs_.p, nullptr, 0, true, means_pool_.p, scal_.p);
    for (size_t i = 0; i < which.size(); ++i) which[i].first->stat_slot[which[i].second] = segs[i].slot;
}

void Engine::node_means(const Node& left, const Node& right, double* mu_l, double* mu_r) {
    auto fresh = [](const Node& n) {
        bool f = !n.has_restrict && n.stat_slot.size() == n.origin.size();
        for (int sl : n.stat_slot) f = f && sl >= 0;
        return f;
    };
    if (fresh(left) && fresh(right) && left.origin.size() + right.origin.size() <= 16) {
        std::vector<int> ns, slots;
        for (const Node* n : {&left, &right})
            for (size_t i = 0; i < n->origin.size(); ++i) {
                ns.push_back(n->origin[i].n);
                slots.push_back(n->stat_slot[i]);
            }
        node_means_from_segments(stream_, means_pool_.p, ns.data(), slots.data(), (int)left.origin.size(),
                                 (int)right.origin.size(), d_, mu_l, mu_r);
    } else {
        node_mean(left, mu_l);
        node_mean(right, mu_r);
    }
}

void Engine::centre_both(Node& left, Node& right, int vid, const double* mu_l, const double* mu_r) {
    // .center_along_batch_vector on both sides (R/fastMNN.R:496-497) with the "new" variances (:498-499) out of the same
    // pass.  The shift of a segment's one-pass variance: its old mean where its statistics are current, else (the side that
    // has just been orthogonalised) its node's mean -- any fixed vector inside the cloud serves
    std::vector<RowSeg> segs;
    for (Node* node : {&left, &right}) {
        int r0 = 0;
        for (size_t i = 0; i < node->origin.size(); ++i) {
            if (n_slots_ + 1 > slot_cap_) throw Error(BMX_ERR_ARG, "internal: statistics slots exhausted");
            const double* mu = node == &left ? mu_l : mu_r;
            const bool cur = node->stat_slot.size() == node->origin.size() && node->stat_slot[i] >= 0;
            segs.push_back(RowSeg{node->data.p, r0, node->origin[i].n, mu,
                                  cur ? means_pool_.p + (size_t)node->stat_slot[i] * d_ : mu, n_slots_++});
            r0 += node->origin[i].n;
        }
    }
    rows_multi(stream_, red_ws_, d_, segs.data(), (int)segs.size(), vecs_.p, &vid, 1, true, means_pool_.p, scal_.p);
    size_t j = 0;
    for (Node* node : {&left, &right}) {
        node->stat_slot.assign(node->origin.size(), -1);
        for (size_t i = 0; i < node->origin.size(); ++i) node->stat_slot[i] = segs[j++].slot;
    }
}

void Engine::orthogonalize(Node& node, const std::vector<int>& extras) {
    // .orthogonalize_other (R/fastMNN.R:642-647): sequentially, each on the result of the previous -- one pass
    row_pass(node, extras, false);
}

std::unique_ptr<Node> Engine::clone_node(const Node& src) {
    auto n = std::make_unique<Node>();
    n->index = src.index;
    n->n = src.n;
    n->origin = src.origin;
    n->stat_slot = src.stat_slot;
    n->extras = src.extras;
    n->has_restrict = src.has_restrict;
    n->n_restrict = src.n_restrict;
    double* p = n->data.reserve((size_t)src.n * d_);
    BMX_HIP(hipMemcpyAsync(p, src.data.p, (size_t)src.n * d_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    if (src.has_restrict) {
        int32_t* r = n->restrict_rows.reserve(src.n_restrict);
        BMX_HIP(hipMemcpyAsync(r, src.restrict_rows.p, (size_t)src.n_restrict * sizeof(int32_t),
                               hipMemcpyDeviceToDevice, stream_));
        if (src.restrict_dups) {
            n->restrict_dups = true;
            const size_t m = (size_t)src.n_restrict;
            BMX_HIP(hipMemcpyAsync(n->dup_next.reserve(m), src.dup_next.p, m * sizeof(int32_t), hipMemcpyDeviceToDevice, stream_));
            BMX_HIP(hipMemcpyAsync(n->dup_head.reserve(m), src.dup_head.p, m * sizeof(int32_t), hipMemcpyDeviceToDevice, stream_));
        }
    }
    return n;
}

int Engine::count_mnn_pairs(const Node& left, const Node& right, const bmx_params_t& p) {
    // one (left, right) evaluation of .count_mnn_pairs (R/MNN_tree.R:171-193) on orthogonalised COPIES
    const Node* l = &left;
    const Node* r = &right;
    std::unique_ptr<Node> lc, rc;
    if (!left.extras.empty()) {
        rc = clone_node(right);
        orthogonalize(*rc, left.extras);
        r = rc.get();
    }
    if (!right.extras.empty()) {
        lc = clone_node(left);
        orthogonalize(*lc, right.extras);
        l = lc.get();
    }
    const MnnOut o = find_mnn(*l, *r, p.k, p.prop_k);
    return (int)o.P;
}

void Engine::merge_step(int mdx, Node& left, Node& right, const bmx_params_t& p, std::unique_ptr<Node>& merged) {
    MergeRecord& rec = merges_[mdx];
    rec.left_set = left.index;
    rec.right_set = right.index;
    rec.var_batches.clear();
    for (const Segment& s : left.origin) rec.var_batches.push_back(s.batch);
    for (const Segment& s : right.origin) rec.var_batches.push_back(s.batch);

    std::unique_ptr<Section> sec = std::make_unique<Section>(this);  // streaming section 1: statistics, orthogonalisation
    // "old" variances (R/fastMNN.R:467-468): a segment untouched since its last statistics keeps them (the left
    // node's segments carry the "new" variances of the merge that made it); the stale ones of both nodes in one pass
    ensure_stats2(left, &right);
    rec.old_slot = left.stat_slot;
    rec.old_slot.insert(rec.old_slot.end(), right.stat_slot.begin(), right.stat_slot.end());

    // the restrict-row column means of both sides: centring along a vector never moves them, so the ones taken here
    // (from the fresh segment statistics where there is no restriction) serve every pass of this merge
    double* mu_l = vecs_.p + (size_t)(2 * B_ + 6) * d_;
    double* mu_r = vecs_.p + (size_t)(2 * B_ + 7) * d_;
    node_means(left, right, mu_l, mu_r);
    row_pass(right, left.extras, false, mu_r);  // .orthogonalize_other, R/fastMNN.R:473-474
    row_pass(left, right.extras, false, mu_l);

    if (mdx == snap_merge_) {
        snap_nl_ = left.n;
        snap_nr_ = right.n;
        BMX_HIP(hipMemcpyAsync(snap_l_.reserve((size_t)left.n * d_), left.data.p, (size_t)left.n * d_ * sizeof(double),
                               hipMemcpyDeviceToDevice, stream_));
        BMX_HIP(hipMemcpyAsync(snap_r_.reserve((size_t)right.n * d_), right.data.p,
                               (size_t)right.n * d_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    }
    sec.reset();
    const MnnOut mo = find_mnn(left, right, p.k, p.prop_k, mu_l, mu_r);  // R/fastMNN.R:476-477
    if (mo.P == 0) throw Error(BMX_ERR_NO_PAIRS, "no mutual nearest neighbours found between batches");
    sec = std::make_unique<Section>(this);  // streaming section 2: pairs, averaging, centring + statistics
    const int nLs = left.has_restrict ? left.n_restrict : left.n;
    const int nRs = right.has_restrict ? right.n_restrict : right.n;
    const int32_t* lrows = left.has_restrict ? left.restrict_rows.p : nullptr;
    const int32_t* rrows = right.has_restrict ? right.restrict_rows.p : nullptr;
    rec.npairs = mo.P;
    rec.stats[0] = nLs;
    rec.stats[1] = nRs;
    rec.stats[2] = mo.U;
    rec.stats[3] = mo.P;
    rec.stats[4] = left.n;
    rec.stats[5] = right.n;
    int32_t* first = rec.first.reserve((size_t)mo.P);
    int32_t* second = rec.second.reserve((size_t)mo.P);
    emit_pairs(stream_, idxLR_.p, mo.nsel, mo.k2, idxRL_.p, mo.k1, offL_.p, lrows, rrows, first, second, lsel_.p,
               maskL_.p, &sorted_);

    // .average_correction + overall.batch (R/fastMNN.R:480-481); the column means of the averaged vectors, the mean squares
    // .get_batch_magnitude wants (R/fastMNN.R:582-595) and the magnitude itself come out of the same pass where its fused
    // form applies
    double* averaged = averaged_.reserve((size_t)mo.U * d_);
    const int vid = n_extras_;  // slot of this merge's overall.batch in the pool
    double* overall = vecs_.p + (size_t)vid * d_;
    double* msq = vecs_.p + (size_t)(2 * B_ + 4) * d_;
    rec.batch_size_na = std::isnan(p.min_batch_skip);
    rec.skipped = false;
    rec.bs_slot = -1;
    if (!rec.batch_size_na) {
        if (n_slots_ + 1 > slot_cap_) throw Error(BMX_ERR_ARG, "internal: statistics slots exhausted");
        rec.bs_slot = n_slots_++;
    }
    double* mag = rec.bs_slot >= 0 ? scal_.p + rec.bs_slot : nullptr;
    const int32_t* rnext = right.restrict_dups ? right.dup_next.p : nullptr;
    // (several ranks: this averaging's workgroups are dealt over them, the column sums all-gathered -- see AvgShard)
    const bool avg_sharded = world_ > 1 || emu_mode_ == 1 || (dev_knobs().exchange_always != 0 && (comm_ || gather_fn_));
    AvgShard ash{rank_, world_, [this](void* buf, int64_t bytes) { exchange(buf, bytes); }};
    if (!average_correction(stream_, red_ws_, left.data.p, lrows, right.data.p, rrows, d_, second_u_.p, mo.U, partR_.p, cntR_.p,
                            mo.k1, averaged, true, overall, msq, mag, nullptr, rnext, avg_sharded ? &ash : nullptr)) {
        if (rec.batch_size_na) {
            col_reduce(stream_, red_ws_, averaged, nullptr, 0, mo.U, d_, 0, nullptr, 1.0 / (double)mo.U, overall);
        } else {
            col_reduce2(stream_, red_ws_, averaged, mo.U, d_, 1.0 / (double)mo.U, overall, msq);
            batch_magnitude(stream_, overall, msq, d_, mag);
        }
    }

    bool do_correct = true;
    if (!rec.batch_size_na && p.min_batch_skip > 0.0) {
        // the host only looks at the magnitude here when a merge can actually be skipped (min.batch.skip > 0), otherwise
        // with everything else at the end of the run
        double* hp = reinterpret_cast<double*>(knn_ws_.pinned_words() + 4);
        BMX_HIP(hipMemcpyAsync(hp, scal_.p + rec.bs_slot, sizeof(double), hipMemcpyDeviceToHost, stream_));
        wait();
        const double h = *hp;
        if (h < p.min_batch_skip) {
            do_correct = false;
            rec.skipped = true;
        }
    }

    if (do_correct) {
        // R/fastMNN.R:496-501: centre both sides along the batch vector; the "new" variances come out of the same pass
        centre_both(left, right, vid, mu_l, mu_r);
        rec.new_slot = left.stat_slot;
        rec.new_slot.insert(rec.new_slot.end(), right.stat_slot.begin(), right.stat_slot.end());

        // R/fastMNN.R:505-507: re-average on the centred data, then the tricube-smoothed correction of the right batch;
        // the rows of the MNN-involved right cells (the reference list of the tricube search) are written on the way
        int32_t* srows = second_rows_.reserve(mo.U);
        if (!average_correction(stream_, red_ws_, left.data.p, lrows, right.data.p, rrows, d_, second_u_.p, mo.U, partR_.p,
                                cntR_.p, mo.k1, averaged, false, nullptr, nullptr, nullptr, srows, rnext)) {
            hipLaunchKernelGGL(gather_rows_i32, dim3(cdiv(mo.U, 256)), dim3(256), 0, stream_, second_u_.p, mo.U, rrows, srows);
            BMX_LAUNCH_CHECK();
        }
        const int k_tc = choose_k(p.k, p.prop_k, right.n);  // unrestricted size of the right batch
        const int safe_k = std::min(k_tc, mo.U);
        const int64_t per1 = bmx_shard_rows_per_rank(right.n, world_);
        const int64_t per = per1 * (int64_t)world_;
        int32_t* idxT = idxT_.reserve((size_t)per * safe_k);
        double* distT = distT_.reserve((size_t)per * safe_k);
        // Several ranks (without var_adj): every rank has the neighbour lists of ITS slice of the right cells from the search;
        // it corrects those rows and the CORRECTED ROWS are all-gathered (n x d x 8 bytes) instead of the lists (n x k x 12):
        // the same order of bytes, and the apply is sharded with the search instead of being repeated on every rank.
        // (testing hook "exchange_always": a single rank with a transport takes this way too, an all-gather of one)
        const bool rows_sharded = !p.var_adj && (world_ > 1 || emu_mode_ == 1 ||
                                                 (dev_knobs().exchange_always != 0 && (comm_ || gather_fn_)));
        sec.reset();
        knn(right.data.p, srows, mo.U, right.data.p, nullptr, right.n, safe_k, idxT, distT, nullptr, mu_r, nullptr, !rows_sharded);
        sec = std::make_unique<Section>(this);  // streaming section 3: tricube apply, rbind
        if (rows_sharded) {
            int64_t b = 0, e = right.n;
            bmx_shard_range_impl(right.n, rank_, world_, &b, &e);
            if (e > b)
                tricube_apply(stream_, right.data.p + (size_t)b * d_, (int)(e - b), d_, averaged, idxT + (size_t)b * safe_k,
                              distT + (size_t)b * safe_k, safe_k, p.ndist);
            // (the node's rows sit in the run's arena, other leaves right behind them: the padded slices meet in a buffer)
            double* rows = corr_.reserve((size_t)per * d_);
            copy_doubles(stream_, rows + (size_t)b * d_, right.data.p + (size_t)b * d_, (e - b) * (int64_t)d_);
            exchange(rows, per1 * d_ * (int64_t)sizeof(double));
            copy_doubles(stream_, right.data.p, rows, b * (int64_t)d_);
            copy_doubles(stream_, right.data.p + (size_t)e * d_, rows + (size_t)e * d_, (right.n - e) * (int64_t)d_);
        } else if (!p.var_adj) {
            tricube_apply(stream_, right.data.p, right.n, d_, averaged, idxT, distT, safe_k, p.ndist);
        } else {
            // mnnCorrect(var.adj=TRUE) on the fastMNN correction (R/mnnCorrect.R:331-342,462-481): every right cell's
            // correction vector is stretched so that the cell lands on the matching quantile of the left batch
            double* corr = corr_.reserve((size_t)right.n * d_);
            tricube_vectors(stream_, right.n, d_, averaged, idxT, distT, safe_k, p.ndist, corr);
            const int32_t* r1 = lrows;
            const int32_t* r2 = rrows;
            if (!r1) {
                int32_t* io = iota_l_.reserve(left.n);
                hipLaunchKernelGGL(iota_offset, dim3(cdiv(left.n, 256)), dim3(256), 0, stream_, io, left.n, 0);
                r1 = io;
            }
            if (!r2) {
                int32_t* io = iota_r_.reserve(right.n);
                hipLaunchKernelGGL(iota_offset, dim3(cdiv(right.n, 256)), dim3(256), 0, stream_, io, right.n, 0);
                r2 = io;
            }
            BMX_LAUNCH_CHECK();
            // The reference's loop is independent per right cell (src/adjust_shift_variance.cpp:51-161): each rank takes its
            // slice of them (the layout of the sharded searches, bmx_shard_range) and the scalings are all-gathered in place.
            // The form (exact / tiled) is decided from the WHOLE call's size, so every rank count gives the same numbers.
            AsvPlan plan = adjust_shift_variance_plan(d_, right.n, nLs, nRs, 1);
            int64_t cb = 0, ce = right.n;
            bmx_shard_range_impl(right.n, rank_, world_, &cb, &ce);
            const int64_t per_cells = bmx_shard_rows_per_rank(right.n, world_);
            double* ws = asv_ws_.reserve(plan.main_doubles + plan.extra_doubles);
            double* scaling = asv_scale_.reserve((size_t)per_cells * world_);
            sec.reset();  // (the variance adjustment is timed on its own: event tag 4, bmx_engine_profile_var_adj)
            std::pair<hipEvent_t, hipEvent_t> aev{nullptr, nullptr};
            if (knn_ws_.profile) {
                aev = knn_ws_.next_events(4);
                (void)hipEventRecord(aev.first, stream_);
            }
            // (testing hook "asv_modes": this merge's share of the tiled form's tallies, and the snapshot merge's per-cell ways --
            // both wait for the device, which a test may)
            const bool tallies = dev_knobs().asv_modes > 0 && !plan.exact;
            unsigned long long t0[3] = {0, 0, 0};
            if (tallies) asv_tally_read(t0, false);
            adjust_shift_variance_device(stream_, left.data.p, d_, left.n, right.data.p, right.n, corr, p.sigma, r1, nLs, r2,
                                         nRs, scaling, ws, plan, /* vect_row_major */ 1, (int)cb, (int)ce);
            if (aev.second) (void)hipEventRecord(aev.second, stream_);
            if (tallies) {
                unsigned long long t1[3] = {0, 0, 0};
                asv_tally_read(t1, false);
                for (int i = 0; i < 3; ++i) rec.asv_tally[i] = (int64_t)(t1[i] - t0[i]);
                if (mdx == snap_merge_) {
                    snap_modes_.assign((size_t)right.n, 255);
                    asv_modes_read(snap_modes_.data(), snap_modes_.size());
                }
            }
            asv_pairs_ += (double)(ce - cb) * ((double)nLs + (double)nRs);
            sec = std::make_unique<Section>(this);
            queued_work_s_ += 5e-8 * (double)(ce - cb) * ((double)nLs + (double)nRs);
            exchange(scaling, per_cells * (int64_t)sizeof(double));
            if (mdx == snap_merge_) {  // diagnostics: what adjust_shift_variance was handed at this merge, and what it returned
                snap_anl_ = left.n;
                snap_anr_ = right.n;
                snap_ar1_ = nLs;
                snap_ar2_ = nRs;
                auto keep = [&](DevBuf<double>& b, const double* src, size_t n) {
                    BMX_HIP(hipMemcpyAsync(b.reserve(n), src, n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
                };
                keep(snap_al_, left.data.p, (size_t)left.n * d_);
                keep(snap_ar_, right.data.p, (size_t)right.n * d_);
                keep(snap_ac_, corr, (size_t)right.n * d_);
                keep(snap_as_, scaling, (size_t)right.n);
                BMX_HIP(hipMemcpyAsync(snap_ai1_.reserve((size_t)std::max(nLs, 1)), r1, (size_t)nLs * sizeof(int32_t),
                                       hipMemcpyDeviceToDevice, stream_));
                BMX_HIP(hipMemcpyAsync(snap_ai2_.reserve((size_t)std::max(nRs, 1)), r2, (size_t)nRs * sizeof(int32_t),
                                       hipMemcpyDeviceToDevice, stream_));
            }
            add_scaled_rows(stream_, right.data.p, right.n, d_, corr, scaling);
        }
        right.stat_slot.assign(right.origin.size(), -1);  // the corrected cells moved
        ++n_extras_;
    } else {
        // skipped: the "new" variances are those of the (orthogonalised) data as it stands (R/fastMNN.R:500-501)
        ensure_stats(left);
        ensure_stats(right);
        rec.new_slot = left.stat_slot;
        rec.new_slot.insert(rec.new_slot.end(), right.stat_slot.begin(), right.stat_slot.end());
    }

    // UPDATE (R/fastMNN.R:520-525): rbind, combine restrict, concatenate origin / extras
    merged = std::make_unique<Node>();
    Node& m = *merged;
    m.index = left.index;
    m.index.insert(m.index.end(), right.index.begin(), right.index.end());
    m.n = left.n + right.n;
    if (left.data.view && right.data.view && left.data.p + (size_t)left.n * d_ == right.data.p) {
        // both children sit next to each other in the run's arena (leaves laid out in tree order): the merged node IS
        // that stretch of rows, rbind copies nothing
        m.data.alias(left.data.p, (size_t)m.n * d_);
    } else {
        double* md = m.data.reserve((size_t)m.n * d_);
        BMX_HIP(hipMemcpyAsync(md, left.data.p, (size_t)left.n * d_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
        BMX_HIP(hipMemcpyAsync(md + (size_t)left.n * d_, right.data.p, (size_t)right.n * d_ * sizeof(double),
                               hipMemcpyDeviceToDevice, stream_));
    }
    m.origin = left.origin;
    m.origin.insert(m.origin.end(), right.origin.begin(), right.origin.end());
    m.stat_slot = left.stat_slot;
    m.stat_slot.insert(m.stat_slot.end(), right.stat_slot.begin(), right.stat_slot.end());
    m.extras = left.extras;
    m.extras.insert(m.extras.end(), right.extras.begin(), right.extras.end());
    if (do_correct) m.extras.push_back(vid);
    if (left.has_restrict || right.has_restrict) {  // .combine_restrict (R/fastMNN.R:610-622)
        m.has_restrict = true;
        m.n_restrict = nLs + nRs;
        int32_t* mr = m.restrict_rows.reserve(m.n_restrict);
        if (left.has_restrict)
            BMX_HIP(hipMemcpyAsync(mr, left.restrict_rows.p, (size_t)nLs * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                   stream_));
        else
            hipLaunchKernelGGL(iota_offset, dim3(cdiv(nLs, 256)), dim3(256), 0, stream_, mr, nLs, 0);
        if (right.has_restrict)
            hipLaunchKernelGGL(add_offset_copy, dim3(cdiv(nRs, 256)), dim3(256), 0, stream_, right.restrict_rows.p,
                               nRs, left.n, mr + nLs);
        else
            hipLaunchKernelGGL(iota_offset, dim3(cdiv(nRs, 256)), dim3(256), 0, stream_, mr + nLs, nRs, left.n);
        BMX_LAUNCH_CHECK();
        if (left.restrict_dups || right.restrict_dups) {  // the chains of repeated cells: positions of the right part shift by nLs
            m.restrict_dups = true;
            int32_t* nx = m.dup_next.reserve(m.n_restrict);
            int32_t* hd = m.dup_head.reserve(m.n_restrict);
            hipLaunchKernelGGL(combine_dup_chains, dim3(cdiv(m.n_restrict, 256)), dim3(256), 0, stream_,
                               left.restrict_dups ? left.dup_next.p : nullptr, left.restrict_dups ? left.dup_head.p : nullptr, nLs,
                               right.restrict_dups ? right.dup_next.p : nullptr, right.restrict_dups ? right.dup_head.p : nullptr,
                               nRs, nx, hd);
            BMX_LAUNCH_CHECK();
        }
    }
    // the copies above read left / right data: the caller releases those nodes into this engine's block cache, whose
    // next user is ordered behind the copies by the stream
}

void Engine::run(const bmx_params_t& p, const int32_t* tree, int tree_len) {
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    check_alive();
    // Optimistic first: the searches leave the count of their uncertified queries on the device and sweep them without a
    // host round trip (knn.hip: search_tiers).  A search that cannot be completed that way (hundreds of uncertified queries,
    // lists overflowing with exact ties) raises a device flag the waits of the merge loop look at; the run then starts over
    // from the resident inputs with host-checked searches.  Nothing of a merge is visible outside before the run returns.
    // With several ranks the flag is agreed on: it travels with every search's lists (Engine::knn) and with the counts of the
    // searches a rank runs alone (gather_counts), so all ranks see it at the same wait and start over together.
    knn_ws_.optimistic = !knn_ws_.force_exact;
    solo_ = solo_flag_ = false;
    try {
        run_once(p, tree, tree_len);
    } catch (const OptimisticRetry&) {
        if (debug_prints()) fprintf(stderr, "[bmx] optimistic run gave up; repeating with host-checked searches\n");
        ++optimistic_retries_;
        knn_ws_.optimistic = false;
        run_once(p, tree, tree_len);
    }
    knn_ws_.optimistic = false;
}

// counts[t] is known on rank t % world only (0 elsewhere): every rank gets them all (an all-gather of the padded slices)
std::vector<int32_t> Engine::gather_counts(const std::vector<int32_t>& mine) {
    if (emu_mode_ == 2) throw Error(BMX_ERR_ARG, "the one-GPU emulation of a rank covers predefined merge trees, not auto-merge's dealt searches");
    if (world_ == 1) return mine;
    const int n = (int)mine.size();
    const int per = (n + world_ - 1) / world_ + 1;  // (+ 1: "one of my searches could not be completed optimistically")
    std::vector<int32_t> slice((size_t)per * world_, 0);
    for (int t = rank_; t < n; t += world_) slice[(size_t)rank_ * per + t / world_] = mine[t];
    slice[(size_t)rank_ * per + per - 1] = solo_flag_ ? 1 : 0;
    solo_flag_ = false;
    int32_t* dev = count_xchg_.reserve((size_t)per * world_);
    BMX_HIP(hipMemcpyAsync(dev, slice.data(), slice.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    BMX_HIP(hipStreamSynchronize(stream_));
    exchange(dev, (int64_t)per * (int64_t)sizeof(int32_t));
    BMX_HIP(hipMemcpyAsync(slice.data(), dev, slice.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    wait();
    for (int r = 0; r < world_; ++r)
        if (slice[(size_t)r * per + per - 1] != 0) throw OptimisticRetry();  // every rank sees the same slices: all restart here
    std::vector<int32_t> out((size_t)n);
    for (int t = 0; t < n; ++t) out[t] = slice[(size_t)(t % world_) * per + t / world_];
    return out;
}

std::vector<int32_t> Engine::solo_counts(int n, const std::function<int(int)>& count, bool dry) {
    std::vector<int32_t> mine((size_t)n, 0);
    if (dry) return mine;
    const int rk = rank_, wd = world_;
    for (int t = 0; t < n; ++t) {
        if (t % wd != rk) continue;
        rank_ = 0;
        world_ = 1;  // an unsharded search on this rank alone
        solo_ = wd > 1;
        try {
            mine[t] = count(t);
        } catch (...) {
            rank_ = rk;
            world_ = wd;
            solo_ = false;
            throw;
        }
        rank_ = rk;
        world_ = wd;
        solo_ = false;
    }
    return gather_counts(mine);
}

void Engine::run_once(const bmx_params_t& p, const int32_t* tree, int tree_len) {
    if (B_ < 2) throw Error(BMX_ERR_ARG, "at least two batches must be specified");
    if (p.k < 1) throw Error(BMX_ERR_ARG, "'k' must be positive");
    queued_work_s_ = 0.0;
    const int nmerges = B_ - 1;
    root_.reset();  // a run that fails half-way leaves nothing to download
    pairs_pinned_ = false;
    merges_.clear();
    merges_.resize(nmerges);
    n_extras_ = 0;
    xchg_calls_ = xchg_bytes_ = 0;
    emu_next_ = 0;
    knn_ws_.events_used = 0;
    knn_ws_.replay_idx = 0;
    asv_pairs_ = 0.0;
    vecs_.reserve((size_t)(2 * B_ + 8) * d_);
    // statistics slots (column means [d] + total variance) and batch.size scalars: a merge takes at most one per
    // segment before and after its centring, plus one
    slot_cap_ = (2 * B_ + 2) * B_ + 8;
    n_slots_ = 0;
    scal_.reserve(slot_cap_);
    means_pool_.reserve((size_t)slot_cap_ * d_);
    BMX_HIP(hipMemsetAsync(scal_.p, 0, (size_t)slot_cap_ * sizeof(double), stream_));
    BMX_HIP(hipMemsetAsync(knn_ws_.opt_state_ptr(stream_), 0, 8 * sizeof(int32_t), stream_));
    scal_host_.assign(slot_cap_, 0.0);
    knn_ws_.exact_total = knn_ws_.tier2_total = 0;

    // leaves: row-major working copies of the resident inputs
    std::vector<TreeSlot> slots;
    // Predefined tree: the leaves go into ONE arena in the order the post-order code names them (= left to right in the
    // tree), so the two children of every merge are adjacent row ranges and the merged node is their union in place.
    // Auto-merge picks pairs as it goes: its leaves own their buffers and a merge copies.
    double* arena = nullptr;
    int64_t arena_rows = 0;
    if (!p.auto_merge) arena = arena_.reserve((size_t)N_ * d_);
    auto make_leaf = [&](int b, int64_t arena_row) {
        auto n = std::make_unique<Node>();
        n->index = {b + 1};
        n->n = nrows_[b];
        n->origin = {Segment{b + 1, nrows_[b]}};
        double* dp;
        if (arena) {
            n->data.alias(arena + (size_t)arena_row * d_, (size_t)n->n * d_);
            dp = n->data.p;
        } else {
            dp = n->data.reserve((size_t)n->n * d_);
        }
        if (lazy_) {  // the batch may still be on its way: the transpose queues behind its copy
            if ((size_t)b >= host_data_.size() && !uploaded_[b])
                throw Error(BMX_ERR_ARG, "internal: a lazily uploaded batch has lost its host matrix");
            ensure_uploaded(b);
            BMX_HIP(hipStreamWaitEvent(stream_, up_ev_[b], 0));
        }
        transpose_cm_to_rm(stream_, inputs_cm_[b].p, n->n, d_, dp);
        if (n_restrict_[b] >= 0) {
            n->has_restrict = true;
            n->n_restrict = n_restrict_[b];
            int32_t* r = n->restrict_rows.reserve(n->n_restrict);
            BMX_HIP(hipMemcpyAsync(r, inputs_restrict_[b].p, (size_t)n->n_restrict * sizeof(int32_t),
                                   hipMemcpyDeviceToDevice, stream_));
            if (has_dups_[b]) {
                n->restrict_dups = true;
                const size_t m = (size_t)n->n_restrict;
                BMX_HIP(hipMemcpyAsync(n->dup_next.reserve(m), inputs_dup_[b].p, m * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                       stream_));
                BMX_HIP(hipMemcpyAsync(n->dup_head.reserve(m), inputs_dup_[b].p + m, m * sizeof(int32_t),
                                       hipMemcpyDeviceToDevice, stream_));
            }
        }
        return n;
    };

    if (!p.auto_merge) {
        // rebuild the binary tree from its post-order encoding and validate the leaves (R/MNN_tree.R:96-105)
        std::vector<int> stack;
        std::vector<char> seen(B_ + 1, 0);
        for (int i = 0; i < tree_len; ++i) {
            const int v = tree[i];
            if (v == 0) {
                if (stack.size() < 2) throw Error(BMX_ERR_TREE, "merge tree structure should contain two children per node");
                TreeSlot s;
                s.right = stack.back();
                stack.pop_back();
                s.left = stack.back();
                stack.pop_back();
                slots.push_back(std::move(s));
                stack.push_back((int)slots.size() - 1);
            } else {
                if (v < 1 || v > B_ || seen[v]) throw Error(BMX_ERR_TREE, "invalid leaf nodes specified in 'merge.order'");
                seen[v] = 1;
                TreeSlot s;  // materialised when its merge comes up (a lazily uploaded batch arrives in the meantime)
                s.batch = v - 1;
                s.arena_row = arena_rows;
                arena_rows += nrows_[v - 1];
                slots.push_back(std::move(s));
                stack.push_back((int)slots.size() - 1);
            }
        }
        int nleaves = 0;
        for (int b = 1; b <= B_; ++b) nleaves += seen[b];
        if (stack.size() != 1 || nleaves != B_) throw Error(BMX_ERR_TREE, "invalid leaf nodes specified in 'merge.order'");
        const int root = stack.back();
        // .get_next_merge (R/MNN_tree.R:61-69): both children finished -> merge; else descend into child 2 if it is still
        // a list, otherwise into child 1
        auto next_merge = [&](const std::vector<char>& done) {
            int cur = root;
            for (;;) {
                const TreeSlot& s = slots[cur];
                const bool l_leaf = done[s.left], r_leaf = done[s.right];
                if (l_leaf && r_leaf) return cur;
                cur = !r_leaf ? s.right : s.left;
            }
        };
        std::vector<char> done(slots.size(), 0);
        for (size_t i = 0; i < slots.size(); ++i) done[i] = slots[i].ready();
        if (lazy_) {  // the order in which the merges will ask for the batches: what to prefetch next
            std::vector<char> sim = done;
            need_order_.clear();
            need_pos_ = 0;
            for (int mdx = 0; mdx < nmerges; ++mdx) {
                const int cur = next_merge(sim);
                for (int child : {slots[cur].left, slots[cur].right})
                    if (slots[child].batch >= 0) need_order_.push_back(slots[child].batch);
                sim[cur] = 1;
            }
        }
        auto materialise = [&](TreeSlot& t) {
            if (!t.node) {
                t.node = make_leaf(t.batch, t.arena_row);
                t.batch = -1;
            }
        };
        const auto t_run = std::chrono::steady_clock::now();
        for (int mdx = 0; mdx < nmerges; ++mdx) {
            const int cur = next_merge(done);
            TreeSlot& s = slots[cur];
            materialise(slots[s.left]);
            materialise(slots[s.right]);
            std::unique_ptr<Node> merged;
            if (debug_timings())  // (host clock: when the host got to queue this merge)
                fprintf(stderr, "[bmx]   merge %d queued from %.2f ms\n", mdx + 1,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count());
            merge_step(mdx, *slots[s.left].node, *slots[s.right].node, p, merged);
            slots[s.left].node.reset();
            slots[s.right].node.reset();
            s.node = std::move(merged);  // .update_tree (R/MNN_tree.R:71-77)
            done[cur] = 1;
        }
        root_ = std::move(slots[root].node);
    } else {
        // auto-merge (R/MNN_tree.R:154-226)
        std::vector<std::unique_ptr<Node>> rem;
        for (int b = 0; b < B_; ++b) rem.push_back(make_leaf(b, 0));
        std::vector<std::vector<int64_t>> stats(B_, std::vector<int64_t>(B_, 0));
        {
            // .initialize_auto_search (R/MNN_tree.R:160-164): B (B - 1) / 2 independent counts.  With several ranks they are
            // dealt round robin -- each a whole, unsharded search on its rank: no gathers inside, all ranks busy -- and the
            // numbers all-gathered at the end, instead of every rank walking every count with row-split searches
            std::vector<std::pair<int, int>> todo;
            for (int i = 0; i < B_; ++i)
                for (int j = 0; j < i; ++j) todo.emplace_back(i, j);
            std::vector<int32_t> counts = solo_counts((int)todo.size(), [&](int t) {
                return count_mnn_pairs(*rem[todo[t].first], *rem[todo[t].second], p);
            });
            for (size_t t = 0; t < todo.size(); ++t) stats[todo[t].first][todo[t].second] = counts[t];
        }
        for (int mdx = 0; mdx < nmerges; ++mdx) {
            // .pick_best_merge: first maximum in column-major order; left = row, right = column
            const int R = (int)rem.size();
            int64_t best = -1;
            int bi = 0, bj = 0;
            for (int j = 0; j < R; ++j)
                for (int i = 0; i < R; ++i)
                    if (stats[i][j] > best) {
                        best = stats[i][j];
                        bi = i;
                        bj = j;
                    }
            if (best <= 0) throw Error(BMX_ERR_NO_PAIRS, "no mutual nearest neighbours found between batches");
            std::unique_ptr<Node> merged;
            merge_step(mdx, *rem[bi], *rem[bj], p, merged);
            // .update_remainders: drop both, recount the new node against every remaining one
            std::vector<std::unique_ptr<Node>> nrem;
            std::vector<int> keep;
            for (int i = 0; i < R; ++i)
                if (i != bi && i != bj) {
                    keep.push_back(i);
                    nrem.push_back(std::move(rem[i]));
                }
            const int K = (int)keep.size();
            std::vector<std::vector<int64_t>> ns(K + 1, std::vector<int64_t>(K + 1, 0));
            for (int a = 0; a < K; ++a)
                for (int c = 0; c < K; ++c) ns[a][c] = stats[keep[a]][keep[c]];
            if (K > 0) {
                // upstream keeps orthogonalising the SAME left copy across j (R/MNN_tree.R:185-186)
                // (.update_remainders, R/MNN_tree.R:205-226: the K counts are dealt over the ranks like the initial ones; the
                // orthogonalisations of the shared left copy are cheap row passes every rank applies in order)
                std::unique_ptr<Node> lcopy = clone_node(*merged);
                std::vector<int32_t> mine((size_t)K, 0);
                for (int c = 0; c < K; ++c) {
                    orthogonalize(*lcopy, nrem[c]->extras);
                    if (c % world_ != rank_) continue;
                    const Node* r = nrem[c].get();
                    std::unique_ptr<Node> rc;
                    if (!merged->extras.empty()) {
                        rc = clone_node(*nrem[c]);
                        orthogonalize(*rc, merged->extras);
                        r = rc.get();
                    }
                    const int rk = rank_, wd = world_;
                    rank_ = 0;
                    world_ = 1;  // an unsharded search on this rank alone
                    solo_ = wd > 1;
                    try {
                        mine[c] = (int32_t)find_mnn(*lcopy, *r, p.k, p.prop_k).P;
                    } catch (...) {
                        rank_ = rk;
                        world_ = wd;
                        solo_ = false;
                        throw;
                    }
                    rank_ = rk;
                    world_ = wd;
                    solo_ = false;
                }
                const std::vector<int32_t> all = gather_counts(mine);
                for (int c = 0; c < K; ++c) ns[K][c] = all[c];
            }
            nrem.push_back(std::move(merged));
            rem = std::move(nrem);
            stats = std::move(ns);
        }
        root_ = std::move(rem[0]);
    }
    const size_t scal_bytes = scal_host_.size() * sizeof(double);
    if (scal_bytes > scal_pin_bytes_) {  // (more than ~60 batches: a block of its own instead of a pooled 64 KiB one)
        if (scal_pin_ && scal_pin_bytes_ > KnnWorkspace::kPinnedSmall) (void)hipHostFree(scal_pin_);
        else KnnWorkspace::pinned_small_give(scal_pin_);
        scal_pin_ = nullptr;
        if (scal_bytes <= KnnWorkspace::kPinnedSmall) {
            scal_pin_ = static_cast<double*>(KnnWorkspace::pinned_small_take());
            scal_pin_bytes_ = KnnWorkspace::kPinnedSmall;
        } else {
            BMX_HIP(hipHostMalloc((void**)&scal_pin_, scal_bytes, hipHostMallocDefault));
            scal_pin_bytes_ = scal_bytes;
        }
    }
    BMX_HIP(hipMemcpyAsync(scal_pin_, scal_.p, scal_host_.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
    // (with var_adj the last merge's adjust_shift_variance is still running: its budget is n2 (nr1 + nr2) pair visits)
    const int32_t* fin = read_state();  // (the last tricube search's flag is looked at here)
    knn_ws_.exact_total += fin[1];
    std::memcpy(scal_host_.data(), scal_pin_, scal_host_.size() * sizeof(double));
    if (lazy_) {  // every batch is resident now: later runs need nothing from the caller
        for (int b = 0; b < B_; ++b) ensure_uploaded(b);
        BMX_HIP(hipStreamSynchronize(copy_stream_));
        host_data_.clear();
        lazy_ = false;
    }
    fallbacks_ = knn_ws_.exact_total;
    stage_pairs();
}

void Engine::download(double* corrected, int32_t* batch, int32_t* merge_left, int32_t* merge_right,
                      double* batch_size, int32_t* skipped, double* lost_var) {
    check_alive();
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    if (!root_) throw Error(BMX_ERR_ARG, "no finished run to download");
    const int nmerges = B_ - 1;
    if (corrected) {
        // rows back in input batch order (R/fastMNN.R:541-547): batch b starts at the sum of earlier batch sizes
        std::vector<int64_t> start(B_ + 1, 0);
        for (int b = 0; b < B_; ++b) start[b + 1] = start[b] + nrows_[b];
        DevBuf<double> out_cm;
        double* oc = out_cm.reserve((size_t)N_ * d_);
        int r0 = 0;
        for (const Segment& s : root_->origin) {
            transpose_rm_to_cm(stream_, root_->data.p + (size_t)r0 * d_, s.n, d_, oc, (int)N_, (int)start[s.batch - 1]);
            r0 += s.n;
        }
        download_pageable(corrected, oc, (size_t)N_ * d_ * sizeof(double), stream_);
    }
    if (batch) {
        int64_t o = 0;
        for (int b = 0; b < B_; ++b)
            for (int i = 0; i < nrows_[b]; ++i) batch[o++] = b + 1;
    }
    for (int m = 0; m < nmerges; ++m) {
        const MergeRecord& rec = merges_[m];
        if (merge_left)
            for (int j = 0; j < B_; ++j) merge_left[(size_t)m * B_ + j] = j < (int)rec.left_set.size() ? rec.left_set[j] : 0;
        if (merge_right)
            for (int j = 0; j < B_; ++j)
                merge_right[(size_t)m * B_ + j] = j < (int)rec.right_set.size() ? rec.right_set[j] : 0;
        if (batch_size)
            batch_size[m] = rec.batch_size_na || rec.bs_slot < 0 ? std::numeric_limits<double>::quiet_NaN()
                                                                  : scal_host_[rec.bs_slot];
        if (skipped) skipped[m] = rec.skipped ? 1 : 0;
        if (lost_var) {
            for (int b = 0; b < B_; ++b) lost_var[(size_t)b * nmerges + m] = 0.0;  // 1 - var.kept, var.kept starts at 1
            for (size_t s = 0; s < rec.var_batches.size(); ++s) {
                const double oldv = scal_host_[rec.old_slot[s]], newv = scal_host_[rec.new_slot[s]];
                lost_var[(size_t)(rec.var_batches[s] - 1) * nmerges + m] = 1.0 - newv / oldv;
            }
        }
    }
}

namespace {
// node position (1-based) -> output row (1-based), R/fastMNN.R:533-547: a node is a run of whole batches; tab[i] = first
// position of its i-th batch inside the node (tab[ns] = its size), tab[ns + 1 + i] = that batch's first row in the input
// batch order
__global__ void remap_pairs_tab(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ tab, int ns,
                                int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int pos = in[i] - 1;
    int sgi = 0;
    for (int t = 1; t < ns; ++t) sgi += pos >= tab[t] ? 1 : 0;
    out[i] = tab[ns + 1 + sgi] + (pos - tab[sgi]) + 1;
}
}  // namespace

int64_t Engine::pairs_count(int merge) const {
    check_alive();
    if (!root_) throw Error(BMX_ERR_ARG, "no finished run to download");
    if (merge < 0 || merge >= (int)merges_.size()) throw Error(BMX_ERR_ARG, "merge index out of range");
    return merges_[merge].npairs;
}

void Engine::stage_pairs() {
    // shift every merge's pairs to their nodes' places in the final merged order, then to the input batch order
    // (R/fastMNN.R:533-547): a short table per side, applied on the device; all merges into one block
    pairs_pinned_ = false;
    const int nmerges = (int)merges_.size();
    pairs_off_.assign(nmerges + 1, 0);
    for (int m = 0; m < nmerges; ++m) pairs_off_[m + 1] = pairs_off_[m] + 2 * merges_[m].npairs;
    const int64_t total = pairs_off_[nmerges];
    if (total == 0) return;
    std::vector<int64_t> in_start(B_ + 1, 0);
    for (int b = 0; b < B_; ++b) in_start[b + 1] = in_start[b] + nrows_[b];
    std::vector<size_t> tab_off(nmerges, 0);
    pairs_tab_host_.clear();
    auto fill = [&](const std::vector<int>& set) {
        const int ns = (int)set.size();
        const size_t base = pairs_tab_host_.size();
        pairs_tab_host_.resize(base + 2 * ns + 1);
        int32_t* t = pairs_tab_host_.data() + base;
        int64_t o = 0;
        for (int i = 0; i < ns; ++i) {
            t[i] = (int32_t)o;
            t[ns + 1 + i] = (int32_t)in_start[set[i] - 1];
            o += nrows_[set[i] - 1];
        }
        t[ns] = (int32_t)o;
    };
    for (int m = 0; m < nmerges; ++m) {
        tab_off[m] = pairs_tab_host_.size();
        fill(merges_[m].left_set);
        fill(merges_[m].right_set);
    }
    // (the previous run's lists may still be on their way out of pairs_all_)
    if (pairs_copy_pending_ && pairs_ev_) BMX_HIP(hipStreamWaitEvent(stream_, pairs_ev_, 0));
    pairs_copy_pending_ = false;
    int32_t* dt = pairs_tab_.reserve(pairs_tab_host_.size());
    BMX_HIP(hipMemcpyAsync(dt, pairs_tab_host_.data(), pairs_tab_host_.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    int32_t* all = pairs_all_.reserve((size_t)total);
    for (int m = 0; m < nmerges; ++m) {
        const MergeRecord& rec = merges_[m];
        const int64_t P = rec.npairs;
        if (P == 0) continue;
        const int nl = (int)rec.left_set.size(), nr = (int)rec.right_set.size();
        hipLaunchKernelGGL(remap_pairs_tab, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, stream_, (const int32_t*)rec.first.p, P,
                           (const int32_t*)(dt + tab_off[m]), nl, all + pairs_off_[m]);
        hipLaunchKernelGGL(remap_pairs_tab, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, stream_, (const int32_t*)rec.second.p, P,
                           (const int32_t*)(dt + tab_off[m] + 2 * nl + 1), nr, all + pairs_off_[m] + P);
    }
    BMX_LAUNCH_CHECK();
    const size_t bytes = (size_t)total * sizeof(int32_t);
    if (bytes <= ((size_t)64 << 20)) {  // (beyond that the lists stay on the device and go out through the staging ring)
        if (bytes > pairs_pin_bytes_) {
            if (pairs_ev_ && pairs_pin_) guarded_event_sync(pairs_ev_);  // (an earlier run's copy into the block that goes away)
            PinnedBlocks::give(PinnedBlocks::Blk{pairs_pin_, pairs_pin_bytes_});
            const PinnedBlocks::Blk b = PinnedBlocks::take(bytes);
            pairs_pin_ = b.p;
            pairs_pin_bytes_ = b.bytes;
        }
        // on the copy stream (a DMA engine), behind the remap kernels: the engine's stream is free for the next run
        if (!pairs_ev_) BMX_HIP(hipEventCreateWithFlags(&pairs_ev_, hipEventDisableTiming));
        if (!pairs_ready_ev_) BMX_HIP(hipEventCreateWithFlags(&pairs_ready_ev_, hipEventDisableTiming));
        if (!copy_stream_) copy_stream_ = StreamPool::take();
        BMX_HIP(hipEventRecord(pairs_ready_ev_, stream_));
        BMX_HIP(hipStreamWaitEvent(copy_stream_, pairs_ready_ev_, 0));
        BMX_HIP(hipMemcpyAsync(pairs_pin_, all, bytes, hipMemcpyDeviceToHost, copy_stream_));
        BMX_HIP(hipEventRecord(pairs_ev_, copy_stream_));
        pairs_pinned_ = true;
        pairs_copy_pending_ = true;
    }
}

void Engine::pairs_into(int merge, int32_t* left, int32_t* right) {
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    const int64_t P = pairs_count(merge);
    if (P == 0) return;
    if (pairs_pinned_) {  // the whole run's lists were sent to pinned memory when the run ended: a host copy
        guarded_event_sync(pairs_ev_);
        const int32_t* src = static_cast<const int32_t*>(pairs_pin_) + pairs_off_[merge];
        host_parallel_memcpy(left, src, (size_t)P * sizeof(int32_t));
        host_parallel_memcpy(right, src + P, (size_t)P * sizeof(int32_t));
        return;
    }
    download_pageable(left, pairs_all_.p + pairs_off_[merge], (size_t)P * sizeof(int32_t), stream_);
    download_pageable(right, pairs_all_.p + pairs_off_[merge] + P, (size_t)P * sizeof(int32_t), stream_);
}

void Engine::pairs_all_into(int nmerges, int32_t* const* left, int32_t* const* right, const int64_t* capacity) {
    check_alive();
    if (!root_) throw Error(BMX_ERR_ARG, "no finished run to download");
    if (nmerges != (int)merges_.size()) throw Error(BMX_ERR_ARG, "bmx_engine_pairs_all_into: the last run had another number of merges");
    for (int m = 0; m < nmerges; ++m) {
        if (capacity[m] < merges_[m].npairs) throw Error(BMX_ERR_ARG, "bmx_engine_pairs_all_into: an array is too short");
        if (merges_[m].npairs > 0 && (!left[m] || !right[m])) throw Error(BMX_ERR_ARG, "bmx_engine_pairs_all_into: null array");
    }
    if (!pairs_pinned_) {  // (lists beyond the pinned block's bound: through the staging ring, one by one)
        for (int m = 0; m < nmerges; ++m) pairs_into(m, left[m], right[m]);
        return;
    }
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    guarded_event_sync(pairs_ev_);
    // the whole run's lists sit in pinned memory: ONE job of the host threads over all of them, in pieces of 64 KB (the
    // first touch of the caller's fresh pages spreads over the threads like the bytes do)
    struct Piece {
        char* dst;
        const char* src;
        size_t bytes;
    };
    constexpr size_t kPiece = (size_t)64 << 10;
    // pieces in round-robin order over the 2 x nmerges arrays: threads that take consecutive pieces then fault pages of
    // DIFFERENT arrays (consecutive pieces of one ~2 MB array share a page-table page and its lock: measured slower than
    // the one-array-at-a-time calls this replaces)
    std::vector<Piece> pieces;
    size_t longest = 0;
    for (int m = 0; m < nmerges; ++m) longest = std::max(longest, (size_t)merges_[m].npairs * sizeof(int32_t));
    for (size_t o = 0; o < longest; o += kPiece)
        for (int m = 0; m < nmerges; ++m) {
            const size_t bytes = (size_t)merges_[m].npairs * sizeof(int32_t);
            if (o >= bytes) continue;
            const char* src = reinterpret_cast<const char*>(static_cast<const int32_t*>(pairs_pin_) + pairs_off_[m]);
            for (int side = 0; side < 2; ++side) {
                char* dst = reinterpret_cast<char*>(side == 0 ? left[m] : right[m]);
                pieces.push_back({dst + o, src + side * bytes + o, std::min(kPiece, bytes - o)});
            }
        }
    HostPool::get().parallel_for(pieces.size(), [&](size_t i) { std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes); });
}

void Engine::pairs(int merge, int32_t** left, int32_t** right, int64_t* npairs) {
    const int64_t P = pairs_count(merge);
    int32_t* L = (int32_t*)std::malloc(std::max<size_t>(1, (size_t)P) * sizeof(int32_t));
    int32_t* R = (int32_t*)std::malloc(std::max<size_t>(1, (size_t)P) * sizeof(int32_t));
    if (!L || !R) {
        std::free(L);
        std::free(R);
        throw std::bad_alloc();
    }
    try {
        pairs_into(merge, L, R);
    } catch (...) {
        std::free(L);
        std::free(R);
        throw;
    }
    *left = L;
    *right = R;
    *npairs = P;
}

void Engine::snapshot(double* left_rm, double* right_rm, int64_t* nl, int64_t* nr) {
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    if (snap_merge_ < 0 || !root_ || !snap_l_.p) throw Error(BMX_ERR_ARG, "no snapshot was taken");
    if (nl) *nl = snap_nl_;
    if (nr) *nr = snap_nr_;
    if (left_rm)
        BMX_HIP(hipMemcpyAsync(left_rm, snap_l_.p, (size_t)snap_nl_ * d_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    if (right_rm)
        BMX_HIP(hipMemcpyAsync(right_rm, snap_r_.p, (size_t)snap_nr_ * d_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    BMX_HIP(hipStreamSynchronize(stream_));
}

void Engine::snapshot_var_adj(double* left_rm, double* right_rm, double* corr_rm, double* scaling, int32_t* r1,
                              int32_t* r2, int64_t* sizes4) {
    CacheScope cache_scope(&cache_);
    BMX_HIP(hipSetDevice(device_));
    if (snap_merge_ < 0 || !root_ || !snap_as_.p) throw Error(BMX_ERR_ARG, "no snapshot of a variance adjustment was taken");
    if (sizes4) {
        sizes4[0] = snap_anl_;
        sizes4[1] = snap_anr_;
        sizes4[2] = snap_ar1_;
        sizes4[3] = snap_ar2_;
    }
    auto out = [&](void* dst, const void* src, size_t bytes) {
        if (dst && bytes) BMX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream_));
    };
    out(left_rm, snap_al_.p, (size_t)snap_anl_ * d_ * sizeof(double));
    out(right_rm, snap_ar_.p, (size_t)snap_anr_ * d_ * sizeof(double));
    out(corr_rm, snap_ac_.p, (size_t)snap_anr_ * d_ * sizeof(double));
    out(scaling, snap_as_.p, (size_t)snap_anr_ * sizeof(double));
    out(r1, snap_ai1_.p, (size_t)snap_ar1_ * sizeof(int32_t));
    out(r2, snap_ai2_.p, (size_t)snap_ar2_ * sizeof(int32_t));
    BMX_HIP(hipStreamSynchronize(stream_));
}

void Engine::var_adj_tally(int merge, int64_t* out3) const {
    if (merge < 0 || merge >= (int)merges_.size()) throw Error(BMX_ERR_ARG, "merge index out of range");
    for (int i = 0; i < 3; ++i) out3[i] = merges_[merge].asv_tally[i];
}

void Engine::snapshot_var_adj_modes(unsigned char* dst, int64_t n) const {
    for (int64_t i = 0; i < n; ++i) dst[i] = (size_t)i < snap_modes_.size() ? snap_modes_[(size_t)i] : 255;
}

void Engine::merge_stats(int merge, int64_t* out6) const {
    if (merge < 0 || merge >= (int)merges_.size()) throw Error(BMX_ERR_ARG, "merge index out of range");
    for (int i = 0; i < 6; ++i) out6[i] = merges_[merge].stats[i];
}

Engine::Section::Section(Engine* eng) : e(eng) {
    if (e->knn_ws_.profile) {
        ev = e->knn_ws_.next_events(3);
        (void)hipEventRecord(ev.first, e->stream_);
    }
}
Engine::Section::~Section() {
    if (ev.second) (void)hipEventRecord(ev.second, e->stream_);
}

void Engine::profile_detail(double* out10) {
    for (int i = 0; i < 10; ++i) out10[i] = 0.0;
    for (size_t i = 0; i < knn_ws_.events_used; ++i) {
        float t = 0.f;
        BMX_HIP(hipEventElapsedTime(&t, knn_ws_.events[i].first, knn_ws_.events[i].second));
        switch (knn_ws_.event_tag[i]) {
            case 1: out10[0] += t; out10[1] += 1; break;
            case 2: out10[2] += t; out10[3] += 1; break;
            case 0: out10[4] += t; out10[5] += 1; break;
            case 4: break;  // (adjust_shift_variance: profile_var_adj)
            default: out10[6] += t; break;
        }
    }
    out10[7] = (double)fallbacks_;
    out10[8] = (double)knn_ws_.tier2_total;
    out10[9] = (double)optimistic_retries_;
}

void Engine::profile_var_adj(double* out3) {
    out3[0] = out3[1] = 0.0;
    for (size_t i = 0; i < knn_ws_.events_used; ++i) {
        if (knn_ws_.event_tag[i] != 4) continue;
        float t = 0.f;
        BMX_HIP(hipEventElapsedTime(&t, knn_ws_.events[i].first, knn_ws_.events[i].second));
        out3[0] += t;
        out3[1] += 1;
    }
    out3[2] = asv_pairs_;
}

void Engine::profile(double* topk_ms, int64_t* launches, int64_t* fallbacks) {
    double ms = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i < knn_ws_.events_used; ++i) {
        if (knn_ws_.event_tag[i] >= 3) continue;  // a streaming section / a variance adjustment, not a candidate-pass launch
        float t = 0.f;
        BMX_HIP(hipEventElapsedTime(&t, knn_ws_.events[i].first, knn_ws_.events[i].second));
        ms += t;
        ++n;
    }
    if (topk_ms) *topk_ms = ms;
    if (launches) *launches = n;
    if (fallbacks) *fallbacks = fallbacks_;
}

}  // namespace bmx
