// Shared host-side helpers for the MI355X fastMNN hot path (gfx950 only; no CUDA paths).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/batchelor_mi355x.h"

namespace bmx {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define BMX_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            throw bmx::Error(BMX_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(e_));    \
    } while (0)

#define BMX_LAUNCH_CHECK() BMX_HIP(hipGetLastError())

// The candidate kernels hand data between waves through LDS words polled in unbounded loops (a bound that ends in
// s_trap doubled their run time): a protocol bug or a hardware fault would show up as a kernel that never
// ends.  So the HOST never waits without a deadline: every wait on the engine's stream polls hipStreamQuery against a
// budget scaled from the work that was queued, and gives up with this error.  The engine that sees it marks itself
// dead (its stream cannot be trusted any more; only a fresh process gets the GPU back).
struct WatchdogTimeout : Error {
    explicit WatchdogTimeout(const std::string& m) : Error(BMX_ERR_HIP, m) {}
};
// budget_s <= 0: plain hipStreamSynchronize
void guarded_stream_sync(hipStream_t stream, double budget_s);

// candidates the fp16 tier keeps per query and reference range for k <= 20 (A/B builds: make VARIANT=x EXTRA=-DBMX_KS1=24)
#ifndef BMX_KS1
#define BMX_KS1 32
#endif

// Testing hooks (bmx_dev_set in the C ABI): process-wide knobs that tests and developer scripts set by an explicit call.
// Nothing in the ENVIRONMENT of the host process changes what the library computes or which tier / form runs
// (BMX_DEBUG=1 only prints, BMX_HOST_THREADS only sizes the staging thread pool).
struct DevKnobs {
    int knn_tier = 0;         // 1 / 2: that candidate tier only, 3: exact FP64 scan only
    int sample = -1;          // rows of the threshold sample of a candidate pass (-1: automatic)
    int split_c = 0;          // reference ranges of the tail query blocks (0: automatic)
    int force_c = 0;          // reference ranges of EVERY query block (0: off)
    int no_margin = 0;        // fp16 tier: lists cut at their KS-th best only (round 2's rule)
    int asv_fast = 0;         // adjust_shift_variance: the tiled form whatever the size
    int asv_cap = -1;         // tiled form: kept addends per chain of the literal re-run (-1: default, 0: no re-run)
    int asv_modes = 0;        // tiled form: record which way each of the first n cells of a call went (bmx_dev_get_bytes)
    int asv_sync = 0;         // tiled form: 1 = the workgroups start each round of tiles together (measured slower twice: rounds 4 and 6)
    int tau_replay = 0;       // developer experiment: 1 record every search's final thresholds, 2 start the full passes from them
    int lk_seed = 1;          // k beyond the tiers' lists: partitions after the first searched within the first's kp-th distance (0: plainly)
    int sample_split = -1;    // ranges of the threshold sample of a search with few query blocks (-1: automatic, 0: never, n: that many)
    int exchange_always = 0;  // a single rank goes through its exchange transport too (an all-gather of one)
    int refine_wave = 0;      // the exact re-rank takes a whole wave for every query (no half-wave form)
};
DevKnobs& dev_knobs();
bool debug_prints();   // BMX_DEBUG=1 in the environment (read once)
bool debug_timings();  // BMX_DEBUG=t or 1

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// Released device blocks are parked (a few, bounded) and handed out again: a merge engine rebuilds its node buffers at
// every merge, and hipMalloc / hipFree cost hundreds of microseconds each (hipFree also waits for the device).
// A cache belongs to ONE engine (one device, one stream): every user of a block is ordered by that engine's stream,
// so reuse needs no extra synchronisation, and a block never crosses devices or streams.  The engine's public
// methods install their cache for the calling thread (CacheScope); a DevBuf released or grown outside any scope
// goes straight to hipFree / hipMalloc.
struct DevBlockCache {
    struct Block {
        void* p;
        size_t bytes;
        int device;
    };
    std::vector<Block> blocks;
    bool leak = false;  // the owner's stream is stuck (watchdog): blocks are abandoned, hipFree would wait for ever
    DevBlockCache() = default;
    DevBlockCache(const DevBlockCache&) = delete;
    DevBlockCache& operator=(const DevBlockCache&) = delete;
    // Blocks of an engine that goes away are parked in a process-wide pool (bounded) for the next engine on the same
    // device: a host that calls fastMNN() again and again (one engine per call at the .Call boundary) then pays the
    // dozens of hipMallocs of the workspaces once.  The owner has drained its stream before its cache is destroyed, so a
    // parked block is idle.
    struct GlobalPool {
        std::mutex mu;
        std::vector<Block> blocks;
        size_t total = 0;
    };
    static GlobalPool& global() {
        static GlobalPool* g = new GlobalPool();  // never destroyed: no hipFree after the runtime is torn down
        return *g;
    }
    ~DevBlockCache() {
        GlobalPool& g = global();
        if (!leak) {
            std::lock_guard<std::mutex> lk(g.mu);
            for (const Block& b : blocks) {
                if (g.blocks.size() < 256 && g.total + b.bytes <= ((size_t)16 << 30)) {
                    g.blocks.push_back(b);
                    g.total += b.bytes;
                } else {
                    (void)hipFree(b.p);
                }
            }
        }
        blocks.clear();
        if (current() == this) current() = nullptr;
    }
    static DevBlockCache*& current() {
        static thread_local DevBlockCache* c = nullptr;
        return c;
    }
    static int best_fit(const std::vector<Block>& v, size_t bytes, int device) {
        int best = -1;
        for (int i = 0; i < (int)v.size(); ++i)
            if (v[i].device == device && v[i].bytes >= bytes && v[i].bytes <= 2 * bytes + (1u << 20) &&
                (best < 0 || v[i].bytes < v[best].bytes))
                best = i;
        return best;
    }
    void* take(size_t bytes, size_t* got) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        int best = best_fit(blocks, bytes, dev);
        if (best >= 0) {
            void* p = blocks[best].p;
            *got = blocks[best].bytes;
            blocks.erase(blocks.begin() + best);
            return p;
        }
        GlobalPool& g = global();
        std::lock_guard<std::mutex> lk(g.mu);
        best = best_fit(g.blocks, bytes, dev);
        if (best < 0) return nullptr;
        void* p = g.blocks[best].p;
        *got = g.blocks[best].bytes;
        g.total -= g.blocks[best].bytes;
        g.blocks.erase(g.blocks.begin() + best);
        return p;
    }
    void give(void* p, size_t bytes) {
        if (leak) return;
        size_t total = bytes;
        for (const Block& x : blocks) total += x.bytes;
        if (blocks.size() >= 160 || total > ((size_t)24 << 30)) {
            (void)hipFree(p);
            return;
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        blocks.push_back({p, bytes, dev});
    }
    // an allocation failed: hand everything parked anywhere back to the driver before the retry
    static void release_global() {
        GlobalPool& g = global();
        std::lock_guard<std::mutex> lk(g.mu);
        for (const Block& b : g.blocks) (void)hipFree(b.p);
        g.blocks.clear();
        g.total = 0;
    }
};
struct CacheScope {
    DevBlockCache* prev;
    explicit CacheScope(DevBlockCache* c) : prev(DevBlockCache::current()) { DevBlockCache::current() = c; }
    ~CacheScope() { DevBlockCache::current() = prev; }
    CacheScope(const CacheScope&) = delete;
    CacheScope& operator=(const CacheScope&) = delete;
};

// Streams of engines that went away cleanly are parked (a few per device) for the next engine: creating and destroying the
// two streams of an engine cost ~1 ms of every one-shot call.  A parked stream is idle: its owner synchronised it.
struct StreamPool {
    static std::mutex& mu() {
        static std::mutex* m = new std::mutex();
        return *m;
    }
    static std::map<int, std::vector<hipStream_t>>& parked() {
        static auto* p = new std::map<int, std::vector<hipStream_t>>();  // never destroyed (no HIP calls at process exit)
        return *p;
    }
    static hipStream_t take() {
        int dev = 0;
        BMX_HIP(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(mu());
            auto& v = parked()[dev];
            if (!v.empty()) {
                hipStream_t s = v.back();
                v.pop_back();
                return s;
            }
        }
        hipStream_t s = nullptr;
        BMX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        return s;
    }
    // bmx_trim_caches: the parked streams of every device go back to the runtime (each destroyed on its own device)
    static void release_all() {
        std::map<int, std::vector<hipStream_t>> all;
        {
            std::lock_guard<std::mutex> lk(mu());
            all.swap(parked());
        }
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        for (auto& kv : all) {
            if (hipSetDevice(kv.first) != hipSuccess) continue;
            for (hipStream_t s : kv.second) (void)hipStreamDestroy(s);
        }
        if (have) (void)hipSetDevice(cur);
    }
    // `s` belongs to the current device and has been synchronised without an error
    static void give(hipStream_t s) {
        if (!s) return;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            std::lock_guard<std::mutex> lk(mu());
            auto& v = parked()[dev];
            if (v.size() < 8) {
                v.push_back(s);
                return;
            }
        }
        (void)hipStreamDestroy(s);
    }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies per device: remember (function, device) pairs already raised.
inline void ensure_dynamic_lds(const void* func, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    int dev = 0;
    BMX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = done[{func, dev}];
    if (bytes > have) {
        BMX_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
}

// (how many hipMallocs the process has made for DevBufs: printed with the one-shot call's timings under BMX_DEBUG=t)
inline std::atomic<long>& dev_malloc_calls() {
    static std::atomic<long> n{0};
    return n;
}

// Grow-only device buffer: the engine keeps these across calls so a steady-state run allocates nothing.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    bool view = false;  // p points into somebody else's block: nothing to release
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), view(o.view) { o.p = nullptr; o.cap = 0; o.view = false; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; cap = o.cap; view = o.view; o.p = nullptr; o.cap = 0; o.view = false; }
        return *this;
    }
    ~DevBuf() { release(); }
    // a window of n elements at q inside a block that outlives this buffer
    void alias(T* q, size_t n) {
        release();
        p = q;
        cap = n;
        view = true;
    }
    void release() {
        if (p && view) {
            p = nullptr;
            cap = 0;
            view = false;
            return;
        }
        if (p) {
            if (DevBlockCache* c = DevBlockCache::current())
                c->give(p, cap * sizeof(T));
            else
                (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
    }
    T* reserve(size_t n) {
        if (n > cap) {
            release();
            const size_t want = n + n / 8 + 64;
            size_t got = 0;
            DevBlockCache* c = DevBlockCache::current();
            if (void* q = c ? c->take(want * sizeof(T), &got) : nullptr) {
                p = static_cast<T*>(q);
                cap = got / sizeof(T);
            } else {
                ++dev_malloc_calls();
                hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
                if (e == hipErrorOutOfMemory) {  // parked blocks of earlier engines count as free memory
                    (void)hipGetLastError();
                    DevBlockCache::release_global();
                    e = hipMalloc((void**)&p, want * sizeof(T));
                }
                if (e != hipSuccess) {
                    p = nullptr;
                    throw bmx::Error(BMX_ERR_HIP, std::string("hipMalloc failed: ") + hipGetErrorString(e));
                }
                cap = want;
            }
        }
        return p;
    }
};

// ---------------------------------------------------------------------------------------------------
// Device-level operations (all on `stream`, all pointers device pointers, matrices row-major cells x dims)
// ---------------------------------------------------------------------------------------------------
struct KnnWorkspace {
    DevBuf<float> pq, pr;          // prepared (centred, fp16 / split-bf16, augmented) queries / references
    DevBuf<double> qn2, rn2, mean, red;
    DevBuf<int32_t> cand;          // [nq][C][KS]
    DevBuf<float> tau;             // [nq][C]
    DevBuf<float> cand_v;          // [nq][C][KS] approximate values of the candidates (refine pre-ranks by them)
    DevBuf<uint32_t> tau_g;        // [nq] per-query thresholds shared across reference ranges
    DevBuf<float> margin;          // [nq] twice the fp16 pass's error bound per query, in the pass's own units
    std::vector<DevBuf<uint32_t>> tau_rec;  // developer experiment "tau_replay": per search of a run, its queries' final thresholds
    size_t replay_idx = 0;
    DevBuf<float> samp_lists;      // [nq][ranges][KS] a split threshold sample's per-range lists (knn_f16.hip: sample_merge_kernel)
    DevBuf<uint32_t> tau_seed;     // [nq] seeded search: each query's seed threshold in the pass's units (orderable image)
    bool slots_clean = false;      // maxslots is zero (knn_refine leaves it so behind an fp16 search)
    DevBuf<unsigned long long> maxslots;  // 64 x 16 words: per-slot maxima of the reference norms (prep kernels)
    // per candidate tier: [count + 1] compact list of the queries it could not certify (+ counter in word 0), their
    // k-th candidate distances, and the scratch of the sub-search the next tier runs on them
    DevBuf<int32_t> flagged_t[2], sub_rows[2], sub_idx[2];
    DevBuf<int32_t> lk_rows, lk_idx;  // k > 36: the partitions' row lists, their neighbour lists [P][nq][36]
    DevBuf<double> lk_d2;             // ... their exact SQUARED distances [P][nq][36] (the merge then gathers no rows)
    bool dist_squared = false;        // set around the partitions' searches: every writer of a search's distances leaves them squared
    DevBuf<double> lk_kth;            // ... the first partition's kp-th distances
    DevBuf<float> lk_seed;            // ... and the seeds of the other partitions' searches made of them
    DevBuf<double> flag_bound_t[2], sub_dist[2];
    DevBuf<double> drow;           // exact-path distance rows
    DevBuf<double> xd;             // short exact lists
    DevBuf<int32_t> xcnt, xi, slow;
    // Optimistic searches (the engine's runs): no host read-back inside a search -- see search_tiers.  opt_state (device):
    // [0] raised when a search could not be completed that way, [1] queries that took the bounded exact sweep.
    static constexpr int OPT_CAP = 256;
    bool optimistic = false;
    bool xcnt_clear = false;
    DevBuf<int32_t> opt_state;
    int32_t* opt_state_ptr(hipStream_t s) {
        if (!opt_state.p) {
            opt_state.reserve(8);
            BMX_HIP(hipMemsetAsync(opt_state.p, 0, 8 * sizeof(int32_t), s));
        }
        return opt_state.p;
    }
    int force_exact = 0;           // testing hook: route every query through the exact path
    double wd_budget_s = 0.0;      // deadline of every host wait inside a search (0 = none); the engine scales it per search
    void sync(hipStream_t s) const { guarded_stream_sync(s, wd_budget_s); }
    // Small read-backs (counts that decide what is launched next) land in PINNED host memory owned by the workspace: the
    // copy is then really asynchronous -- a read-back into pageable memory may block inside hipMemcpyAsync, where no
    // deadline applies -- and its target outlives a wait that gives up.  64 words: [0] read_count, the rest the engine's.
    int64_t* pinned_words() {
        if (!pin_) pin_ = static_cast<int64_t*>(pinned_small_take());
        return pin_;
    }
    // 64 KiB pinned blocks are kept in a process-wide free list: hipHostMalloc / hipHostFree cost milliseconds, an engine
    // per call (the .Call boundary) would pay them every time
    static constexpr size_t kPinnedSmall = (size_t)64 << 10;
    static std::vector<void*>& pinned_small_pool() {
        static std::vector<void*>* v = new std::vector<void*>();
        return *v;
    }
    static std::mutex& pinned_small_mu() {
        static std::mutex* m = new std::mutex();
        return *m;
    }
    static void* pinned_small_take() {
        {
            std::lock_guard<std::mutex> lk(pinned_small_mu());
            auto& v = pinned_small_pool();
            if (!v.empty()) {
                void* p = v.back();
                v.pop_back();
                return p;
            }
        }
        void* p = nullptr;
        BMX_HIP(hipHostMalloc(&p, kPinnedSmall, hipHostMallocDefault));
        return p;
    }
    static void pinned_small_give(void* p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(pinned_small_mu());
        pinned_small_pool().push_back(p);
    }
    int64_t* pin_ = nullptr;
    int read_seq = 0;              // sequence number of the last word published to pin_ (knn.hip: read_count)
    bool abandon = false;          // the stream is stuck (watchdog): nothing that would wait for the device may be called
    // diagnostics of the last search / totals since the engine reset them
    int64_t last_exact = 0;             // queries that took the exact FP64 path
    int64_t last_flagged_tier[2] = {0, 0};  // queries each candidate tier could not certify
    int64_t exact_total = 0, tier2_total = 0;
    // profiling: when on, every launch of a candidate-pass kernel is bracketed by an event pair from this pool
    int last_variant = -1;  // candidate pass of the last search's first tier: 3 = fp16 ring, 2 = split-bf16 ring
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<int> event_tag;  // per used pair: 0 = sample pass, 1 = fp16 full pass, 2 = split-bf16 full pass, 3 = streaming section
    size_t events_used = 0;
    std::string last_kernel;     // name of the last full-pass candidate kernel, as rocprofv3 prints it
    std::pair<hipEvent_t, hipEvent_t> next_events(int tag = 1) {
        if (event_tag.size() <= events_used) event_tag.resize(events_used + 1);
        event_tag[events_used] = tag;
        if (events_used == events.size()) {
            hipEvent_t a, b;
            BMX_HIP(hipEventCreate(&a));
            BMX_HIP(hipEventCreate(&b));
            events.emplace_back(a, b);
        }
        return events[events_used++];
    }
    ~KnnWorkspace() {
        if (abandon) return;
        pinned_small_give(pin_);
        for (auto& e : events) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    }
};

// split-bf16 candidate pass (knn_bf16.hip)
struct Bf16Launch {
    const uint16_t* pq;
    const uint16_t* pr;
    int nqb, first_begin, range_len, nranges, r_limit, out_chunk0, out_nchunks;
    uint32_t* tau_g;        // per-query threshold shared by all ranges (orderable image); sample pass writes it
    int sample;             // non-zero: threshold-estimation pass, writes tau_g[q] only
    int32_t* cand;
    float* cand_v;          // approximate values of the candidates
    float* tau;
    int n_full = 0;         // the first n_full query blocks sweep [first_begin, r_limit) as ONE range; the others split it
    // fp16 tier: with margin[q] = twice the pass's error bound for query q (its own units) a list is cut at
    // (k-th best value + margin) instead of at its KS-th best -- what lies beyond cannot be among the k nearest
    const float* margin = nullptr;
    int k = 0;
    const uint32_t* tau_seed = nullptr;  // fp16 tier, sample pass of a seeded search: tau_g = min(sampled, tau_seed)
};
int bf16_pick_ns(int d);         // MFMA k-steps (16 bf16 each) for 3 d + 3 columns; 0 = unsupported
int bf16_ncons(int NS, int KS);  // consumer waves (32 queries each) per workgroup
void bf16_prep(hipStream_t stream, const double* X, const int32_t* rows, int n, int n_pad, int d, int NS,
               const double* mean, int is_query, uint16_t* P, double* n2, unsigned long long* maxbits,
               unsigned long long* slots);  // slots: 64 x 16 words of scratch for the reference-norm maximum
bool bf16_launch(hipStream_t stream, KnnWorkspace& ws, int NS, int KS, const Bf16Launch& L);
int f16_rows_per_slot(int NS, int KS);  // reference rows a ring slot of the fp16 tier holds (ranges are multiples of it)

// For rows q in [q_begin, q_end) of the query list: the k nearest rows of the reference list (exact, FP64 Euclidean,
// ties by lowest position).  X/Q are row-major [*, d]; ref_rows / q_rows (0-based, may be null = identity) select
// nr / nq rows.  idx_out [nq][k] receives 0-based POSITIONS in the reference list; dist_out [nq][k] Euclidean
// distances (may be null).  Only rows [q_begin, q_end) of the outputs are written.
// seed_d2 (nullable, [nq] f32): per query an upper bound of the squared distance beyond which the caller has no use
// for neighbours.  The search then starts from that threshold instead of a sampled one and a row may come back with
// fewer than k neighbours, padded with -1: exactly the references within the bound, or the k nearest if there are
// more than k of them.
// kth_out (nullable, [nq] device): the Euclidean distance of each row's k-th (last) neighbour where the row is full and
// certified by the first tier, +inf otherwise -- what lets the mutual-pair probe reject a candidate without reading the row.
// centre (nullable, [d] device): a point near the middle of the reference rows (the prepared images are taken relative to
// it: any vector is valid, a good one keeps the error bound tight); null: the mean of a strided sample is computed here.
void knn_device(hipStream_t stream, KnnWorkspace& ws, const double* X, const int32_t* ref_rows, int nr,
                const double* Q, const int32_t* q_rows, int nq, int d, int k, int32_t* idx_out, double* dist_out,
                int q_begin, int q_end, const float* seed_d2 = nullptr, const double* centre = nullptr,
                double* kth_out = nullptr);

}  // namespace bmx
