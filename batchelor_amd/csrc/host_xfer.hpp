// Host <-> device transfers of the caller's PAGEABLE memory at link speed (the .Call boundary hands over R-owned
// matrices, src/RcppExports.cpp:10-47: nothing about them can be assumed pinned, and pinning 100s of MB per call with
// hipHostRegister costs more than the copy).
//
// A plain hipMemcpy of pageable memory is staged by the runtime through small pinned buffers by ONE host thread:
// measured 4-5 GB/s on the GPU box against > 50 GB/s that the PCIe link takes.  Here the staging is explicit: two pinned
// buffers per direction (allocated once per process and kept), a few host threads moving the bytes between the caller's
// memory and the pinned buffer (which also spreads the first-touch page faults of a freshly allocated result matrix),
// and the DMA of one buffer in flight while the other is being filled / emptied.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "bmx_common.hpp"

namespace bmx {

// A small persistent pool (the helpers only ever run memcpy: they never touch the caller's runtime, R or Python).
// A transfer is a burst of short jobs (one per 8 MB slot of the ring, ~150 us apart), so (a) the pieces of a job are claimed
// with one compare-and-swap on a (generation, next index) word instead of under the pool's mutex, and (b) a helper that has
// run out of work keeps watching the generation for a short while before it goes to sleep on the condition variable: with
// mutex + wake-up per job, 13 threads moved a result faster than 25 and 49 were slower still.
class HostPool {
  public:
    static HostPool& get() {
        static HostPool* pool = new HostPool();  // never destroyed: worker threads must not be joined at exit of a host
        return *pool;                            // process that may already have unloaded us
    }
    int workers() const { return (int)threads_.size() + 1; }
    // fn(i) for i in [0, n); the calling thread takes part; returns when all are done
    void parallel_for(size_t n, const std::function<void(size_t)>& fn) {
        if (n == 0) return;
        if (n == 1 || threads_.empty() || n >= 0xffffffffull) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        std::unique_lock<std::mutex> run_lock(run_mu_);  // one job at a time
        unsigned long long gen;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            pending_.store(n, std::memory_order_relaxed);
            gen = ++generation_;
            ticket_.store(gen << 32, std::memory_order_release);
            published_.store(gen, std::memory_order_release);
        }
        cv_.notify_all();
        work(gen, &fn, n);
        for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; ++spin) {
            if (spin < 4000) {
                relax();
            } else {
                std::unique_lock<std::mutex> lk(mu_);
                done_cv_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
                break;
            }
        }
    }

  private:
    static void relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        __asm__ __volatile__("isb" ::: "memory");
#else
        std::this_thread::yield();
#endif
    }
    HostPool() {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw >= 96 ? 24 : (hw >= 32 ? 12 : (hw >= 8 ? 6 : (hw >= 4 ? 3 : 0)));
        if (const char* v = std::getenv("BMX_HOST_THREADS")) n = std::max(0, std::atoi(v) - 1);
        for (int i = 0; i < n; ++i) threads_.emplace_back([this] { loop(); });
        for (auto& t : threads_) t.detach();
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            // stay awake for a moment after a job: the next one of a burst is picked up without a futex round trip
            const auto t0 = std::chrono::steady_clock::now();
            for (int spin = 0; published_.load(std::memory_order_acquire) == seen; ++spin) {
                relax();
                if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
            }
            const std::function<void(size_t)>* fn;
            size_t n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                fn = fn_;
                n = n_;
            }
            work(seen, fn, n);
        }
    }
    // Pieces of job `gen` (whose fn / n the caller read together with gen under mu_).  The ticket carries the job's
    // generation: a helper that arrives late fails the compare and never calls a function that has gone away -- a job's fn
    // is alive until its last claimed piece has reported back, and all n pieces are claimed before that.
    void work(unsigned long long gen, const std::function<void(size_t)>* fn, size_t n) {
        for (;;) {
            unsigned long long v = ticket_.load(std::memory_order_acquire);
            for (;;) {
                if ((v >> 32) != (gen & 0xffffffffull) || (v & 0xffffffffull) >= n) return;
                if (ticket_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) break;
            }
            (*fn)((size_t)(v & 0xffffffffull));
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0;
    unsigned long long generation_ = 0;  // (under mu_)
    alignas(64) std::atomic<unsigned long long> ticket_{0};     // (generation << 32) | next piece
    alignas(64) std::atomic<size_t> pending_{0};                // pieces not yet reported back
    alignas(64) std::atomic<unsigned long long> published_{0};  // the generation, for the helpers that are still awake
};

inline void host_parallel_memcpy(void* dst, const void* src, size_t bytes) {
    if (bytes <= ((size_t)128 << 10)) {
        std::memcpy(dst, src, bytes);
        return;
    }
    // (mid-size copies -- a merge's pair list into a freshly allocated R vector -- in finer pieces: what they cost is the
    // first touch of the destination's pages, which spreads over the threads like the bytes do)
    const size_t kPiece = bytes <= ((size_t)8 << 20) ? (size_t)64 << 10 : (size_t)256 << 10;
    const size_t n = (bytes + kPiece - 1) / kPiece;
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    HostPool::get().parallel_for(n, [&](size_t i) {
        const size_t o = i * kPiece;
        std::memcpy(d + o, s + o, std::min(kPiece, bytes - o));
    });
}

// Pinned staging buffers (two per direction and DEVICE: an event belongs to the device it was created on, and recording it
// on another device's stream fails), handed out under the ring's lock: transfers of different engines of one device take
// turns, engines on different devices do not wait for each other.
class PinnedRing {
  public:
    static constexpr size_t kChunk = (size_t)32 << 20;
    static PinnedRing& for_device(int direction) {
        static std::mutex mu;
        static std::map<std::pair<int, int>, PinnedRing*>* rings = new std::map<std::pair<int, int>, PinnedRing*>();
        int dev = 0;
        BMX_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        PinnedRing*& r = (*rings)[{dev, direction}];
        if (!r) r = new PinnedRing();  // never destroyed: no hipHostFree after the runtime is torn down
        return *r;
    }
    static PinnedRing& upload_ring() { return for_device(0); }
    std::mutex mu;
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int cur = 0;
    void ensure() {
        if (buf[0]) return;
        for (int i = 0; i < 2; ++i) {
            BMX_HIP(hipHostMalloc(&buf[i], kChunk, hipHostMallocDefault));
            BMX_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
    }
};

// Pinned blocks of a few MB (the pair lists of a run on their way to the host), kept in a small process-wide free list:
// hipHostMalloc / hipHostFree cost milliseconds, and the .Call boundary makes an engine per call.
struct PinnedBlocks {
    struct Blk {
        void* p;
        size_t bytes;
    };
    static std::mutex& mu() {
        static std::mutex* m = new std::mutex();
        return *m;
    }
    static std::vector<Blk>& pool() {
        static std::vector<Blk>* v = new std::vector<Blk>();
        return *v;
    }
    static Blk take(size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu());
            auto& v = pool();
            int best = -1;
            for (int i = 0; i < (int)v.size(); ++i)
                if (v[i].bytes >= bytes && (best < 0 || v[i].bytes < v[best].bytes)) best = i;
            if (best >= 0) {
                const Blk b = v[best];
                v.erase(v.begin() + best);
                return b;
            }
        }
        Blk b{nullptr, std::max(bytes + bytes / 4, (size_t)1 << 20)};
        BMX_HIP(hipHostMalloc(&b.p, b.bytes, hipHostMallocDefault));
        return b;
    }
    static void give(Blk b) {
        if (!b.p) return;
        std::lock_guard<std::mutex> lk(mu());
        auto& v = pool();
        if (v.size() >= 4) {  // keep the largest few
            int smallest = 0;
            for (int i = 1; i < (int)v.size(); ++i)
                if (v[i].bytes < v[smallest].bytes) smallest = i;
            if (v[smallest].bytes >= b.bytes) {
                (void)hipHostFree(b.p);
                return;
            }
            (void)hipHostFree(v[smallest].p);
            v.erase(v.begin() + smallest);
        }
        v.push_back(b);
    }
};

// A staging buffer's DMA is waited for with a deadline too (the watchdog contract of bmx_common.hpp: the host never waits
// without one): a copy queued behind a kernel that never ends gives up with an error instead of blocking the process.
// stale_ok: the event may have been recorded, last, on the stream of an engine that is gone (the UPLOAD ring's busy slots
// only: that ring outlives the engines that use it).  Such an event cannot be queried any more -- the runtime answers with an
// error of its stream-capture / invalid-handle family --, the engine drained its streams before it went, so there is
// nothing left to wait for and the slot is free.  Everywhere else (the download ring and the pair lists record their events
// in the call that waits for them) an error is a device fault and is reported: the bytes behind it must not be taken.
inline void guarded_event_sync(hipEvent_t ev, double budget_s = 120.0, bool stale_ok = false) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            const bool stale = e == hipErrorStreamCaptureUnsupported || e == hipErrorStreamCaptureInvalidated ||
                               e == hipErrorStreamCaptureIsolation || e == hipErrorStreamCaptureImplicit ||
                               e == hipErrorCapturedEvent || e == hipErrorInvalidHandle || e == hipErrorContextIsDestroyed ||
                               e == hipErrorInvalidResourceHandle;
            if (stale_ok && stale) return;
            BMX_HIP(e);
        }
        (void)hipGetLastError();
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > budget_s) throw WatchdogTimeout("watchdog: a staged host transfer did not finish in time");
        if (el > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// host (pageable) -> device on `stream`.  The caller's memory has been read completely when this returns; the last DMA
// may still be in flight on the stream (later work on the stream is ordered behind it).
inline void upload_pageable(void* dev, const void* host, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return;
    PinnedRing& r = PinnedRing::upload_ring();
    std::lock_guard<std::mutex> lk(r.mu);
    r.ensure();
    const char* src = static_cast<const char*>(host);
    char* dst = static_cast<char*>(dev);
    for (size_t o = 0; o < bytes; o += PinnedRing::kChunk) {
        const size_t m = std::min(PinnedRing::kChunk, bytes - o);
        const int c = r.cur;
        if (r.busy[c]) {
            try {
                guarded_event_sync(r.ev[c], 120.0, /* stale_ok */ true);
            } catch (...) {
                r.busy[0] = r.busy[1] = false;  // (the stream these were recorded on is stuck: the next user starts afresh)
                throw;
            }
        }
        host_parallel_memcpy(r.buf[c], src + o, m);
        BMX_HIP(hipMemcpyAsync(dst + o, r.buf[c], m, hipMemcpyHostToDevice, stream));
        BMX_HIP(hipEventRecord(r.ev[c], stream));
        r.busy[c] = true;
        r.cur ^= 1;
    }
}

// device -> host (pageable) on `stream`; returns when the caller's memory holds the bytes.  The device-side work the
// copies depend on must already be queued on `stream`.  The pieces (any number, any sizes) go through a ring of FOUR pinned
// slots of 8 MB: up to three DMAs are in flight while the host threads empty the fourth into the caller's memory (which also
// spreads the first-touch page faults of a freshly allocated result) -- with two slots of 32 MB the first DMA and the last
// host copy, 0.6 ms + 1.3 ms, were exposed at either end of every call.
struct XferPiece {
    void* host;
    const void* dev;
    size_t bytes;
};

class DownloadRing {
  public:
    static constexpr int kSlots = 4;
    static constexpr size_t kSlot = (size_t)8 << 20;
    static DownloadRing& for_device() {
        static std::mutex mu;
        static std::map<int, DownloadRing*>* rings = new std::map<int, DownloadRing*>();
        int dev = 0;
        BMX_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        DownloadRing*& r = (*rings)[dev];
        if (!r) r = new DownloadRing();  // never destroyed: no hipHostFree after the runtime is torn down
        return *r;
    }
    std::mutex mu;
    void* buf[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    void ensure() {
        if (buf[0]) return;
        for (int i = 0; i < kSlots; ++i) {
            BMX_HIP(hipHostMalloc(&buf[i], kSlot, hipHostMallocDefault));
            BMX_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
    }
};

inline void download_pieces(const XferPiece* pieces, size_t npieces, hipStream_t stream) {
    // cut into slot-sized jobs
    std::vector<XferPiece> jobs;
    for (size_t i = 0; i < npieces; ++i)
        for (size_t o = 0; o < pieces[i].bytes; o += DownloadRing::kSlot)
            jobs.push_back(XferPiece{static_cast<char*>(pieces[i].host) + o, static_cast<const char*>(pieces[i].dev) + o,
                                     std::min(DownloadRing::kSlot, pieces[i].bytes - o)});
    if (jobs.empty()) return;
    DownloadRing& r = DownloadRing::for_device();
    std::lock_guard<std::mutex> lk(r.mu);
    r.ensure();
    const size_t n = jobs.size();
    auto issue = [&](size_t i) {
        const int s = (int)(i % DownloadRing::kSlots);
        BMX_HIP(hipMemcpyAsync(r.buf[s], jobs[i].dev, jobs[i].bytes, hipMemcpyDeviceToHost, stream));
        BMX_HIP(hipEventRecord(r.ev[s], stream));
    };
    const size_t ahead = DownloadRing::kSlots - 1;
    for (size_t i = 0; i < std::min(ahead, n); ++i) issue(i);
    for (size_t i = 0; i < n; ++i) {
        const int s = (int)(i % DownloadRing::kSlots);
        guarded_event_sync(r.ev[s]);
        if (i + ahead < n) issue(i + ahead);  // (its slot was emptied in the previous round)
        host_parallel_memcpy(jobs[i].host, r.buf[s], jobs[i].bytes);
    }
}

inline void download_pageable(void* host, const void* dev, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return;
    const XferPiece p{host, dev, bytes};
    download_pieces(&p, 1, stream);
}

}  // namespace bmx
