// Single-pass exclusive scan across the workgroups of ONE launch (decoupled look-back): the three launches of a classic
// scan (block sums, scan of the sums, add the offsets) and the flag / scatter kernels around them become one.  The merge
// loop compacts three lists per merge (listed left cells, pairs per left cell, MNN-involved right cells); at 100 000 cells
// each of those kernels ran 3-8 us, i.e. mostly launch latency.
//
// Logical block numbers are handed out by a ticket counter in arrival order, so every predecessor a block waits for has
// already started: no deadlock whatever order the hardware dispatches workgroups in.  A block publishes (epoch, flag,
// value) as ONE 64-bit word per block -- flag 1: the block's own sum, flag 2: the inclusive prefix up to and including it --
// and looks back a wavefront of predecessors at a time.  Words carry the launch's epoch, so the chain is never cleared
// between launches (a stale word simply reads as "not there yet"); the last logical block resets the ticket.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bmx {
namespace scan {

constexpr int ITEMS = 8;             // per thread
constexpr int BLOCK = 256 * ITEMS;   // per workgroup of 256 threads

struct Chain {
    unsigned long long* status;  // [>= number of blocks of the launch]
    unsigned int* ticket;        // zero before the first launch ever; every launch leaves it at zero
    unsigned int epoch;          // different from the epoch of the previous launch on this chain (and never 0)
};

__device__ __forceinline__ unsigned long long pack(unsigned int epoch, unsigned int flag, unsigned int value) {
    return ((unsigned long long)epoch << 34) | ((unsigned long long)flag << 32) | value;
}

// Called by ALL 256 threads of a workgroup, once, before anything else of the scan.  Returns the logical block number.
__device__ __forceinline__ int take_ticket(const Chain& ch, int* sh_word) {
    if (threadIdx.x == 0) *sh_word = (int)atomicAdd(ch.ticket, 1u);
    __syncthreads();
    const int b = *sh_word;
    __syncthreads();
    return b;
}

// `mine` = this thread's sum over its ITEMS consecutive items of logical block `bid`.  Returns the exclusive prefix of the
// thread's first item over the whole list; *total (if this is the last logical block, for every thread) the list's sum.
// nblocks = logical blocks of this launch.  All 256 threads call it.
__device__ __forceinline__ unsigned int exclusive_prefix(const Chain& ch, int bid, int nblocks, unsigned int mine,
                                                         unsigned int* sh4 /* [8] */, unsigned int* total, bool* is_last) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned int inc = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh4[w] = inc;
    __syncthreads();
    unsigned int woff = 0, bsum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned int t = sh4[i];
        bsum += t;
        woff += i < w ? t : 0u;
    }
    if (w == 0) {
        unsigned int excl = 0;
        if (bid == 0) {
            if (lane == 0) __hip_atomic_store(&ch.status[0], pack(ch.epoch, 2u, bsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0)
                __hip_atomic_store(&ch.status[bid], pack(ch.epoch, 1u, bsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int look = bid - 1;
            for (;;) {
                const int idx = look - lane;
                unsigned long long word = pack(ch.epoch, 2u, 0u);  // in front of block 0: an inclusive prefix of 0
                if (idx >= 0) word = __hip_atomic_load(&ch.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int flag = (unsigned int)(word >> 32) & 3u;
                const bool ok = (unsigned int)(word >> 34) == ch.epoch && flag != 0u;
                const unsigned long long m2 = __builtin_amdgcn_ballot_w64(ok && flag == 2u);
                const unsigned long long mbad = __builtin_amdgcn_ballot_w64(!ok);
                const int first2 = m2 ? __builtin_ctzll(m2) : 64;
                const int firstbad = mbad ? __builtin_ctzll(mbad) : 64;
                if (firstbad < (first2 < 63 ? first2 + 1 : 64)) {  // a predecessor this window needs is not there yet
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                unsigned int v = lane <= first2 ? (unsigned int)word : 0u;
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                excl += v;
                if (first2 < 64) break;
                look -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&ch.status[bid], pack(ch.epoch, 2u, excl + bsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            sh4[4] = excl;
            if (bid == nblocks - 1) atomicExch(ch.ticket, 0u);  // every ticket of this launch has been taken
        }
    }
    __syncthreads();
    const unsigned int base = sh4[4];
    *is_last = bid == nblocks - 1;
    *total = base + bsum;
    const unsigned int r = base + woff + (inc - mine);
    __syncthreads();  // (sh4 may be reused by a second scan of the same workgroup)
    return r;
}

}  // namespace scan
}  // namespace bmx
