// Split-bf16 candidate pass for the exact kNN (MI355X / gfx950).
//
// The f32-input MFMA runs at 1/16 of the bf16 rate, and the candidate pass only has to be accurate enough for the
// FP64 certificate of knn_refine to hold.  So the f32 operands are split into bf16 pieces,
//     r = rh + rl (+ 2^-16),   q' = -2q = qh + ql (+ 2^-16),
// and ONE v_mfma_f32_32x32x16_bf16 chain over the concatenated K = [qh|qh|ql|1 1 1] . [rh|rl|rh|n1 n2 n3] yields
//     v = |r|^2 + qh.rh + qh.rl + ql.rh        (|r|^2 as three bf16 pieces: exact to 2^-24)
// with |v - (|r|^2 - 2 q.r)| <= 3.03 * 2^-16 * 2 |q||r| + f32 accumulation error -- the bound knn_refine certifies
// against.  10 MFMAs of 32 cycles per 32x32 tile at 50 PCs instead of 28 of 64 cycles.
//
// At that rate the matrix pipe is no longer the limit; staging and selection are.  Structure of a workgroup:
//   * NCONS consumer waves, 32 queries each: query fragments resident in VGPRs; per query a sorted kept list and two
//     lane-private pending lists in the LDS, their lengths and the working threshold in registers (knn_select.hpp);
//   * NPROD producer waves stream the reference tiles (fragment-major, 1 KiB coalesced reads, two or three tiles in
//     flight per producer) into an LDS ring (as many slots as the LDS left by the lists holds) shared by all
//     consumers -- one read of the reference set per 32*NCONS queries;
//   * no s_barrier in the main loop: slots are handed over through LDS words (ready[slot] = tile number + 1, one
//     done word per slot and consumer), which the in-order LDS queue of each wave makes safe without extra waits.
//     A consumer that is busy merging a candidate list therefore delays nobody until the ring runs dry;
//   * the consumer loop is software-pipelined: under the dependent MFMA chain of tile t the wave refills its
//     fragment registers with tile t + 1 and filters tile t - 1; appends and list merges of tile t - 1 follow.
#include "bmx_common.hpp"
#include "knn_select.hpp"

#include <cmath>

namespace bmx {
namespace {

using namespace sel;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Workgroup shape per (fragment count, list size): NCONS consumer waves of 32 queries, NPROD producer waves (producer p
// stages tiles p, p + NPROD, ... into ring slot tile % NSLOT), PLN entries per lane-private pending list.
// 8 + 4 puts two consumers and one producer on every SIMD.  Measured alternatives at 100k x 100k, d = 50 (cycles per
// tile per consumer wave, 1480 for 8 + 4): 9 + 3 -> 1750, 10 + 2 -> 1940 -- the SIMDs that get a third consumer fall
// behind and the ring makes everybody else wait for them, so more queries per CU did not pay.
struct RingShape {
    int ncons, nprod, pln;
};
__host__ __device__ constexpr RingShape ring_shape(int NS, int KS) {
    return (NS <= 13 && KS == 24) ? RingShape{8, 4, 12} : RingShape{4, 4, 12};
}

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);  // inf / nan pass through
    u += 0x7FFFu + ((u >> 16) & 1u);                                   // round to nearest even
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

// logical element L of a prepared row -> position inside the fragment-major reference tile
__device__ __forceinline__ int64_t frag_pos(int r, int L, int NS) {
    const int hk = 8 * NS;  // elements per lane half
    const int h = L / hk, s = (L % hk) >> 3, j = L & 7;
    return ((((int64_t)(r >> 5) * NS + s) * 64 + h * 32 + (r & 31)) << 3) + j;
}

// One workgroup prepares one 32-row tile: rows are staged in LDS with coalesced reads, every output element is
// computed from there, and the tile image (fragment-major for references, row-major for queries -- both contiguous
// per tile) is written out in 16-byte pieces, consecutive threads writing consecutive pieces.
__global__ __launch_bounds__(256) void knn_prep_bf16(const double* __restrict__ X, const int32_t* __restrict__ rows,
                                                     int n, int n_pad, int d, int NS, const double* __restrict__ mean,
                                                     int is_query, uint16_t* __restrict__ P, double* __restrict__ n2,
                                                     unsigned long long* __restrict__ max_n2_bits) {
    extern __shared__ __attribute__((aligned(16))) char smem_pp[];
    const int K = 16 * NS;
    float* xs = reinterpret_cast<float*>(smem_pp);                       // [32][d] centred, rounded to f32
    float* nrm = xs + 32 * d;                                            // [32] f32(|x~|^2)
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 32;
    (void)n_pad;
    // (row, column) of a flat element index without an integer division: e < 32 * 128, so the float quotient is exact
    const float inv_d = 1.0f / (float)d;
    for (int e = tid; e < 32 * d; e += 256) {
        int rr = (int)(((float)e + 0.5f) * inv_d);
        int c = e - rr * d;
        if (c < 0) {
            --rr;
            c += d;
        } else if (c >= d) {
            ++rr;
            c -= d;
        }
        const int r = r0 + rr;
        float f = 0.f;
        if (r < n) {
            const int64_t row = rows ? rows[r] : r;
            f = (float)(X[row * d + c] - mean[c]);
        }
        xs[e] = f;
    }
    __syncthreads();
    // eight threads per row from here on: thread (rr, sub) owns the elements L = sub (mod 8) of row rr
    const int rr = tid >> 3, sub = tid & 7;
    const bool live = r0 + rr < n;
    {
        double s = 0.0;  // |x~|^2 of the f32-rounded centred row (exact products, FP64 sum; any order will do)
        for (int c = sub; c < d; c += 8) {
            const double f = (double)xs[rr * d + c];
            s += f * f;
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (sub == 0) {
            nrm[rr] = (float)s;
            if (live) n2[r0 + rr] = s;
        }
        if (!is_query) {
            // one atomic per wave, spread over 64 words a cache line apart (a single word serialises the whole
            // launch in the L2's atomic unit: it was the bulk of this kernel's time); fold_max_slots() finishes
            double m = live ? s : 0.0;
            for (int o = 8; o < 64; o <<= 1) m = fmax(m, __shfl_xor(m, o));
            if ((tid & 63) == 0)
                atomicMax(max_n2_bits + (size_t)(blockIdx.x & 63) * 16, (unsigned long long)__double_as_longlong(m));
        }
    }
    __syncthreads();
    // 16-byte pieces (8 consecutive elements of a row) straight from the staged rows to global memory: consecutive
    // threads write consecutive pieces of the tile image (fragment-major for references: k-step, lane half, row;
    // row-major for queries)
    uint4* out = reinterpret_cast<uint4*>(P + (int64_t)r0 * K);
    const int npieces = 32 * 2 * NS;
    const float inv_pc = 1.0f / (float)(2 * NS);
    for (int p = tid; p < npieces; p += 256) {
        int prow, i;  // row of the tile, 8-element chunk of the row
        if (is_query) {
            prow = (int)(((float)p + 0.5f) * inv_pc);
            i = p - prow * 2 * NS;
            if (i < 0) {
                --prow;
                i += 2 * NS;
            } else if (i >= 2 * NS) {
                ++prow;
                i -= 2 * NS;
            }
        } else {
            prow = p & 31;
            const int ih = p >> 5;  // = 2 * k-step + lane half
            i = (ih & 1) * NS + (ih >> 1);
        }
        const bool plive = r0 + prow < n;
        const float nf = nrm[prow];  // its three bf16 pieces reproduce the f32 value exactly
        const uint16_t na = f32_to_bf16(nf);
        const float nr1 = nf - bf16_to_f32(na);
        const uint16_t nb = f32_to_bf16(nr1);
        const uint16_t nc = f32_to_bf16(nr1 - bf16_to_f32(nb));
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int L = 8 * i + e;
            uint16_t v = 0;
            if (L < 3 * d) {
                const int blk = (L >= d) + (L >= 2 * d), c = L - blk * d;
                const float f = xs[prow * d + c];
                const float g = is_query ? -2.f * f : f;
                const uint16_t hi = f32_to_bf16(g);
                const bool want_lo = is_query ? blk == 2 : blk == 1;
                v = want_lo ? f32_to_bf16(g - bf16_to_f32(hi)) : hi;
            } else if (L < 3 * d + 3) {
                const int piece = L - 3 * d;
                if (is_query)
                    v = plive ? 0x3F80 : 0;  // 1.0 (padded queries stay all-zero)
                else if (!plive)
                    v = piece == 0 ? 0x7F80 : 0;  // +inf: a padded reference never passes a threshold
                else
                    v = piece == 0 ? na : (piece == 1 ? nb : nc);
            }
            if (e & 1)
                w[e >> 1] |= (uint32_t)v << 16;
            else
                w[e >> 1] = v;
        }
        out[p] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Ring depth: as many staged tiles as the LDS left over by the candidate lists holds (at most 8).  The consumers of a
// workgroup stall at different times (a compaction costs a couple of tile times); the deeper the ring, the less one
// consumer's stall holds up the others.
__host__ __device__ constexpr int ring_slots(int NS, int KS, int NCONS, int PLN) {
    const int rest = 160 * 1024 - NCONS * 32 * list_pitch(KS, PLN) * 8 - 512;
    const int n = rest / (NS * 1024);
    return n > 8 ? 8 : n;
}

#ifndef BMX_YIELD_TILES
#define BMX_YIELD_TILES 8
#endif
#ifdef BMX_STAMPS
__device__ unsigned long long bmx_dbg[48];
#define STAMP() __builtin_readcyclecounter()
#endif
__device__ __forceinline__ int lds_load_volatile(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// SAMPLE = true: threshold estimation only, entirely in registers and branch-free.  Each (tile, lane) contributes the
// minimum of its 16 values as ONE candidate; a lane keeps the KS/2 smallest of its candidates in a sorted register
// list (one min/max pair per entry and tile, issued in the gaps of the MFMA chain).  The two lanes of a query thus
// hold KS distinct references, so the larger of their two last entries bounds the KS-th nearest reference from
// above: a valid starting threshold for the full pass.  Nothing else is kept (the full pass rescans the sample
// rows), no list in the LDS is touched.  Output: tau_g[q] (order-preserving integer image of the threshold).
// SAMPLE = false: the full pass.  tau_g[q] is SHARED by all reference ranges of query q: every workgroup folds its
// own list's threshold into it (atomicMin) whenever it compacts, and refreshes its working threshold from it, so a
// range benefits from what the others have already found and extra ranges cost no extra selection work.  Each
// range's threshold is an upper bound of the KS-th nearest reference overall, hence so is their minimum; a stale
// read only leaves the filter looser, never wrong.
template <int NS, int KS, int NCONS, int NPROD, int PL, bool SAMPLE>
__global__ __launch_bounds__((NCONS + NPROD) * 64) void knn_topk_bf16(
    const uint16_t* __restrict__ Pq, const uint16_t* __restrict__ PrF, int first_begin, int range_len, int r_limit,
    int n_full, int nranges, int out_chunk0, int out_nchunks, uint32_t* __restrict__ tau_g, int32_t* __restrict__ cand,
    float* __restrict__ cand_v, float* __restrict__ tau_out) {
    constexpr int CAP = KS + 2 * PL;
    constexpr int PITCH = list_pitch(KS, PL);
    constexpr int NQ = NCONS * 32;
    constexpr int TILE_BYTES = NS * 1024;
    constexpr int NSLOT = ring_slots(NS, KS, NCONS, PL);
    static_assert(CAP <= 64, "one candidate per lane during compaction");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                                                        // [NSLOT][TILE_BYTES]
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem + NSLOT * TILE_BYTES);  // [NQ][CAP]
    int* ready = reinterpret_cast<int*>(buf + NQ * PITCH);                                     // [NSLOT]
    int* done = ready + NSLOT;                                                                // [NSLOT][NCONS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: branches on it are scalar
    // work items in launch order: first the query blocks that sweep the whole reference as one range (best selection
    // efficiency), then the remaining query blocks split into `nranges` ranges each (short items that fill the last
    // round of workgroups evenly)
    int qblock = blockIdx.x, rng = 0, nrng = 1, r_begin = first_begin, r_end = r_limit;
    if ((int)blockIdx.x >= n_full) {
        // range-major: the workgroups in flight at any time stream the same stretch of the reference (L2 reuse)
        const int nsplit = (gridDim.x - n_full) / nranges, j = blockIdx.x - n_full;
        rng = j / nsplit;
        qblock = n_full + (j - rng * nsplit);
        nrng = nranges;
        r_begin = first_begin + rng * range_len;
        r_end = min(r_limit, r_begin + range_len);
    }
    const int ntiles = (r_end - r_begin) >> 5;
    const int out_chunk = out_chunk0 + rng;
    if (tid < NSLOT * (1 + NCONS)) ready[tid] = 0;  // ready[] and done[] are contiguous
    __syncthreads();

    if (wave >= NCONS) {
        // ------------------------------------------------------------------ producer
        const int p = wave - NCONS;
        f32x4 ra[NS], rb[NS];
        const f32x4* src = reinterpret_cast<const f32x4*>(PrF) + ((int64_t)(r_begin >> 5) * NS) * 64 + lane;
        f32x4* ring_l = reinterpret_cast<f32x4*>(ring) + lane;
        auto load = [&](f32x4 (&r)[NS], int t) {
#pragma unroll
            for (int s = 0; s < NS; ++s) r[s] = src[((int64_t)t * NS + s) * 64];
        };
        // tile t goes to slot t % NSLOT once every consumer has read the tile staged there NSLOT tiles earlier (that
        // tile may have come from another producer: the consumers' done words order the two).  done[slot][c] is the
        // number of the last tile consumer c has read from the slot, plus one.
#ifdef BMX_STAMPS
        unsigned long long dbg_pw = 0, dbg_pt0 = STAMP();
#endif
        auto wait_free = [&](int t, int slot) {
            if (t < NSLOT) return;
            const int need = t - NSLOT + 1;
#ifdef BMX_STAMPS
            const unsigned long long s0 = STAMP();
            for (;;) {
                const int v = lane < NCONS ? lds_load_volatile(&done[slot * NCONS + lane]) : need;
                if (__builtin_amdgcn_ballot_w64(v < need) == 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
            dbg_pw += STAMP() - s0;
            return;
#endif
            for (;;) {
                const int v = lane < NCONS ? lds_load_volatile(&done[slot * NCONS + lane]) : need;
                if (__builtin_amdgcn_ballot_w64(v < need) == 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
        };
        auto publish = [&](const f32x4 (&r)[NS], int t) {
            const int slot = t % NSLOT;
            wait_free(t, slot);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            f32x4* dst = ring_l + slot * (TILE_BYTES / 16);
#pragma unroll
            for (int s = 0; s < NS; ++s) dst[s * 64] = r[s];
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            // same wave, in-order LDS queue: the flag lands after the tile
            if (lane == 0) __hip_atomic_store(&ready[slot], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // three register sets: the loads of tiles t + 4 and t + 8 are in flight while tile t is handed over
        f32x4 rc[NS];
        int t = p;
        if (t < ntiles) load(ra, t);
        if (t + NPROD < ntiles) load(rb, t + NPROD);
        for (; t < ntiles; t += 3 * NPROD) {
            if (t + 2 * NPROD < ntiles) load(rc, t + 2 * NPROD);
            publish(ra, t);
            if (t + NPROD < ntiles) {
                if (t + 3 * NPROD < ntiles) load(ra, t + 3 * NPROD);
                publish(rb, t + NPROD);
            }
            if (t + 2 * NPROD < ntiles) {
                if (t + 4 * NPROD < ntiles) load(rb, t + 4 * NPROD);
                publish(rc, t + 2 * NPROD);
            }
        }
        // two empty tiles past the end: the consumers' loop runs one tile longer than the data (the selection of a
        // tile happens under the MFMAs of the next one) and always prefetches "tile t + 1", with no last-tile case
        for (int e = ntiles; e < ntiles + 2; ++e) {
            if (ntiles > 0 && e % NPROD == p) {
                const int slot = e % NSLOT;
                wait_free(e, slot);
                if (lane == 0) __hip_atomic_store(&ready[slot], e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#ifdef BMX_STAMPS
        if (!SAMPLE && lane == 0) {
            atomicAdd(&bmx_dbg[32 + p], dbg_pw);
            atomicAdd(&bmx_dbg[36 + p], STAMP() - dbg_pt0);
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumer
    const int j = lane & 31, h = lane >> 5;
    const int qs = wave * 32 + j;
    const int q = qblock * NQ + qs;

    float tau = (!SAMPLE && tau_g) ? orderable_f32(tau_g[q]) : __builtin_inff();
    int nk = 0;  // entries in this query's kept list (the same in both of its lanes, like tau)

    bf16x8 bq[NS];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(Pq + (int64_t)q * (16 * NS) + h * (8 * NS));
#pragma unroll
        for (int s = 0; s < NS; ++s) bq[s] = __builtin_bit_cast(bf16x8, src[s]);
    }

    unsigned long long* pend = buf + qs * PITCH + KS + h * PL;
    int mycnt = 0;
#ifdef BMX_STAMPS
    unsigned long long dbg_spin = 0, dbg_flush = 0, dbg_nspin = 0, dbg_nflush = 0, dbg_ncomp = 0, dbg_evt = 0, dbg_grp = 0, dbg_evc = 0;
    const unsigned long long dbg_t0 = STAMP();
#endif

    // Tile hand-over.  Tile t + 1 (the producers stage two empty tiles past the end, so there always is one) is read
    // into the fragment registers while tile t computes, each register right after the MFMA that consumed it.  Its
    // ready word was polled one tile earlier (`seen`), so in steady state no LDS round trip is waited for.  After the
    // fragment reads the wave stores "read up to t + 1" in its done word of the slot -- queued behind the reads in
    // the wave's in-order LDS queue, so the producer cannot overwrite them early -- and polls the ready word of
    // tile t + 2.
    // The explicit lgkmcnt(0) at the top of every tile is free (everything queued is a tile old, the fragments of
    // tile t included) and leaves the compiler's wait-count pass with nothing pending: without it the pass makes the
    // MFMAs of tile t wait for the reads of tile t + 1 issued between them.
    const bool shared_tau = !SAMPLE && tau_g != nullptr && nrng > 1;
    uint32_t tau_fetch = 0xFFFFFFFFu;
    int seen = 0;
    int slot_n = 0;  // slot of the tile to read next
    int yield_tiles = 0;
    constexpr int YIELD = BMX_YIELD_TILES;
    auto spin_until_staged = [&](int tile) {
#ifdef BMX_STAMPS
        if (__builtin_amdgcn_readfirstlane(seen) < tile + 1) {
            yield_tiles = YIELD;
            const unsigned long long s0 = STAMP();
            ++dbg_nspin;
            while (seen < tile + 1) {
                __builtin_amdgcn_s_sleep(1);
                seen = lds_load_volatile(&ready[slot_n]);
            }
            dbg_spin += STAMP() - s0;
        }
#else
        if (__builtin_amdgcn_readfirstlane(seen) < tile + 1) {
            yield_tiles = YIELD;
            do {
                __builtin_amdgcn_s_sleep(1);
                seen = lds_load_volatile(&ready[slot_n]);
            } while (seen < tile + 1);
        }
#endif
    };
    auto read_tile = [&](f32x4 (&a)[NS]) {
        const f32x4* tp = reinterpret_cast<const f32x4*>(ring + slot_n * TILE_BYTES) + lane;
#pragma unroll
        for (int s = 0; s < NS; ++s) a[s] = tp[s * 64];
    };
    auto hand_back = [&](int tile) {
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (lane == 0)
            __hip_atomic_store(&done[slot_n * NCONS + wave], tile + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        slot_n = slot_n + 1 == NSLOT ? 0 : slot_n + 1;
        seen = lds_load_volatile(&ready[slot_n]);
    };

    // One step = the MFMA chain of tile t into `cur`, with three other things issued in the gaps of the chain (the
    // wave cannot issue the next, dependent MFMA before the previous one is through the pipe): the refill of each
    // fragment register with tile t + 1 right after the MFMA that read it (a single fragment set: a refill lands a
    // whole tile period before its use), and the clean-tile filter of tile t - 1, whose products sit in `prev`.
    f32x4 a[NS];
    uint32_t best[SAMPLE ? KS / 2 : 1];
#pragma unroll
    for (int i = 0; i < (SAMPLE ? KS / 2 : 1); ++i) best[i] = 0xFF800000u;  // image of +inf
    auto step = [&](f32x16& cur, const f32x16& prev, const int t, const int phase) {
        // Each SIMD hosts consumers w and w + 4; instruction arbitration goes by priority, then age, so at equal
        // priority the younger half (w >= 4) loses every contested slot, falls behind, and the older half waits for
        // it at the ring (measured: 290 vs 50 cycles of ring wait per tile).  A consumer that had to wait for the
        // ring is ahead of the others: it yields (priority 0) for the next few tiles, everybody else runs at 1.
        if (yield_tiles > 0) {
            __builtin_amdgcn_s_setprio(0);
            --yield_tiles;
        } else {
            __builtin_amdgcn_s_setprio(1);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        spin_until_staged(t + 1);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        typedef __attribute__((address_space(3))) const char* lds_cptr;
        typedef __attribute__((address_space(3))) const f32x4* lds_f4ptr;
        lds_cptr tbase = (lds_cptr)ring + (slot_n * TILE_BYTES + lane * 16);
        asm volatile("" : "+v"(tbase));     // the address is formed here, ...
        __builtin_amdgcn_sched_barrier(0);  // ... outside the interleaved block below
        const lds_f4ptr tp = (lds_f4ptr)tbase;
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s]), bq[s], cur, 0, 0, 0);
            a[s] = tp[s * 64];
        }
        // group minima (4 registers each), then the lane minimum, of the previous tile
        float g[4];
        unsigned long long gmask[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            g[u] = fminf(fminf(prev[4 * u], prev[4 * u + 1]), fminf(prev[4 * u + 2], prev[4 * u + 3]));
        const float mn = fminf(fminf(g[0], g[1]), fminf(g[2], g[3]));
        if constexpr (SAMPLE) {
            // sorted insertion on the order-preserving integer images (integer min/max need no NaN canonicalisation);
            // at step 0 `prev` is +inf and changes nothing
            uint32_t x = f32_orderable(mn);
#pragma unroll
            for (int i = 0; i < KS / 2; ++i) {
                const uint32_t lo = min(best[i], x);
                x = max(best[i], x);
                best[i] = lo;
            }
            asm volatile("" ::"v"(best[KS / 2 - 1]));  // keep it in this block
        } else {
            // the four group tests are evaluated here, under the MFMA chain, and leave scalar masks: the branches after
            // the chain then hang on scalar registers only (a mask that a later merge of this tile makes stale is
            // merely looser: every value is still compared with the current threshold)
#pragma unroll
            for (int u = 0; u < 4; ++u) gmask[u] = __builtin_amdgcn_ballot_w64(g[u] < tau);
            asm volatile("" ::"s"(gmask[0]), "s"(gmask[1]), "s"(gmask[2]), "s"(gmask[3]));  // keep the filter in this block
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
            __builtin_amdgcn_sched_group_barrier(0x002, SAMPLE ? 2 + (KS + NS - 1) / NS : 2, 0);  // a share of the VALU
        }
        hand_back(t + 1);
        if constexpr (SAMPLE) return;

        const int r0 = r_begin + ((t - 1) << 5);  // the tile whose products are in `prev`
        if ((gmask[0] | gmask[1] | gmask[2] | gmask[3]) == 0) return;
#ifdef BMX_STAMPS
        ++dbg_evt;
        const unsigned long long dbg_e0 = STAMP();
#endif
        auto flush_full = [&]() {
            unsigned long long fm = __builtin_amdgcn_ballot_w64(mycnt > PL - 4);  // keep 4 slots free
            if (fm) {
#ifdef BMX_STAMPS
                const unsigned long long s0 = STAMP();
                ++dbg_nflush;
                dbg_ncomp += __builtin_popcountll((fm | (fm >> 32)) & 0xFFFFFFFFull);
#endif
                fm = (fm | (fm >> 32)) & 0xFFFFFFFFull;
                while (fm) {
                    const int jj = __builtin_ctzll(fm);
                    fm &= fm - 1;
                    compact_regs<KS, PL>(buf, wave * 32 + jj, jj, lane, mycnt, nk, tau);
                }
                if constexpr (!SAMPLE) {
                    // publish this list's threshold (fire and forget; what the other ranges publish is picked up by
                    // the periodic refresh above)
                    if (shared_tau && h == 0) atomicMin(&tau_g[q], f32_orderable(tau));
                }
#ifdef BMX_STAMPS
                dbg_flush += STAMP() - s0;
#endif
            }
        };
        {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (gmask[u] == 0) continue;
#ifdef BMX_STAMPS
                ++dbg_grp;
#endif
                // a pending list keeps 4 free slots at this point, so the group's four values are appended in straight
                // line code: a lane whose value does not pass writes it to the list's last slot instead, which stays
                // unused as long as anything fails (at most 3 real entries land in the 4 free slots then)
                const int rbase = r0 + 8 * u + 4 * h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = prev[4 * u + e];
                    const bool pass = v < tau;
                    uint32_t* dst = reinterpret_cast<uint32_t*>(pend + (pass ? mycnt : PL - 1));
                    dst[0] = (uint32_t)(rbase + e);
                    dst[1] = __float_as_uint(v);
                    mycnt += pass ? 1 : 0;
                }
                flush_full();
            }
        }
#ifdef BMX_STAMPS
        dbg_evc += STAMP() - dbg_e0;
#endif
    };

    f32x16 accA, accB;
#pragma unroll
    for (int e = 0; e < 16; ++e) accA[e] = accB[e] = __builtin_inff();  // "previous tile" of step 0: nothing passes
    if (ntiles > 0) {
        spin_until_staged(0);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        read_tile(a);
        hand_back(0);
        // steps 0 .. ntiles: step t multiplies tile t (step ntiles: an empty tile, product unused) and filters tile
        // t - 1; two steps per iteration so that the two accumulators alternate statically
        for (int t2 = 0; t2 <= ntiles; t2 += 2) {
            if constexpr (!SAMPLE) {
                // periodic refresh of the shared threshold: the (L1-bypassing) load is issued here and looked at one
                // iteration later, so nobody waits for the round trip
                if (shared_tau) {
                    if ((t2 & 15) == 2) tau = fminf(tau, orderable_f32(tau_fetch));
                    if ((t2 & 15) == 0)
                        tau_fetch = __hip_atomic_load(&tau_g[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            step(accA, accB, t2, 0);
            if (t2 + 1 <= ntiles) step(accB, accA, t2 + 1, 1);
        }
    }

#ifdef BMX_STAMPS
    if (!SAMPLE && lane == 0) {
        atomicAdd(&bmx_dbg[0], STAMP() - dbg_t0);
        atomicAdd(&bmx_dbg[1], dbg_spin);
        atomicAdd(&bmx_dbg[2], dbg_flush);
        atomicAdd(&bmx_dbg[3], dbg_nspin);
        atomicAdd(&bmx_dbg[4], dbg_nflush);
        atomicAdd(&bmx_dbg[5], dbg_ncomp);
        atomicAdd(&bmx_dbg[6], dbg_evt);
        atomicAdd(&bmx_dbg[7], dbg_grp);
        atomicAdd(&bmx_dbg[8], (unsigned long long)ntiles);
        atomicAdd(&bmx_dbg[9], 1ull);
        atomicAdd(&bmx_dbg[10], dbg_evc);
        atomicAdd(&bmx_dbg[16 + wave], dbg_spin);
        atomicAdd(&bmx_dbg[24 + wave], dbg_flush);
    }
#endif
    if constexpr (SAMPLE) {
        const uint32_t mine = best[KS / 2 - 1], other = __shfl_xor(mine, 32);
        if (h == 0) tau_g[q] = max(mine, other);
        return;
    }
    for (int jj = 0; jj < 32; ++jj) compact_regs<KS, PL>(buf, wave * 32 + jj, jj, lane, mycnt, nk, tau);
    for (int jj = 0; jj < 32; ++jj) {
        const int s = wave * 32 + jj;
        const int qq = qblock * NQ + s;
        const int n = __builtin_amdgcn_readlane(nk, jj);
        // what this range rejected was rejected against thresholds >= the final working threshold of its lane pair;
        // kept entries at or above that threshold are as good as rejected (another range holds KS better ones), so
        // they are dropped here and the refine kernel only sees the few that matter
        const float w0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tau), jj));
        const float w1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tau), jj + 32));
        const float eff = w0 < w1 ? w0 : w1;
        if (lane < KS) {
            const unsigned long long key = buf[s * PITCH + lane];
            const float val = __uint_as_float((uint32_t)(key >> 32));
            const bool keep = lane < n && (val < eff || nrng == 1);
            cand[((int64_t)qq * out_nchunks + out_chunk) * KS + lane] = keep ? (int32_t)(uint32_t)key : -1;
            if (cand_v) cand_v[((int64_t)qq * out_nchunks + out_chunk) * KS + lane] = val;
        }
        if (lane == 0) tau_out[(int64_t)qq * out_nchunks + out_chunk] = eff;
        if (nrng == 1) {  // a whole-reference item owns every list column of its queries: the others stay empty
            for (int c = 1; c < out_nchunks; ++c) {
                if (lane < KS) cand[((int64_t)qq * out_nchunks + out_chunk + c) * KS + lane] = -1;
                if (lane == 0) tau_out[(int64_t)qq * out_nchunks + out_chunk + c] = __builtin_inff();
            }
        }
    }
}

template <int NS, int KS>
void launch(hipStream_t stream, KnnWorkspace& ws, const Bf16Launch& L) {
    constexpr RingShape SH = ring_shape(NS, KS);
    constexpr int NCONS = SH.ncons, NPROD = SH.nprod, PLN = SH.pln;
    constexpr size_t lds = (size_t)ring_slots(NS, KS, NCONS, PLN) * NS * 1024 +
                           (size_t)NCONS * 32 * list_pitch(KS, PLN) * 8 + 512;
    static_assert(lds <= 160 * 1024, "LDS budget");
    ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_topk_bf16<NS, KS, NCONS, NPROD, PLN, false>), lds);
    ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_topk_bf16<NS, KS, NCONS, NPROD, PLN, true>), lds);
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (ws.profile) {
        ev = ws.next_events(L.sample ? 0 : 2);
        BMX_HIP(hipEventRecord(ev.first, stream));
    }
    const int items = L.n_full + (L.nqb - L.n_full) * L.nranges;
    if (L.sample)
        hipLaunchKernelGGL((knn_topk_bf16<NS, KS, NCONS, NPROD, PLN, true>), dim3(items), dim3((NCONS + NPROD) * 64),
                           lds, stream, L.pq, L.pr, L.first_begin, L.range_len, L.r_limit, L.n_full, L.nranges,
                           L.out_chunk0, L.out_nchunks,
                           L.tau_g, L.cand, L.cand_v, L.tau);
    else
        hipLaunchKernelGGL((knn_topk_bf16<NS, KS, NCONS, NPROD, PLN, false>), dim3(items), dim3((NCONS + NPROD) * 64),
                           lds, stream, L.pq, L.pr, L.first_begin, L.range_len, L.r_limit, L.n_full, L.nranges,
                           L.out_chunk0, L.out_nchunks,
                           L.tau_g, L.cand, L.cand_v, L.tau);
    BMX_LAUNCH_CHECK();
    if (ws.profile) BMX_HIP(hipEventRecord(ev.second, stream));
#ifdef BMX_STAMPS
    if (!L.sample) {
        unsigned long long h[48];
        BMX_HIP(hipStreamSynchronize(stream));
        BMX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(bmx_dbg), sizeof(h)));
        const double w = (double)h[9], nt = (double)h[8];
        fprintf(stderr, "[stamps] waves=%.0f tiles/wave=%.0f cyc/tile=%.0f spin/tile=%.0f flush/tile=%.0f nspin/tile=%.3f nflush/tile=%.3f ncomp/tile=%.3f evt/tile=%.3f grp/tile=%.3f cyc/flush=%.0f evpath/tile=%.0f\n",
                w, nt / w, h[0] / nt, h[1] / nt, h[2] / nt, h[3] / nt, h[4] / nt, h[5] / nt, h[6] / nt, h[7] / nt, h[4] ? (double)h[2] / h[4] : 0.0, h[10] / nt);
        fprintf(stderr, "[stamps] spin/tile by consumer:");
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %.0f", h[16 + i] / (nt / 8));
        fprintf(stderr, "  flush/tile:");
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %.0f", h[24 + i] / (nt / 8));
        fprintf(stderr, "  producer waiting fraction:");
        for (int i = 0; i < 4; ++i) fprintf(stderr, " %.2f", h[36 + i] ? (double)h[32 + i] / h[36 + i] : 0.0);
        fprintf(stderr, "\n");
        unsigned long long z[48] = {0};
        BMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(bmx_dbg), z, sizeof(z)));
    }
#endif
}

}  // namespace

int bf16_pick_ns(int d) {
    static const int opts[] = {1, 2, 3, 4, 6, 8, 10, 13, 16, 19, 24};
    for (int o : opts)
        if (3 * d + 3 <= 16 * o) return o;
    return 0;
}

int bf16_ncons(int NS, int KS) { return ring_shape(NS, KS).ncons; }

__global__ void fold_max_slots(const unsigned long long* __restrict__ slots, unsigned long long* __restrict__ out) {
    double m = __longlong_as_double((long long)slots[(size_t)threadIdx.x * 16]);
    for (int o = 1; o < 64; o <<= 1) m = fmax(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

void bf16_prep(hipStream_t stream, const double* X, const int32_t* rows, int n, int n_pad, int d, int NS,
               const double* mean, int is_query, uint16_t* P, double* n2, unsigned long long* maxbits,
               unsigned long long* slots) {
    const size_t lds = (size_t)32 * d * 4 + 128 + 16;
    if (!is_query) BMX_HIP(hipMemsetAsync(slots, 0, 64 * 16 * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(knn_prep_bf16, dim3(n_pad / 32), dim3(256), lds, stream, X, rows, n, n_pad, d, NS, mean, is_query,
                       P, n2, slots);
    BMX_LAUNCH_CHECK();
    if (!is_query) {
        hipLaunchKernelGGL(fold_max_slots, dim3(1), dim3(64), 0, stream, slots, maxbits);
        BMX_LAUNCH_CHECK();
    }
}

bool bf16_launch(hipStream_t stream, KnnWorkspace& ws, int NS, int KS, const Bf16Launch& L) {
#define BMX_CASE(N)                        \
    case N:                                \
        if (KS == 24)                      \
            launch<N, 24>(stream, ws, L);  \
        else if constexpr (N <= 16)        \
            launch<N, 40>(stream, ws, L);  \
        else                               \
            return false;                  \
        return true;
    switch (NS) {
        BMX_CASE(1)
        BMX_CASE(2)
        BMX_CASE(3)
        BMX_CASE(4)
        BMX_CASE(6)
        BMX_CASE(8)
        BMX_CASE(10)
        BMX_CASE(13)
        BMX_CASE(16)
        BMX_CASE(19)
        BMX_CASE(24)
        default:
            return false;
    }
#undef BMX_CASE
}

}  // namespace bmx
