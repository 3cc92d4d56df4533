// Tier 1 of the exact kNN: a single-product fp16 candidate pass (MI355X / gfx950).
//
// The candidate pass only has to be accurate enough for the FP64 certificate of knn_refine to hold for almost every
// query; the few it cannot certify go on to the split-bf16 pass (knn_bf16.hip) and, failing that, to the exact
// FP64 scan.  So here every operand is ONE fp16 value: with the centred coordinates scaled by a power of two s so
// that the largest reference norm lands in (24, 48] (no overflow, no underflow that matters),
//     v' = |r'|^2 - 2 q'.r'   comes out of one chain of K/16 v_mfma_f32_32x32x16_f16 over
//     refs    [ r'_1 .. r'_d | n1 n2 n3 | 0 .. ],   n1 + n2 + n3 = f32(|r'|^2) exactly (three fp16 pieces)
//     queries [ -2 q'_1 .. -2 q'_d | 1 1 1 | 0 .. ]
// with |v' - exact| <= 2^-9 (1 + 2^-12) |q'||r'| + f32 accumulation terms: 4 MFMAs per 32 x 32 tile at 50 PCs
// (K = 64) against 10 for the split-bf16 pass (K = 160).  At 50 PCs on 100 000 reference cells the bound is about
// 0.05 against a gap of 0.35 between the 20th and the 32nd neighbour, so with KS = 32 kept candidates practically
// every query is certified.
//
// With the matrix work that cheap, what happens around it sets the pace, so a workgroup (one per CU) splits it over
// three kinds of waves:
//   * producers (2 waves) stream the prepared reference image into an LDS ring of slots of two tiles;
//   * consumers (8 waves x 32 queries) sweep the ring.  Per slot: the two tiles' MFMA chains interleaved on independent
//     accumulators with the fragment reads of the next slot in their gaps, then the filter -- the 32 x 32 accumulator
//     puts a query on each lane; min over its 16 values (v_min3 tree), one compare, one scalar branch: most tiles end
//     here -- and the spill: a lane whose group of 4 consecutive references holds a survivor dumps the group's four raw
//     values (one ds_write_b128) and a tag (tile, group, lane) into its wave's queue in the LDS.  Nothing else: no
//     list handling, no threshold bookkeeping in the sweep;
//   * service waves (4, lowest issue priority) work the queues off while the consumers sweep: 32 records = 128 values
//     per pass with all 64 lanes busy, each lane re-tests values against their query's current threshold and appends
//     survivors to the query's list with an LDS atomic; a list (KS kept + pending, <= 64 entries, unsorted) is cut back
//     to about its KS smallest by pivot trials / quickselect over the wave and the threshold, an LDS word the consumer
//     re-reads once per slot, drops to the cut.  They also write the final lists out (the consumers help).
// Hand-over everywhere through LDS words and counters, no barrier in the loop.  Shared thresholds across reference
// ranges and the register-only sample pass follow knn_bf16.hip.
#include "bmx_common.hpp"
#include "knn_select.hpp"

#include <cmath>
#include <string>

namespace bmx {
namespace {

using namespace sel;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int NCONS = 8;   // consumer waves (32 queries each): two per SIMD
constexpr int NPROD = 2;   // producer waves
#ifndef BMX_NSERV
#define BMX_NSERV 4
#endif
// service waves (each works off the spill queues of NCONS / n consumers).  Fourteen waves on a CU leave a wave 128
// registers, twelve 168: the long-K consumers (fragments of two tiles + query fragments + two accumulators) need the latter
__host__ __device__ constexpr int serv_waves(int NS) { return NS <= 5 ? BMX_NSERV : 2; }
constexpr int NQ = NCONS * 32;
constexpr int QCAP = 64;   // spill records per consumer wave (one group of one tile can fill all 64)
#ifndef BMX_DRAIN_MIN
#define BMX_DRAIN_MIN 16
#endif
#ifndef BMX_SPRIO
#define BMX_SPRIO 0
#endif
#ifndef BMX_CPRIO_LO
#define BMX_CPRIO_LO 0
#endif
#ifndef BMX_CPRIO_HI
#define BMX_CPRIO_HI 1
#endif
#ifndef BMX_HEAD
#define BMX_HEAD 4
#endif
#ifndef BMX_RING_MAX
#define BMX_RING_MAX 4
#endif
#ifndef BMX_NPIV
#define BMX_NPIV 4
#endif
#ifndef BMX_PSLEEP
#define BMX_PSLEEP 1
#endif
constexpr int EPI_CONS = 20;  // of a consumer's 32 final lists, those it writes out itself (its service wave: the rest)
constexpr int DRAIN_MIN = BMX_DRAIN_MIN;  // records in a queue before its service wave bothers (unless the consumer waits)
constexpr int HEAD = BMX_HEAD;    // a list is compacted ahead of time once fewer than HEAD slots are free

__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    const _Float16 h = (_Float16)f;  // v_cvt_f16_f32: round to nearest even, overflow to infinity
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }

// power-of-two scale that puts the largest reference norm into (24, 48]; 1 for degenerate input
__host__ __device__ __forceinline__ double f16_scale(double max_n2) {
    const double rm = sqrt(max_n2);
    if (!(rm > 1e-150) || !(rm < 1e150)) return 1.0;
    int e = 0;
    (void)frexp(48.0 / rm, &e);  // 48 / rm = f * 2^e, f in [0.5, 1)
    return ldexp(1.0, e - 1);
}

// ---------------------------------------------------------------------------------------------------
// prep.  PHASE 0 (references only): centred f32 rows -> exact norms n2[] and their maximum (the scale needs it
// before anything is written).  PHASE 1: the fp16 tile images (fragment-major for references, row-major for
// queries; same addressing as knn_prep_bf16), scaled.
// ---------------------------------------------------------------------------------------------------
// One tile of 32 rows (tile number `block`).  scale: PHASE 1 only.  Returns the row's exact squared norm (the eight threads
// of a row all hold it).
template <int PHASE>
__device__ __forceinline__ double prep_rows(char* smem_pp, int block, const double* __restrict__ X,
                                            const int32_t* __restrict__ rows, int n, int d, int NS,
                                            const double* __restrict__ mean, int is_query, uint16_t* __restrict__ P,
                                            double* __restrict__ n2, unsigned long long* __restrict__ max_slots,
                                            double scale) {
    const int K = 16 * NS;
    float* xs = reinterpret_cast<float*>(smem_pp);  // [32][d] centred, rounded to f32
    float* nrm = xs + 32 * d;                       // [32] f32(|x'|^2), scaled
    const int tid = threadIdx.x;
    const int r0 = block * 32;
    const float inv_d = 1.0f / (float)d;
    for (int e = tid; e < 32 * d; e += 256) {
        int rr = (int)(((float)e + 0.5f) * inv_d);  // e < 32 * 128: the float quotient is exact up to +-1
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
    const int rr = tid >> 3, sub = tid & 7;  // eight threads per row
    const bool live = r0 + rr < n;
    double s = 0.0;  // |x~|^2 of the f32-rounded centred row (exact products, FP64 sum)
    for (int c = sub; c < d; c += 8) {
        const double f = (double)xs[rr * d + c];
        s += f * f;
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if constexpr (PHASE == 0) {
        if (sub == 0 && live) n2[r0 + rr] = s;
        double m = live ? s : 0.0;
        for (int o = 8; o < 64; o <<= 1) m = fmax(m, __shfl_xor(m, o));
        if ((tid & 63) == 0)  // one atomic per wave, spread over 64 words a cache line apart
            atomicMax(max_slots + (size_t)(block & 63) * 16, (unsigned long long)__double_as_longlong(m));
        return s;
    }
    const float sf = (float)scale;  // a power of two: every product below is exact
    if (is_query) {
        // a query so far out that -2 q' leaves the fp16 range cannot be handled here: poison its norm, the
        // certificate of knn_refine then fails and the query goes to the next tier
        bool bad = false;
        for (int c = sub; c < d; c += 8) bad |= !(fabsf(-2.f * xs[rr * d + c] * sf) <= 65504.f);
        bad |= __shfl_xor((int)bad, 1) != 0;
        bad |= __shfl_xor((int)bad, 2) != 0;
        bad |= __shfl_xor((int)bad, 4) != 0;
        if (sub == 0 && live) n2[r0 + rr] = bad ? __builtin_nan("") : s;
    }
    if (sub == 0) nrm[rr] = (float)(s * scale * scale);
    __syncthreads();
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
        const float nf = nrm[prow];  // three fp16 pieces reproduce the f32 value exactly (33 bits)
        const uint16_t na = f32_to_f16_bits(nf);
        const float nr1 = nf - f16_bits_to_f32(na);
        const uint16_t nb = f32_to_f16_bits(nr1);
        const uint16_t nc = f32_to_f16_bits(nr1 - f16_bits_to_f32(nb));
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int L = 8 * i + e;
            uint16_t v = 0;
            if (L < d) {
                const float f = xs[prow * d + L] * sf;
                v = f32_to_f16_bits(is_query ? -2.f * f : f);
            } else if (L < d + 3) {
                const int piece = L - d;
                if (is_query)
                    v = plive ? 0x3C00 : 0;  // 1.0 (padded queries stay all-zero)
                else if (!plive)
                    v = piece == 0 ? 0x7C00 : 0;  // +inf: a padded reference never passes a threshold
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
    return s;
}

// PHASE 0 over the references: norms + the 64 slot maxima (the slots are zero when the search starts)
__global__ __launch_bounds__(256) void knn_prep_f16_norms(const double* __restrict__ X, const int32_t* __restrict__ rows, int n,
                                                          int d, int NS, const double* __restrict__ mean,
                                                          double* __restrict__ n2, unsigned long long* __restrict__ max_slots) {
    extern __shared__ __attribute__((aligned(16))) char smem_pp[];
    (void)prep_rows<0>(smem_pp, blockIdx.x, X, rows, n, d, NS, mean, 0, nullptr, n2, max_slots, 1.0);
}

// PHASE 1 of references AND queries in one launch (workgroups [0, nrb): reference tiles, the others: query tiles), with
// what used to be four more launches folded in: every workgroup folds the 64 slot maxima itself (workgroup 0 leaves the
// result in *max_n2_bits for the kernels behind and zeroes the search's flagged-query counter); the query workgroups write
// each query's margin (twice the pass's error bound) and, for a seeded search, its seed threshold -- tau_seed, which the
// sample pass folds into the thresholds it publishes, or straight into tau_g when there is no sample pass (tau_init).
__global__ __launch_bounds__(256) void knn_prep_f16_rq(const double* __restrict__ X, const int32_t* __restrict__ rrows, int nr,
                                                       int nrb, const double* __restrict__ Q,
                                                       const int32_t* __restrict__ qrows, int nq, int d, int NS,
                                                       const double* __restrict__ mean, uint16_t* __restrict__ Pr,
                                                       uint16_t* __restrict__ Pq, double* __restrict__ rn2,
                                                       double* __restrict__ qn2, const unsigned long long* __restrict__ slots,
                                                       unsigned long long* __restrict__ max_n2_bits,
                                                       int32_t* __restrict__ flagged0, float* __restrict__ margin, PassEps pe,
                                                       const float* __restrict__ seed_d2, uint32_t* __restrict__ tau_seed,
                                                       uint32_t* __restrict__ tau_init) {
    extern __shared__ __attribute__((aligned(16))) char smem_pp[];
    __shared__ double sh_max;
    if (threadIdx.x < 64) {
        double m = __longlong_as_double((long long)slots[(size_t)threadIdx.x * 16]);
        for (int o = 1; o < 64; o <<= 1) m = fmax(m, __shfl_xor(m, o));
        if (threadIdx.x == 0) {
            sh_max = m;
            if (blockIdx.x == 0) {
                *max_n2_bits = (unsigned long long)__double_as_longlong(m);
                if (flagged0) *flagged0 = 0;
            }
        }
    }
    __syncthreads();
    const double max_rn2 = sh_max;
    const double scale = f16_scale(max_rn2);
    if ((int)blockIdx.x < nrb) {
        (void)prep_rows<1>(smem_pp, blockIdx.x, X, rrows, nr, d, NS, mean, 0, Pr, rn2, nullptr, scale);
        return;
    }
    const int qb = blockIdx.x - nrb;
    const double s = prep_rows<1>(smem_pp, qb, Q, qrows, nq, d, NS, mean, 1, Pq, qn2, nullptr, scale);
    const int rr = threadIdx.x >> 3, sub = threadIdx.x & 7;
    if (sub == 0) {
        const int q = qb * 32 + rr;
        const bool live = q < nq;
        // (a query that left the fp16 range has a NaN norm in qn2: its margin and threshold are NaN-safe -- it fails the
        // certificate whatever the pass collects)
        const double nn = live ? qn2[q] : 0.0;
        (void)s;
        if (margin) margin[q] = live ? pass_margin(nn, max_rn2, pe) : 0.f;
        uint32_t ts = 0xFF800000u;  // the orderable image of +inf: no seed
        if (seed_d2) ts = live ? pass_seed_tau((double)seed_d2[q], nn, max_rn2, pe) : f32_orderable(-__builtin_inff());
        if (tau_seed) tau_seed[q] = ts;
        if (tau_init) tau_init[q] = ts;
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS budget: ring of reference tiles | per-query lists | per-wave spill queues | list counters | published
// thresholds | hand-over words
// ---------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int lds_fixed_bytes() { return NCONS * QCAP * 20 + NQ * 4 + NQ * 4 + NQ * 4 + 512; }
// A ring slot holds TP PAIRS of reference tiles (64 TP rows): the consumers wait, hand back and poll once per slot.
__host__ __device__ constexpr int ring_slots_for(int NS, int LCAP, int TP) {
    const int rest = 160 * 1024 - lds_fixed_bytes() - NQ * LCAP * 8;
    const int n = rest / (TP * 2 * NS * 1024);
    return n > BMX_RING_MAX ? BMX_RING_MAX : n;
}
#ifndef BMX_LCAP_MAX
#define BMX_LCAP_MAX 56
#endif
#ifndef BMX_TP_MAX
#define BMX_TP_MAX 2
#endif
// Slots of FOUR tiles (TP = 2) where a ring of three such slots fits beside lists of at least KS + 16 entries (up to 61
// columns at KS = 32): the hand-over -- ready check, polls, hand-back: a quarter of the base loop -- then comes once per
// four tiles (config 3: candidate passes -5 %; with two such slots: no gain, with four: as with three).  Else two tiles.
__host__ __device__ constexpr int tile_pairs_for(int NS, int KS) {
    return (BMX_TP_MAX >= 2 && KS + 16 <= BMX_LCAP_MAX && ring_slots_for(NS, KS + 16, 2) >= 3) ? 2 : 1;
}
// list capacity: the longest that leaves the ring three slots of four tiles, or at least two slots of two; else as short as
// a useful pending part allows
__host__ __device__ constexpr int list_cap(int NS, int KS) {
    const int TP = tile_pairs_for(NS, KS);
    for (int c = BMX_LCAP_MAX; c >= KS + 16; c -= 8)
        if (ring_slots_for(NS, c, TP) >= (TP == 2 ? 3 : 2)) return c;
    return KS + 16 <= 64 ? KS + 16 : 64;
}

__device__ __forceinline__ int lds_load_volatile(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int mbcnt64(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

#ifdef BMX_STAMPS
__device__ unsigned long long bmx_dbg16[48];
#define STAMP() __builtin_readcyclecounter()
#endif

template <int NS, int KS, int LCAP, bool SAMPLE>
__global__ __launch_bounds__((NCONS + NPROD + serv_waves(NS)) * 64) void knn_topk_f16(
    const uint16_t* __restrict__ Pq, const uint16_t* __restrict__ PrF, int first_begin, int range_len, int r_limit,
    int n_full, int nranges, int out_chunk0, int out_nchunks, uint32_t* __restrict__ tau_g, int32_t* __restrict__ cand,
    float* __restrict__ cand_v, float* __restrict__ tau_out, const float* __restrict__ margin_g, int kq,
    const uint32_t* __restrict__ tau_seed) {
    constexpr int TILE_BYTES = NS * 1024;
    constexpr int TP = tile_pairs_for(NS, KS);  // tile pairs per ring slot
    constexpr int SLOT_BYTES = TP * 2 * TILE_BYTES;
    constexpr int NSLOT = ring_slots_for(NS, LCAP, TP);
    static_assert(NSLOT >= 2, "LDS ring");
    static_assert(LCAP <= 64 && LCAP > KS, "one list entry per lane during compaction");
    // what a cut in mid-sweep may keep: a full list must come out with room again (the appends that found it full retry
    // until they fit), and a cut that frees next to nothing would be back at once
    constexpr int KEEP_MAX = KS + 11 < LCAP - 4 ? KS + 11 : LCAP - 4;
    static_assert(KEEP_MAX >= KS, "list capacity");
    constexpr int NSERV = serv_waves(NS);
    static_assert(NCONS % NSERV == 0 && (NCONS / NSERV) % 2 == 0, "a service wave's consumers: whole groups of 64 queries");
    static_assert((QCAP & (QCAP - 1)) == 0 && QCAP >= 64, "queue positions wrap by masking; one group can hold 64 records");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                                                           // [NSLOT][2][TILE_BYTES]
    unsigned long long* lists = reinterpret_cast<unsigned long long*>(smem + NSLOT * SLOT_BYTES);  // [NQ][LCAP]
    float* qvals = reinterpret_cast<float*>(lists + NQ * LCAP);                                   // [NCONS][QCAP][4]
    uint32_t* qtags = reinterpret_cast<uint32_t*>(qvals + NCONS * QCAP * 4);                      // [NCONS][QCAP]
    int* cnt = reinterpret_cast<int*>(qtags + NCONS * QCAP);                                      // [NQ]
    float* tauL = reinterpret_cast<float*>(cnt + NQ);  // [NQ] thresholds as the service waves last published them
    float* mgL = tauL + NQ;  // [NQ] twice the pass's error bound of each query (0: the KS-th-best cut only)
    int* ready = reinterpret_cast<int*>(mgL + NQ);     // [NSLOT]
    int* done = ready + NSLOT;  // [NSLOT] hand-backs of the position so far (every consumer adds one per slot read)
    int* wrL = done + NSLOT;    // [NCONS] records pushed so far (written by the consumer)
    int* rdL = wrL + NCONS;                            // [NCONS] records worked off so far (written by its service wave)
    int* stL = rdL + NCONS;  // [NCONS] consumer state: 0 sweeping, 1 waiting for room in its queue, 2 through
#ifdef BMX_EXP_ROWFILTER
    // timing experiment (EXPERIMENTS.md, round 5: what a SECOND, per-reference filter would cost the sweep -- the one sweep for
    // both findMutualNN directions of DESIGN section 7): 32 per-row thresholds a tile, read by every consumer for every tile
    // and held against all 16 accumulator registers; they stand at -inf, nothing ever passes
    float* rtau = reinterpret_cast<float*>(stL + NCONS);  // [2][16] in the fixed area's spare bytes
    if (threadIdx.x < 32) rtau[threadIdx.x] = -__builtin_inff();
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: branches on it are scalar
    // work items in launch order: first the query blocks that sweep the whole reference as one range, then the
    // remaining query blocks split into `nranges` ranges each, range-major (see knn_bf16.hip)
    int qblock = blockIdx.x, rng = 0, nrng = 1, r_begin = first_begin, r_end = r_limit;
    if ((int)blockIdx.x >= n_full) {
        const int nsplit = (gridDim.x - n_full) / nranges, jx = blockIdx.x - n_full;
        rng = jx / nsplit;
        qblock = n_full + (jx - rng * nsplit);
        nrng = nranges;
        r_begin = first_begin + rng * range_len;
        r_end = min(r_limit, r_begin + range_len);
    }
    const int ntiles = (r_end - r_begin) >> 5;  // even: ranges are multiples of 64 rows
    const int nslots = ntiles / (2 * TP);  // (ranges are multiples of 64 TP rows)
    const int out_chunk = out_chunk0 + rng;
    if (tid < 2 * NSLOT + 3 * NCONS) ready[tid] = 0;  // ready[], done[], wrL[], rdL[], stL[] are contiguous
    if (tid < NQ) {
        cnt[tid] = 0;
        tauL[tid] = (!SAMPLE && tau_g) ? orderable_f32(tau_g[qblock * NQ + tid]) : __builtin_inff();
        mgL[tid] = (!SAMPLE && margin_g) ? margin_g[qblock * NQ + tid] : 0.f;
    }
    __syncthreads();
    const bool shared_tau = !SAMPLE && tau_g != nullptr && nrng > 1;

#ifdef BMX_STAMPS
    unsigned long long dbg_ncomp = 0, dbg_comp = 0, dbg_rounds = 0;
#endif
    // (every lambda of this kernel is force-inlined: left to its own devices the compiler turned `compact` and
    // `drain` into real calls once they grew -- their captures then live in scratch and every LDS atomic becomes
    // a flat one.)
    // ---- compaction of query jj's list (jj wave-uniform) of consumer c: cut it back to about KS entries,
    // tighten the threshold.  The cut need not sit exactly at rank KS: any entry with at least KS - 1 smaller
    // ones is a valid new threshold (everything above it is dropped, at least KS stay).  BMX_NPIV entries from
    // fixed, evenly spread places of the (unordered) list are tried at once -- independent readlane / compare /
    // popcount chains instead of a dependent chain of quickselect rounds -- and the one that keeps the fewest,
    // at least KS, wins.  Only when none qualifies or the best would free too little does the exact quickselect
    // run.
    auto compact = [&](const int c, const int jj, const bool exact) __attribute__((always_inline)) {
        unsigned long long* mylists = lists + c * 32 * LCAP;
        int* mycnt = cnt + c * 32;
        const int n_raw = __builtin_amdgcn_readfirstlane(lds_load_volatile(&mycnt[jj]));
        const int n = n_raw < LCAP ? n_raw : LCAP;
        if (n <= KS && (n < kq || margin_g == nullptr)) return;
#ifdef BMX_STAMPS
        const unsigned long long c0 = STAMP();
        ++dbg_ncomp;
#endif
        const unsigned long long raw = lane < n ? mylists[jj * LCAP + lane] : 0ull;
        // 31-bit rank key: order-preserving image of the value with its low bits replaced by the lane number
        // (unique keys; values closer than 2^-17 relative may swap places at the cut, see below)
        const uint32_t key =
            lane < n ? (((f32_orderable(__uint_as_float((uint32_t)(raw >> 32))) >> 1) & ~63u) | (uint32_t)lane)
                     : 0x7FFFFFFFu;
        unsigned long long M = 0, keep = 0;
        int pl = 0, kept = 0x7FFFFFFF, nkeep = 0;
        uint32_t pk = 0;
        float newtau = 0.f;
        bool cut_done = false;
        // quickselect: the key with exactly `target` smaller keys among the lanes of B; leaves it in (pl, pk, M)
        auto select_rank = [&](unsigned long long B, const int target) __attribute__((always_inline)) {
            int it = 0;
            for (;;) {
                const int rot = (it * 29 + jj * 7) & 63;  // pivots from changing places
                ++it;
                const unsigned long long Br = B >> rot;
                pl = Br ? rot + __builtin_ctzll(Br) : __builtin_ctzll(B);
                pk = (uint32_t)__builtin_amdgcn_readlane((int)key, pl);
                M = __builtin_amdgcn_ballot_w64(key < pk);
                const int cc = __builtin_popcountll(M);
                if (cc == target) break;
                if (cc > target)
                    B &= M;
                else
                    B &= ~M & ~(1ull << pl);
            }
#ifdef BMX_STAMPS
            dbg_rounds += it;
#endif
        };
        const unsigned long long all_lanes = n == 64 ? ~0ull : ((1ull << n) - 1ull);
        // The margin cut: everything above (k-th best value + twice the error bound) is farther, exactly, than k others
        // of the list whatever the rounding did, so it can go -- typically a handful of entries beyond the k-th stay
        // instead of KS - k, and the threshold (what the consumers filter with) sits that much lower.  It applies when
        // the list has k entries and what it keeps fits; otherwise the rank cut below.
        const float mg = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(mgL[c * 32 + jj])));
        if (mg > 0.f && n >= kq) {
            select_rank(all_lanes, kq - 1);
            const float vk = orderable_f32((pk & ~63u) << 1);  // the k-th best value, low bits cleared (within 2^-17 of it)
            const float tm = vk + mg + fabsf(vk) * 6.103515625e-05f;  // (+ 2^-14: the cleared bits of vk and of the cut)
            const uint32_t ck = (f32_orderable(tm) >> 1) & ~63u;  // as a key with its lane bits cleared: <= tm
            const unsigned long long Mm = __builtin_amdgcn_ballot_w64(key < ck);
            const int cc = __builtin_popcountll(Mm);
            if (cc >= kq && cc <= (exact ? KS : KEEP_MAX)) {
                keep = Mm;
                nkeep = cc;
                newtau = orderable_f32(ck << 1);  // <= the value of everything dropped
                cut_done = true;
            }
        }
        if (!cut_done) {
            if (n <= KS) return;
            M = 0;
            pl = 0;
            pk = 0;
#pragma unroll
            for (int i = 0; i < BMX_NPIV; ++i) {
                const int pv = ((2 * i + 1) * n) / (2 * BMX_NPIV);  // < n
                const uint32_t k_i = (uint32_t)__builtin_amdgcn_readlane((int)key, pv);
                const unsigned long long m_i = __builtin_amdgcn_ballot_w64(key < k_i);
                const int c_i = __builtin_popcountll(m_i);
                if (c_i >= KS - 1 && c_i < kept) {
                    kept = c_i;
                    M = m_i;
                    pl = pv;
                    pk = k_i;
                }
            }
            if (exact ? kept != KS - 1 : kept + 1 > KEEP_MAX) {
                // quickselect for the key with exactly KS - 1 smaller keys; B = lanes that can still be it
                unsigned long long B = all_lanes;
                if (kept != 0x7FFFFFFF) B &= M;  // below the best pivot found
                select_rank(B, KS - 1);
                kept = KS - 1;
            }
            keep = M | (1ull << pl);  // kept + 1 lanes, at least KS
            nkeep = kept + 1;
            // the new threshold: the cut key with its lane bits cleared, which is <= the value of everything
            // dropped, so "rejected => value >= threshold" holds exactly
            newtau = orderable_f32((pk & ~63u) << 1);
        }
        const int pos = mbcnt64(keep);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every lane holds its entry before slots are rewritten
        if ((keep >> lane) & 1ull) mylists[jj * LCAP + pos] = raw;
        if (lane == 0) mycnt[jj] = nkeep;
        if (lane == jj) {
            const float told = tauL[c * 32 + jj];
            tauL[c * 32 + jj] = newtau < told ? newtau : told;  // the consumer picks it up at its next slot
            if (shared_tau) atomicMin(&tau_g[qblock * NQ + c * 32 + jj], f32_orderable(newtau));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef BMX_STAMPS
        dbg_comp += STAMP() - c0;
#endif
    };

    // ---- final lists [j0, j1) of consumer c out (its queue is empty, nothing appends any more)
    auto out_lists = [&](const int c, const int j0, const int j1) __attribute__((always_inline)) {
        unsigned long long* mylists = lists + c * 32 * LCAP;
        int* mycnt = cnt + c * 32;
        for (int jj = j0; jj < j1; ++jj) compact(c, jj, true);  // the output holds exactly KS entries per list
        for (int jj = j0; jj < j1; ++jj) {
            const int qq = qblock * NQ + c * 32 + jj;
            const int n = __builtin_amdgcn_readfirstlane(lds_load_volatile(&mycnt[jj]));  // <= KS now
            // what this range rejected was rejected against thresholds >= the final working threshold (the consumer
            // filters with a copy of its service wave's threshold, never a tighter one); kept entries at or above it
            // are as good as rejected (another range holds KS better ones): dropped here
            const float eff = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane(
                (int)__float_as_uint(__hip_atomic_load(&tauL[c * 32 + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))));
            if (lane < KS) {
                const unsigned long long e = lane < n ? mylists[jj * LCAP + lane] : 0ull;
                const float val = __uint_as_float((uint32_t)(e >> 32));
                const bool keep = lane < n && (val < eff || nrng == 1);
                const int64_t o = ((int64_t)qq * out_nchunks + out_chunk) * KS + lane;
                cand[o] = keep ? (int32_t)(uint32_t)e : -1;
                cand_v[o] = lane < n ? val : __builtin_inff();
            }
            if (lane == 0) tau_out[(int64_t)qq * out_nchunks + out_chunk] = eff;
            if (nrng == 1) {  // a whole-reference item owns every list column of its queries: the others stay empty
                for (int cc = 1; cc < out_nchunks; ++cc) {
                    if (lane < KS) cand[((int64_t)qq * out_nchunks + out_chunk + cc) * KS + lane] = -1;
                    if (lane == 0) tau_out[(int64_t)qq * out_nchunks + out_chunk + cc] = __builtin_inff();
                }
            }
        }
    };

    if (wave >= NCONS && wave < NCONS + NPROD) {
        // ------------------------------------------------------------------ producer (as in knn_bf16.hip)
#ifdef BMX_EXP_PPRIO
        __builtin_amdgcn_s_setprio(BMX_EXP_PPRIO);
#endif
        const int p = wave - NCONS;
        f32x4 ra[2 * NS], rb[2 * NS];
        const f32x4* src = reinterpret_cast<const f32x4*>(PrF) + ((int64_t)(r_begin >> 5) * NS) * 64 + lane;
        f32x4* ring_l = reinterpret_cast<f32x4*>(ring) + lane;
        auto load = [&](f32x4(&r)[2 * NS], int sl) {  // both tiles of slot sl: 2 NS contiguous 1 KiB pieces
#pragma unroll
            for (int s = 0; s < 2 * NS; ++s) r[s] = src[((int64_t)sl * 2 * NS + s) * 64];
        };
        // slot sl goes to ring position sl % NSLOT once every consumer has read the slots staged there before it:
        // done[pos] counts the hand-backs of the position, NCONS per slot
        auto wait_free = [&](int sl, int pos) __attribute__((always_inline)) {
            if (sl < NSLOT) return;
            const int need = (sl / NSLOT) * NCONS;
            while (__builtin_amdgcn_readfirstlane(lds_load_volatile(&done[pos])) < need)
                __builtin_amdgcn_s_sleep(BMX_PSLEEP);
        };
        auto publish = [&](const f32x4(&r)[2 * NS], int sl) {
            const int pos = sl % NSLOT;
            wait_free(sl, pos);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            f32x4* dst = ring_l + pos * (SLOT_BYTES / 16);
#pragma unroll
            for (int s = 0; s < 2 * NS; ++s) dst[s * 64] = r[s];
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            // same wave, in-order LDS queue: the flag lands after the tiles
            if (lane == 0) __hip_atomic_store(&ready[pos], sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // two register sets: the loads of slot sl + NPROD stay in flight while slot sl is handed over.  Every load below is
        // unconditional (past the end: the last slot again, never published): with conditional loads the compiler has to
        // wait for ALL outstanding loads before it touches a register set -- the set just asked for included -- because
        // it cannot tell how many are in flight
        if constexpr (TP == 1) {
            if (nslots > 0) {
                const int last = nslots - 1;
                int sl = p;
                load(ra, sl < last ? sl : last);
                load(rb, sl + NPROD < last ? sl + NPROD : last);
                for (; sl < nslots; sl += 2 * NPROD) {
                    publish(ra, sl);
                    load(ra, sl + 2 * NPROD < last ? sl + 2 * NPROD : last);
                    if (sl + NPROD < nslots) publish(rb, sl + NPROD);
                    load(rb, sl + 3 * NPROD < last ? sl + 3 * NPROD : last);
                }
            }
        } else {
            // slots of four tiles: a producer stages whole slots, its two register sets hold the two halves of one; the
            // ready word goes out after the second half
            auto load_half = [&](f32x4(&r)[2 * NS], int sl, int half) {
#pragma unroll
                for (int s = 0; s < 2 * NS; ++s) r[s] = src[(((int64_t)sl * TP + half) * 2 * NS + s) * 64];
            };
            auto store_half = [&](const f32x4(&r)[2 * NS], int pos, int half) {
                f32x4* dst = ring_l + pos * (SLOT_BYTES / 16) + half * (2 * TILE_BYTES / 16);
#pragma unroll
                for (int s = 0; s < 2 * NS; ++s) dst[s * 64] = r[s];
            };
            if (nslots > 0) {
                const int last = nslots - 1;
                int sl = p;
                load_half(ra, sl < last ? sl : last, 0);
                load_half(rb, sl < last ? sl : last, 1);
                for (; sl < nslots; sl += NPROD) {
                    const int pos = sl % NSLOT;
                    const int nx = sl + NPROD < last ? sl + NPROD : last;
                    wait_free(sl, pos);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    store_half(ra, pos, 0);
                    load_half(ra, nx, 0);
                    store_half(rb, pos, 1);
                    load_half(rb, nx, 1);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    if (lane == 0) __hip_atomic_store(&ready[pos], sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        // two empty slots past the end: the consumers' loop runs one step longer than the data and always refills from
        // "slot sl + 1"
        for (int e = nslots; e < nslots + 2; ++e) {
            if (nslots > 0 && e % NPROD == p) {
                const int pos = e % NSLOT;
                wait_free(e, pos);
                if (lane == 0) __hip_atomic_store(&ready[pos], e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        return;
    }

    if (wave >= NCONS + NPROD) {
        // ------------------------------------------------------------------ service wave: works off the spill queues
        // of CPS consumer waves while they sweep -- re-test, append to the lists, compact, publish the thresholds --
        // so that a consumer's own instruction stream holds the filter and the spill only
        if constexpr (SAMPLE) {
            return;
        } else {
            __builtin_amdgcn_s_setprio(BMX_SPRIO);
            constexpr int CPS = NCONS / NSERV;
            const int c0 = (wave - NCONS - NPROD) * CPS;
#ifdef BMX_STAMPS
            unsigned long long dbg_drain = 0, dbg_ndrain = 0, dbg_idle = 0;
            const unsigned long long dbg_t0 = STAMP();
#endif
            // ---- one drain pass: up to 32 records (128 values) of consumer c's queue, from position r0.  Every lane
            // takes one value of a record of the first half and one of the second (two independent chains of LDS read ->
            // the query's threshold -> LDS atomic slot -> store, issued together).
            auto drain = [&](const int c, const int r0, const int n) __attribute__((always_inline)) {
#ifdef BMX_STAMPS
                const unsigned long long d0 = STAMP();
                ++dbg_ndrain;
#endif
                unsigned long long* mylists = lists + c * 32 * LCAP;
                int* mycnt = cnt + c * 32;
                const float* qv = qvals + c * QCAP * 4;
                const uint32_t* qt = qtags + c * QCAP;
                const float* mytau = tauL + c * 32;
                bool pend[2];
                uint32_t tag[2];
                float v[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const int rec = 16 * x + (lane >> 2);
                    const int at = (r0 + rec) & (QCAP - 1);
                    pend[x] = rec < n;
                    tag[x] = qt[at];
                    v[x] = qv[(at << 2) + (lane & 3)];
                }
                // the records are on their way into registers: their queue positions are free again (the store queues
                // up behind the reads in this wave's in-order LDS queue)
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                if (lane == 0) __hip_atomic_store(&rdL[c], r0 + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                int jq[2], ref[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const int srcl = tag[x] & 63;  // the lane that spilled the record: query srcl & 31, row half srcl >> 5
                    jq[x] = srcl & 31;
                    ref[x] = r_begin + ((int)(tag[x] >> 8) << 5) + (int)((tag[x] >> 6) & 3u) * 8 + (srcl >> 5) * 4 +
                             (lane & 3);
                }
                for (;;) {
                    float tq[2];
                    int slot[2];
#pragma unroll
                    for (int x = 0; x < 2; ++x)
                        tq[x] = __hip_atomic_load(&mytau[jq[x]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        pend[x] = pend[x] && v[x] < tq[x];
                        slot[x] = LCAP;
                        if (pend[x]) slot[x] = atomicAdd(&mycnt[jq[x]], 1);  // ds_add_rtn_u32
                    }
#pragma unroll
                    for (int x = 0; x < 2; ++x)
                        if (slot[x] < LCAP) {
                            mylists[jq[x] * LCAP + slot[x]] =
                                ((unsigned long long)__float_as_uint(v[x]) << 32) | (uint32_t)ref[x];
                            pend[x] = false;
                        }
                    unsigned long long ov0 = __builtin_amdgcn_ballot_w64(pend[0]);
                    unsigned long long ov1 = __builtin_amdgcn_ballot_w64(pend[1]);
                    if ((ov0 | ov1) == 0) break;
                    // full lists: cut them back (the threshold drops), then the lanes left over try again
                    while (ov0 | ov1) {
                        const int qq = ov0 ? __builtin_amdgcn_readlane(jq[0], __builtin_ctzll(ov0))
                                           : __builtin_amdgcn_readlane(jq[1], __builtin_ctzll(ov1));
                        compact(c, qq, false);
                        ov0 &= ~__builtin_amdgcn_ballot_w64(jq[0] == qq);
                        ov1 &= ~__builtin_amdgcn_ballot_w64(jq[1] == qq);
                    }
                }
                // lists about to fill are cut back now, so that the appends of the next pass rarely find one full
                const int cn = lane < 32 ? lds_load_volatile(&mycnt[lane]) : 0;
                unsigned long long need = __builtin_amdgcn_ballot_w64(cn > LCAP - HEAD);
                while (need) {
                    const int jj = __builtin_ctzll(need);
                    need &= need - 1;
                    compact(c, jj, false);
                }
#ifdef BMX_STAMPS
                dbg_drain += STAMP() - d0;
#endif
            };

            uint32_t finmask = 0;  // bit s: consumer c0 + s is through and its final lists are out
            uint32_t tau_fetch[(CPS * 32 + 63) / 64];
#pragma unroll
            for (int x = 0; x < (CPS * 32 + 63) / 64; ++x) tau_fetch[x] = 0xFFFFFFFFu;
            int iter = 0;
            for (;;) {
                bool worked = false;
#pragma nounroll
                for (int s = 0; s < CPS; ++s) {
                    if ((finmask >> s) & 1u) continue;
                    const int c = c0 + s;
                    // the consumer writes its last record count before the "through" state: read in the opposite order
                    const int st = __builtin_amdgcn_readfirstlane(lds_load_volatile(&stL[c]));
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    const int w = __builtin_amdgcn_readfirstlane(lds_load_volatile(&wrL[c]));
                    const int r = __builtin_amdgcn_readfirstlane(lds_load_volatile(&rdL[c]));  // this wave's own word
                    const int avail = w - r;
                    // a short queue is left to grow (a pass costs the same for 1 record as for 32) unless its consumer
                    // waits for room or is through
                    if (avail >= DRAIN_MIN || (st != 0 && avail > 0)) {
                        drain(c, r, avail < 32 ? avail : 32);
                        worked = true;
                    } else if (st == 2) {
                        // nothing will be appended any more: the consumer wave writes the first EPI_CONS final lists out
                        // itself, this wave the rest
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        if (lane == 0) __hip_atomic_store(&stL[c], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        out_lists(c, EPI_CONS, 32);
                        finmask |= 1u << s;
                        worked = true;
                    }
                }
                if (finmask == (1u << CPS) - 1u) break;
                if (shared_tau) {
                    // thresholds other ranges have reached: the (L1-bypassing) loads are issued in one round of the loop
                    // and looked at in the next, so nobody waits for the round trip
                    ++iter;
#pragma unroll
                    for (int x = 0; x < (CPS * 32 + 63) / 64; ++x) {
                        const int qi = c0 * 32 + x * 64 + lane;  // query of the block (CPS * 32 is a multiple of 64)
                        if ((iter & 15) == 1) {
                            const float t = orderable_f32(tau_fetch[x]);
                            if (t < tauL[qi]) tauL[qi] = t;
                        }
                        if ((iter & 15) == 0)
                            tau_fetch[x] = __hip_atomic_load(&tau_g[qblock * NQ + qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (!worked) {
                    __builtin_amdgcn_s_sleep(BMX_PSLEEP);
#ifdef BMX_STAMPS
                    ++dbg_idle;
#endif
                }
            }
#ifdef BMX_STAMPS
            if (lane == 0) {
                atomicAdd(&bmx_dbg16[2], dbg_drain);
                atomicAdd(&bmx_dbg16[3], dbg_ndrain);
                atomicAdd(&bmx_dbg16[4], dbg_ncomp);
                atomicAdd(&bmx_dbg16[5], dbg_comp);
                atomicAdd(&bmx_dbg16[11], dbg_rounds);
                atomicAdd(&bmx_dbg16[13], STAMP() - dbg_t0);
                atomicAdd(&bmx_dbg16[14], 1ull);
                atomicAdd(&bmx_dbg16[15], dbg_idle);
            }
#endif
            return;
        }
    }

    // ---------------------------------------------------------------------- consumer
    const int j = lane & 31, h = lane >> 5;
    const int q = qblock * NQ + wave * 32 + j;

    float tau = tauL[wave * 32 + j];

    f16x8 bq[NS];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(Pq + (int64_t)q * (16 * NS) + h * (8 * NS));
#pragma unroll
        for (int s = 0; s < NS; ++s) bq[s] = __builtin_bit_cast(f16x8, src[s]);
    }

    float* qv = qvals + wave * QCAP * 4;
    uint32_t* qt = qtags + wave * QCAP;
    int wr = 0;       // records pushed so far (wave-uniform)
    int rd_seen = 0;  // what the service wave had worked off when last looked at
#ifdef BMX_STAMPS
    unsigned long long dbg_spin = 0, dbg_evt = 0, dbg_grp = 0, dbg_evc = 0, dbg_qwait = 0;
    const unsigned long long dbg_t0 = STAMP();
#endif

    // Hand-over, once per slot (two tiles).  While the tiles of slot sl compute, their fragment registers are refilled
    // from slot sl + 1 (each register right after the MFMA that consumed it); the ready word of slot sl + 1 is checked
    // before the first refill (it was polled a slot earlier: no LDS round trip in steady state), and after the last
    // refill the wave adds one to the position's hand-back counter -- queued behind the reads in the wave's in-order
    // LDS queue, so the producers cannot overwrite them early -- and polls the ready word of slot sl + 2 and its
    // queries' thresholds as the service wave has them now.  The slot loop is unrolled over the ring positions, so
    // every address below is a register set up once plus a constant.
    typedef __attribute__((address_space(3))) int* lds_iptr;
    lds_iptr done_p[NSLOT], ready_p[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        done_p[k] = (lds_iptr)&done[k];
        ready_p[k] = (lds_iptr)&ready[k];
        asm volatile("" : "+v"(done_p[k]), "+v"(ready_p[k]));
    }
    lds_iptr wr_p = (lds_iptr)&wrL[wave];
    int one = 1;
    asm volatile("" : "+v"(wr_p), "+v"(one));
    float tau_next = tau;
    int seen = 0;
    auto spin_until_staged = [&](const int slot, const int k) __attribute__((always_inline)) {  // slot, at position k
        if (__builtin_amdgcn_readfirstlane(seen) < slot + 1) {
#ifdef BMX_STAMPS
            const unsigned long long s0 = STAMP();
#endif
            do {
                __builtin_amdgcn_s_sleep(1);
                seen = *(volatile lds_iptr)ready_p[k];
            } while (seen < slot + 1);
#ifdef BMX_STAMPS
            dbg_spin += STAMP() - s0;
#endif
        }
    };
    auto lane0_store = [&](int* word, const int val) __attribute__((always_inline)) {
        // EXEC is all ones here (uniform control flow), so it is narrowed and restored with two scalar moves instead of
        // a compare / saveexec / branch sequence
        const lds_iptr dp = (lds_iptr)word;
        asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(dp), "v"(val) : "memory");
    };
    auto hand_back = [&](const int k) __attribute__((always_inline)) {  // the slot at position k has been read
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if constexpr (SAMPLE) {
            asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(done_p[k]), "v"(one) : "memory");
        } else {
            // ... and, in the same breath, the number of records pushed so far (they sit behind the records in the wave's
            // in-order LDS queue): one store per slot instead of one per spill
            asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, -1" ::"v"(done_p[k]),
                         "v"(one), "v"(wr_p), "v"(wr)
                         : "memory");
        }
    };
    // the ready word of the slot after next and the thresholds: polled ahead of the slot's fragment reads, looked at
    // one slot later (they are back long before the reads the wave has to wait for anyway)
    auto poll = [&](const int k) __attribute__((always_inline)) {
        seen = *(volatile lds_iptr)ready_p[k];
        if constexpr (!SAMPLE) tau_next = __hip_atomic_load(&tauL[wave * 32 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    f32x4 a[NS], b[NS];  // fragments of the even / odd tile of the current slot
    // SAMPLE: the lane's KS / 2 smallest candidates, ascending.  Insertion of x into a sorted list is one median per
    // entry, best'[i] = med3(best[i - 1], best[i], x), all of them independent when taken from the top down
    float best[SAMPLE ? KS / 2 : 1];
#pragma unroll
    for (int i = 0; i < (SAMPLE ? KS / 2 : 1); ++i) best[i] = __builtin_inff();
    auto sample_insert = [&](const float x) __attribute__((always_inline)) {
#pragma unroll
        for (int i = (SAMPLE ? KS / 2 : 1) - 1; i > 0; --i) best[i] = __builtin_amdgcn_fmed3f(best[i - 1], best[i], x);
        best[0] = fminf(fminf(best[0], x), x);
    };

    // room for `need` (<= QCAP) more records: the service wave's progress is looked at only when the last known state
    // does not leave enough
    auto wait_room = [&](const int need) __attribute__((always_inline)) {
#ifdef BMX_STAMPS
        const unsigned long long w0 = STAMP();
#endif
        rd_seen = __builtin_amdgcn_readfirstlane(lds_load_volatile(&rdL[wave]));
        if (wr + need - rd_seen > QCAP) {
            lane0_store(&wrL[wave], wr);
            lane0_store(&stL[wave], 1);  // "waiting": the service wave works the queue off whatever its length
            do {
                __builtin_amdgcn_s_sleep(1);
                rd_seen = __builtin_amdgcn_readfirstlane(lds_load_volatile(&rdL[wave]));
            } while (wr + need - rd_seen > QCAP);
            lane0_store(&stL[wave], 0);
        }
#ifdef BMX_STAMPS
        dbg_qwait += STAMP() - w0;
#endif
    };

    // ---- filter of one tile (products in `acc`, tile number t) and, where a value is below its query's threshold, the
    // spill.  Returns the lane minimum (SAMPLE uses nothing else).
#ifdef BMX_EXP_ROWFILTER
    f32x4 rt[4];
    float rt0 = -__builtin_inff();
    asm volatile("" : "+v"(rt0));
#endif
    auto sift = [&](const f32x16& acc, const int t) __attribute__((always_inline)) {
        // group minima (4 consecutive references each), then the lane minimum.  Every fminf below is half of a
        // three-way minimum: the compiler forms v_min3_f32 from such pairs and, unlike for a lone v_min_f32, does not put
        // a v_max x, x (signalling-NaN quieting) in front of its inputs.  The threshold rides along as the sixth operand
        // of a group: min(group, tau) < tau iff min(group) < tau (SAMPLE: tau = +inf, the plain minimum)
        float g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float m3 = fminf(fminf(acc[4 * u], acc[4 * u + 1]), acc[4 * u + 2]);
            g[u] = fminf(fminf(m3, acc[4 * u + 3]), tau);
        }
        const float mn = fminf(fminf(fminf(fminf(g[0], g[1]), g[2]), g[3]), tau);
#ifdef BMX_EXP_ROWFILTER
        if constexpr (!SAMPLE) {
            // register e of a lane holds row 8 (e / 4) + 4 (lane / 32) + (e % 4) of the tile: the lane half's 16 thresholds
            // (rt[]: asked for at the top of the tile pair, like the fragments)
            // (one subtraction per register, the minimum of the differences by v_min3, ONE compare: 16 v_cmp writing scalar
            // masks OR-ed in a chain measured 2.3x the whole sweep)
            float dm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a0 = acc[4 * u] - rt[u][0], a1 = acc[4 * u + 1] - rt[u][1], a2 = acc[4 * u + 2] - rt[u][2],
                            a3 = acc[4 * u + 3] - rt[u][3];
                dm[u] = fminf(fminf(fminf(a0, a1), a2), a3);
            }
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(fminf(fminf(fminf(dm[0], dm[1]), dm[2]), dm[3]) < 0.f);
            if (hm != 0) wr += 1;  // (never: the thresholds stand at -inf)
        }
#endif
        if constexpr (SAMPLE) {
            return mn;
        } else {
#if defined(BMX_EXP_NOFILTER)
            const unsigned long long any = 0;  // timing experiment: no filter, no events (results are garbage)
#elif defined(BMX_EXP_NOEVENT)
            const unsigned long long any = __builtin_amdgcn_ballot_w64(mn < -3.0e38f);  // timing experiment: nothing passes
#else
            const unsigned long long any = __builtin_amdgcn_ballot_w64(mn < tau);
#endif
            if (any == 0) return mn;
#ifdef BMX_STAMPS
            ++dbg_evt;
            const unsigned long long dbg_e0 = STAMP();
#endif
            // ---- spill the groups with survivors: four raw values and a tag per (lane, group) into the wave's queue;
            // the service wave takes it from there
            const uint32_t tagbase = ((uint32_t)t << 8) | (uint32_t)lane;
            auto spill = [&](const int u, const unsigned long long m) __attribute__((always_inline)) {
                if (g[u] < tau) {  // the lanes of m
                    const int slot = (wr + mbcnt64(m)) & (QCAP - 1);
                    f32x4 rec;
                    rec[0] = acc[4 * u];
                    rec[1] = acc[4 * u + 1];
                    rec[2] = acc[4 * u + 2];
                    rec[3] = acc[4 * u + 3];
                    *reinterpret_cast<f32x4*>(qv + slot * 4) = rec;
                    qt[slot] = tagbase | ((uint32_t)u << 6);
                }
                wr += __builtin_popcountll(m);
#ifdef BMX_STAMPS
                ++dbg_grp;
#endif
            };
            unsigned long long m[4];
            int tot = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                m[u] = __builtin_amdgcn_ballot_w64(g[u] < tau);
                tot += __builtin_popcountll(m[u]);
            }
            if (wr + tot - rd_seen > QCAP) {
                if (tot > QCAP) {
                    // the first tiles of a sweep that starts without a threshold: one group (<= 64 records) at a time,
                    // each against the threshold as it stands once there is room for it
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // (taken again: an earlier group of this tile may have waited and come back with a lower threshold)
                        unsigned long long mu = __builtin_amdgcn_ballot_w64(g[u] < tau);
                        if (mu == 0) continue;
                        const int c = __builtin_popcountll(mu);
                        if (wr + c - rd_seen > QCAP) {
                            wait_room(c);
                            const float tl =
                                __hip_atomic_load(&tauL[wave * 32 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            tau = tl < tau ? tl : tau;
                            mu = __builtin_amdgcn_ballot_w64(g[u] < tau);
                        }
                        if (mu) {
                            spill(u, mu);
                            __atomic_signal_fence(__ATOMIC_SEQ_CST);
                            lane0_store(&wrL[wave], wr);
                        }
                    }
#ifdef BMX_STAMPS
                    dbg_evc += STAMP() - dbg_e0;
#endif
                    return mn;
                }
                wait_room(tot);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) spill(u, m[u]);  // (the count goes out with the next done word)
#ifdef BMX_STAMPS
            dbg_evc += STAMP() - dbg_e0;
#endif
            return mn;
        }
    };

    // at equal priority the second-dispatched half of the consumers (waves 4..7, one per SIMD) loses every contested
    // issue slot to its older partner: it gets the higher static priority (MI355X_MICROARCH.md, two waves per SIMD)
    if (wave >= NCONS / 2)
        __builtin_amdgcn_s_setprio(BMX_CPRIO_HI);
    else
        __builtin_amdgcn_s_setprio(BMX_CPRIO_LO);
    // the query fragments have arrived: said here once, so that the compiler keeps no vmcnt wait for them in the loop
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (nslots > 0) {
        typedef __attribute__((address_space(3))) const char* lds_cptr;
        typedef __attribute__((address_space(3))) const f32x4* lds_f4ptr;
        lds_cptr ring_lane = (lds_cptr)ring + lane * 16;
        asm volatile("" : "+v"(ring_lane));
        spin_until_staged(0, 0);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        {
            const lds_f4ptr tp = (lds_f4ptr)ring_lane;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                a[s] = tp[s * 64];
                b[s] = tp[(NS + s) * 64];
            }
        }
        if constexpr (TP == 1) hand_back(0);  // (a slot of four tiles is handed back once its second half has been read)
        poll(1 % NSLOT);
        // One iteration = one pair of tiles.  Matrix phase: the two tiles' MFMA chains interleaved (independent
        // accumulators: the wave issues its 2 NS MFMAs back to back without waiting on its own results), each fragment
        // register refilled from the next pair right after the MFMA that consumed it.  Vector phase: filter and spill of
        // both tiles.  The two consumer waves of a SIMD fall into opposite phases: one's vector phase runs under the
        // other's MFMAs.  With slots of four tiles (TP = 2) only every other pair crosses a slot boundary: the ready
        // check, the polls and the hand-back happen once per four tiles.
        auto tile_pair = [&](const lds_f4ptr tp, const int t0, const int hand_back_pos) __attribute__((always_inline)) {
            f32x16 accA, accB;
#ifdef BMX_EXP_ROWFILTER
            if constexpr (!SAMPLE) {
#ifdef BMX_EXP_ROWFILTER_NOLDS
                // (the arithmetic alone: thresholds from registers, made opaque to the optimiser once per tile pair)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rt[u] = f32x4{rt0, rt0, rt0, rt0};
                    asm volatile("" : "+v"(rt[u]));
                }
#else
                typedef __attribute__((address_space(3))) volatile f32x4* lds_vf4;
                lds_vf4 rp = (lds_vf4)(rtau + (lane >> 5) * 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) rt[u] = rp[u];
#endif
            }
#endif
#pragma unroll
            for (int e = 0; e < 16; ++e) accA[e] = accB[e] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[s]), bq[s], accA, 0, 0, 0);
                a[s] = tp[s * 64];
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b[s]), bq[s], accB, 0, 0, 0);
                b[s] = tp[(NS + s) * 64];
            }
#pragma unroll
            for (int s = 0; s < 2 * NS; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
            }
            __builtin_amdgcn_sched_barrier(0);
            if (hand_back_pos >= 0) hand_back(hand_back_pos);  // that slot's tiles are all on their way into registers
            const float mnA = sift(accA, t0);
            const float mnB = sift(accB, t0 + 1);
            if constexpr (SAMPLE) {
                // sorted insertion of ONE candidate per pair: the lane's minimum over both tiles (32 references of
                // one query; still one distinct reference per candidate, half the insertions)
                sample_insert(fminf(fminf(mnA, mnB), mnB));
            }
        };
        for (int sl0 = 0; sl0 < nslots; sl0 += NSLOT) {
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) {
                const int sl = sl0 + k;  // its (first) fragments are in a[], b[]; slot sl + 1 sits at position kn
                if (sl >= nslots) break;
                const int kn = (k + 1) % NSLOT;
                if constexpr (TP == 2) {
                    // first pair of the slot: refill from the slot's own second half, then the slot is read
                    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this pair's fragments
                    tile_pair((lds_f4ptr)(ring_lane + k * SLOT_BYTES + 2 * TILE_BYTES), 4 * sl, k);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this pair's fragments, the next slot's ready word, tau
                if constexpr (!SAMPLE) asm("v_min_f32 %0, %0, %1" : "+v"(tau) : "v"(tau_next));
                spin_until_staged(sl + 1, kn);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                poll((kn + 1) % NSLOT);
                tile_pair((lds_f4ptr)(ring_lane + kn * SLOT_BYTES), 2 * TP * sl + 2 * (TP - 1), TP == 1 ? kn : -1);
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);

    if constexpr (SAMPLE) {
        const uint32_t mine = f32_orderable(best[KS / 2 - 1]), other = __shfl_xor(mine, 32);
        uint32_t start = max(mine, other);
        if constexpr (KS / 2 <= 24) {
            if (margin_g && kq >= 1 && kq <= KS) {
                // the margin form (see `compact`): the k-th smallest of the 2 x KS / 2 values the query's two lanes hold --
                // distinct references, so k references lie at or below it -- plus twice the error bound.  Each entry's
                // rank in the union = its place in its own (sorted) list + the partner's entries in front of it.
                float part[KS / 2];
#pragma unroll
                for (int e = 0; e < KS / 2; ++e) part[e] = __shfl_xor(best[e], 32);
                float uk = __builtin_inff();
#pragma unroll
                for (int i = 0; i < KS / 2; ++i) {
                    int r = i;
#pragma unroll
                    for (int e = 0; e < KS / 2; ++e) r += (part[e] < best[i] || (part[e] == best[i] && h == 1)) ? 1 : 0;
                    uk = r == kq - 1 ? best[i] : uk;
                }
                uk = fminf(uk, __shfl_xor(uk, 32));
                const float tm = uk + margin_g[q] + fabsf(uk) * 2.384185791015625e-07f;  // (the f32 sum rounded up)
                if (tm == tm) start = min(start, f32_orderable(tm));  // (no k-th value yet: inf, nothing changes)
            }
        }
        // (a seeded search: the tighter of the sampled threshold and the seed's, prep wrote the latter's image)
        // A sample split into ranges (few query blocks, see knn.hip): every range's threshold is valid by itself -- k of ITS
        // references lie at or below it -- so the tightest of them is; prep has written the seed (or +inf) there.
        // Round 6: the ranges also leave their lists (cand_v is free in a sample pass: [query][range][2 x KS / 2] values, all
        // of distinct references), and sample_merge_kernel takes the k-th smallest of their UNION -- what the unsplit sample
        // arrives at, so that splitting costs the full pass nothing (the tightest single range, round 5, left it 10 % looser).
        if (h == 0) {
            if (nrng > 1) atomicMin(&tau_g[q], start);
            else tau_g[q] = tau_seed ? min(start, tau_seed[q]) : start;
        }
        if (nrng > 1 && cand_v) {
            float* dst = cand_v + ((int64_t)q * nrng + rng) * KS + h * (KS / 2);
#pragma unroll
            for (int e = 0; e < KS / 2; e += 4)
                *reinterpret_cast<f32x4*>(dst + e) = f32x4{best[e], best[e + 1], best[e + 2], best[e + 3]};
        }
#ifdef BMX_STAMPS
        if (lane == 0) {
            atomicAdd(&bmx_dbg16[0], STAMP() - dbg_t0);
            atomicAdd(&bmx_dbg16[1], dbg_spin);
            atomicAdd(&bmx_dbg16[8], (unsigned long long)ntiles);
            atomicAdd(&bmx_dbg16[9], 1ull);
        }
#endif
    } else {
        // everything is in the queue: once the service wave has worked it off, both write final lists out
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        lane0_store(&wrL[wave], wr);
        lane0_store(&stL[wave], 2);
        while (__builtin_amdgcn_readfirstlane(lds_load_volatile(&stL[wave])) != 3) __builtin_amdgcn_s_sleep(2);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        out_lists(wave, 0, EPI_CONS);
#ifdef BMX_STAMPS
        if (lane == 0) {
            atomicAdd(&bmx_dbg16[0], STAMP() - dbg_t0);
            atomicAdd(&bmx_dbg16[1], dbg_spin);
            atomicAdd(&bmx_dbg16[6], dbg_evt);
            atomicAdd(&bmx_dbg16[7], dbg_grp);
            atomicAdd(&bmx_dbg16[8], (unsigned long long)ntiles);
            atomicAdd(&bmx_dbg16[9], 1ull);
            atomicAdd(&bmx_dbg16[10], dbg_evc);
            atomicAdd(&bmx_dbg16[12], dbg_qwait);
        }
#endif
    }
}

template <int NS, int KS>
void launch(hipStream_t stream, KnnWorkspace& ws, const Bf16Launch& L) {
    constexpr int LCAP = list_cap(NS, KS);
    constexpr int TP = tile_pairs_for(NS, KS);
    constexpr size_t lds =
        (size_t)ring_slots_for(NS, LCAP, TP) * TP * 2 * NS * 1024 + (size_t)NQ * LCAP * 8 + lds_fixed_bytes();
    static_assert(lds <= 160 * 1024, "LDS budget");
    ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_topk_f16<NS, KS, LCAP, false>), lds);
    ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_topk_f16<NS, KS, LCAP, true>), lds);
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (ws.profile) {
        ev = ws.next_events(L.sample ? 0 : 1);
        BMX_HIP(hipEventRecord(ev.first, stream));
    }
    if (!L.sample)
        ws.last_kernel = "knn_topk_f16<" + std::to_string(NS) + ", " + std::to_string(KS) + ", " + std::to_string(LCAP) +
                         ", false>";
    const int items = L.n_full + (L.nqb - L.n_full) * L.nranges;
    if (L.sample)
        hipLaunchKernelGGL((knn_topk_f16<NS, KS, LCAP, true>), dim3(items), dim3((NCONS + NPROD + serv_waves(NS)) * 64), lds, stream,
                           L.pq, L.pr, L.first_begin, L.range_len, L.r_limit, L.n_full, L.nranges, L.out_chunk0,
                           L.out_nchunks, L.tau_g, L.cand, L.cand_v, L.tau, L.margin, L.k, L.tau_seed);
    else
        hipLaunchKernelGGL((knn_topk_f16<NS, KS, LCAP, false>), dim3(items), dim3((NCONS + NPROD + serv_waves(NS)) * 64), lds, stream,
                           L.pq, L.pr, L.first_begin, L.range_len, L.r_limit, L.n_full, L.nranges, L.out_chunk0,
                           L.out_nchunks, L.tau_g, L.cand, L.cand_v, L.tau, L.margin, L.k, (const uint32_t*)nullptr);
    BMX_LAUNCH_CHECK();
    if (ws.profile) BMX_HIP(hipEventRecord(ev.second, stream));
#ifdef BMX_STAMPS
    {
        if (L.sample) fprintf(stderr, "(sample pass) ");
        unsigned long long hh[48];
        BMX_HIP(hipStreamSynchronize(stream));
        BMX_HIP(hipMemcpyFromSymbol(hh, HIP_SYMBOL(bmx_dbg16), sizeof(hh)));
        const double nt = (double)hh[8];
        const double sw = hh[14] ? (double)hh[14] : 1.0;
        fprintf(stderr,
                "[stamps f16] consumers=%.0f tiles/wave=%.0f cyc/tile=%.0f spin/tile=%.0f evpath/tile=%.0f qwait/tile=%.0f "
                "event tiles=%.3f groups/tile=%.3f | service waves=%.0f cyc=%.0f drain=%.0f (%.0f passes, %.0f cyc each) "
                "compact=%.0f (%.0f calls, %.1f rounds each) idle polls=%.0f\n",
                (double)hh[9], nt / hh[9], hh[0] / nt, hh[1] / nt, hh[10] / nt, hh[12] / nt, hh[6] / nt, hh[7] / nt,
                (double)hh[14], hh[13] / sw, hh[2] / sw, hh[3] / sw, hh[3] ? (double)hh[2] / hh[3] : 0.0, hh[5] / sw,
                hh[4] / sw, hh[4] ? (double)hh[11] / hh[4] : 0.0, hh[15] / sw);
        unsigned long long z[48] = {0};
        BMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(bmx_dbg16), z, sizeof(z)));
    }
#endif
}

}  // namespace

namespace {
// The split sample's thresholds (see the SAMPLE epilogue of knn_topk_f16): one wave per query, n = nranges x KS values of
// distinct references; the k-th smallest of them -- every value's rank by counting, ties by position -- plus twice the pass's
// error bound is a valid starting threshold (the margin form of the unsplit sample), taken if tighter than what stands there.
__global__ __launch_bounds__(256) void sample_merge_kernel(const float* __restrict__ lists, int n, int len, int nq, int k,
                                                           const float* __restrict__ margin_g, uint32_t* __restrict__ tau_g) {
    (void)len;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;  // (whole waves: nothing below synchronises the block)
    // the k-th smallest of the n <= 512 values (the ranges' lists, in any order): a lane holds up to eight of them as orderable
    // bit patterns, a bisection over the 32 bits counts with ballots -- no LDS, no dependent look-ups.  (Until late in round 6:
    // every entry's rank by a binary search in every other list, 36 dependent LDS reads an entry: 44 us a launch at 12 500 queries.)
    uint32_t val[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = lane + 64 * u;
        val[u] = i < n ? f32_orderable(lists[(int64_t)q * n + i]) : 0xFFFFFFFFu;
    }
    uint32_t t = 0;  // the smallest pattern with at least k values at or below it
    for (int b = 31; b >= 0; --b) {
        const uint32_t trial = t | ((1u << b) - 1u);  // the largest pattern with the bits decided so far and this bit clear
        int c = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) c += __popcll(__builtin_amdgcn_ballot_w64(val[u] <= trial));
        if (c < k) t |= 1u << b;
    }
    if (lane == 0) {
        const float uk = orderable_f32(t);
        const float tm = uk + margin_g[q] + fabsf(uk) * 2.384185791015625e-07f;  // (the f32 sum rounded up)
        if (tm == tm && tm < __builtin_inff()) atomicMin(&tau_g[q], f32_orderable(tm));
    }
}
}  // namespace

void f16_sample_merge(hipStream_t stream, const float* lists, int nranges, int KS, int nq, int k, const float* margin,
                      uint32_t* tau_g) {
    const int n = nranges * KS;
    if (n > 512 || nq <= 0 || k < 1 || k > n || !margin) return;  // (the ranges' own thresholds stand)
    hipLaunchKernelGGL(sample_merge_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, stream, lists, n, KS / 2, nq, k, margin, tau_g);
    BMX_LAUNCH_CHECK();
}

// MFMA k-steps (16 fp16 each) for d + 3 columns; 0 = this tier does not take the shape
int f16_pick_ns(int d, int KS) {
    const int ns = (d + 3 + 15) / 16;
    if (KS == BMX_KS1) return ns <= 8 ? ns : 0;
    if (KS == 48) return ns <= 4 ? ns : 0;
    return 0;
}

// References (norm pass, then the scaled image) and queries (image, norm, margin, seed threshold) of one search: two
// launches.  `slots` (64 x 16 words) must be zero on entry -- knn_refine leaves them so for the next search.
void f16_prep_all(hipStream_t stream, const double* X, const int32_t* rrows, int nr, int nr_pad, const double* Q,
                  const int32_t* qrows, int nq, int nq_pad, int d, int NS, const double* mean, uint16_t* Pr, uint16_t* Pq,
                  double* rn2, double* qn2, unsigned long long* maxbits, unsigned long long* slots, int32_t* flagged0,
                  float* margin, const PassEps& pe, const float* seed_d2, uint32_t* tau_seed, uint32_t* tau_init) {
    const size_t lds = (size_t)32 * d * 4 + 128 + 16;
    hipLaunchKernelGGL(knn_prep_f16_norms, dim3(nr_pad / 32), dim3(256), lds, stream, X, rrows, nr, d, NS, mean, rn2, slots);
    BMX_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_prep_f16_rq, dim3(nr_pad / 32 + nq_pad / 32), dim3(256), lds, stream, X, rrows, nr, nr_pad / 32, Q, qrows,
                       nq, d, NS, mean, Pr, Pq, rn2, qn2, (const unsigned long long*)slots, maxbits, flagged0, margin, pe, seed_d2,
                       tau_seed, tau_init);
    BMX_LAUNCH_CHECK();
}

bool f16_launch(hipStream_t stream, KnnWorkspace& ws, int NS, int KS, const Bf16Launch& L) {
#define BMX_CASE(N)                       \
    case N:                               \
        if (KS == BMX_KS1)                \
            launch<N, BMX_KS1>(stream, ws, L); \
        else if constexpr (N <= 4)        \
            launch<N, 48>(stream, ws, L); \
        else                              \
            return false;                 \
        return true;
    switch (NS) {
        BMX_CASE(1)
        BMX_CASE(2)
        BMX_CASE(3)
        BMX_CASE(4)
        BMX_CASE(5)
        BMX_CASE(6)
        BMX_CASE(7)
        BMX_CASE(8)
        default:
            return false;
    }
#undef BMX_CASE
}

double f16_scale_host(double max_n2) { return f16_scale(max_n2); }
int f16_rows_per_slot(int NS, int KS) { return 64 * tile_pairs_for(NS, KS); }

}  // namespace bmx
