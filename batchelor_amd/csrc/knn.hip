// Exact k-nearest-neighbour search on MI355X (gfx950) -- the >95 % hot spot of fastMNN / reducedMNN.
//
// Replaces the two BiocNeighbors::queryKNN calls inside findMutualNN (R/MNN_tree.R:129) and the one inside
// .tricube_weighted_correction (R/fastMNN.R:605).  Contract: exact Euclidean kNN, ascending distance; ties broken by
// lowest index (upstream leaves ties unpinned).
//
// Pipeline (all on one stream):
//   1. knn_prep        : centre on the reference mean (FP64), round to f32, append the augmented column so that one
//                        MFMA chain yields  v = |r|^2 - 2 q.r  (= squared distance minus the query's own norm).
//   2. knn_topk_mfma   : v_mfma_f32_32x32x2_f32 distance tiles; the 32x32 accumulator puts a QUERY on each lane and
//                        32 references in its registers, so the per-query threshold filter is lane-local; survivors
//                        go to a small per-query LDS buffer that one wave compacts (rank-by-counting) when it fills.
//                        Keeps KS = k + slack candidates per (query, reference chunk) and the chunk's final threshold.
//   3. knn_refine      : FP64 distances of the candidates in the reference's summation order (left-to-right over
//                        dimensions, no FMA contraction), exact (distance, index) ranking, and a rigorous check that
//                        no rejected reference can enter the top k given the f32 error bound; otherwise the query is
//                        flagged.
//   4. knn_exact       : flagged queries (ties, pathological data) and shapes outside the MFMA path are re-scanned
//                        entirely in FP64.
// Result: indices are exactly those of an FP64 brute-force search with (distance, index) ordering.
#include "bmx_common.hpp"
#include "knn_select.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace bmx {
namespace {

using namespace sel;

constexpr int QB = 128;          // queries per workgroup: 4 waves x one 32-query MFMA column tile
constexpr int RT = 64;           // references per staged LDS tile (two 32-row MFMA tiles)
constexpr int THREADS = 256;
constexpr int MAX_CHUNKS = 8;

// ---------------------------------------------------------------------------------------------------
// column sums over a row list (two deterministic stages)
// ---------------------------------------------------------------------------------------------------
// (row i of the sum is row i * stride of the list: a strided sample when stride > 1)
__global__ void colsum_partial(const double* __restrict__ X, const int32_t* __restrict__ rows, int n, int d,
                               int rows_per_block, int stride, double* __restrict__ partial) {
    __shared__ double sm[4][64];
    const int c0 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    for (int cb = 0; cb < d; cb += 64) {
        const int c = cb + c0;
        double s = 0.0;
        if (c < d)
            for (int r = r0 + rl; r < r1; r += 4) {
                const int64_t rs = (int64_t)r * stride;
                const int64_t row = rows ? rows[rs] : rs;
                s += X[row * d + c];
            }
        sm[rl][c0] = s;
        __syncthreads();
        if (rl == 0 && c < d) partial[(int64_t)blockIdx.x * d + c] = (sm[0][c0] + sm[1][c0]) + (sm[2][c0] + sm[3][c0]);
        __syncthreads();
    }
}

__global__ void colsum_final(const double* __restrict__ partial, int nblocks, int d, double scale,
                             double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * d + c];
    out[c] = s * scale;
}

// ---------------------------------------------------------------------------------------------------
// 1. prep: FP64 rows -> centred f32 rows [n_pad][KP] with the augmented column; exact norm^2 of the rounded row
// ---------------------------------------------------------------------------------------------------
// frag != 0 writes the MFMA-fragment-major layout of the wave-per-workgroup kernel: for every 32-row tile and every
// group m of four K elements, the 64 lanes' 16-byte pieces are contiguous (lane = 32 * K-half + row), so each
// global_load_dwordx4 of a wave is one fully coalesced 1 KiB read:  P[((tile * KP/8 + m) * 64 + lane) * 4 + x].
__global__ void knn_prep(const double* __restrict__ X, const int32_t* __restrict__ rows, int n, int n_pad, int d,
                         int KP, const double* __restrict__ mean, int is_query, int frag, float* __restrict__ P,
                         double* __restrict__ n2, unsigned long long* __restrict__ max_n2_bits) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    const int HK = KP / 2;
    auto at = [&](int c) -> float& {
        if (!frag) return P[(int64_t)r * KP + c];
        const int hh = c / HK, m = (c % HK) >> 2, x = c & 3;
        return P[(((int64_t)(r >> 5) * (HK >> 2) + m) * 64 + hh * 32 + (r & 31)) * 4 + x];
    };
    if (r >= n) {
        for (int c = 0; c < KP; ++c) at(c) = 0.f;
        if (!is_query) at(d) = __builtin_inff();  // padded references can never pass a threshold
        return;
    }
    const int64_t row = rows ? rows[r] : r;
    const double* x = X + row * d;
    double s = 0.0;
    for (int c = 0; c < d; ++c) {
        const float f = (float)(x[c] - mean[c]);
        s += (double)f * (double)f;
        at(c) = is_query ? -2.f * f : f;
    }
    at(d) = is_query ? 1.f : (float)s;
    for (int c = d + 1; c < KP; ++c) at(c) = 0.f;
    n2[r] = s;
    if (!is_query) atomicMax(max_n2_bits, (unsigned long long)__double_as_longlong(s));
}

// ---------------------------------------------------------------------------------------------------
// 2. MFMA distance tiles + per-query threshold / buffer selection
// ---------------------------------------------------------------------------------------------------
// Staging registers as a recursive struct (an array here ends up in scratch memory).
template <int N>
struct StageRegs {
    f32x4 v;
    StageRegs<N - 1> rest;
};
template <>
struct StageRegs<0> {};

template <int N>
__device__ __forceinline__ void stage_load(StageRegs<N>& s, const f32x4* __restrict__ src, int e) {
    s.v = src[e];  // may over-read into the next tile / the tail padding of the prepared references
    if constexpr (N > 1) stage_load(s.rest, src, e + THREADS);
}
template <int N, int TOTAL>
__device__ __forceinline__ void stage_store(const StageRegs<N>& s, f32x4* dst, int e) {
    if (e < TOTAL) dst[e] = s.v;
    if constexpr (N > 1) stage_store<N - 1, TOTAL>(s.rest, dst, e + THREADS);
}

// grid = (query blocks, reference ranges).  Range c covers prepared reference rows [r_begin[c], r_end[c]) (tile
// multiples); tau_init (nullable) holds a valid starting threshold per query (from the sample pre-pass).
template <int KP, int KS>
__global__ __launch_bounds__(THREADS, 2) void knn_topk_mfma(const float* __restrict__ Pq, const float* __restrict__ Pr,
                                                            int first_begin, int range_len, int r_limit,
                                                            int out_chunk0, int out_nchunks,
                                                            const float* __restrict__ tau_init,
                                                            int32_t* __restrict__ cand, float* __restrict__ tau_out) {
    constexpr int CAP = KS + 2 * PL;
    constexpr int HK = KP / 2;  // K elements per lane half
    constexpr int TILE_F4 = RT * KP / 4;
    constexpr int NST = (TILE_F4 + THREADS - 1) / THREADS;
    static_assert(CAP <= 64, "one candidate per lane during compaction");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                                               // [2][RT][KP]
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem + 2 * RT * KP * 4);  // [QB][CAP]
    int* kcnt = reinterpret_cast<int*>(buf + QB * CAP);                                       // [QB]
    float* tau_s = reinterpret_cast<float*>(kcnt + QB);                                       // [QB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int qs = wave * 32 + j;
    const int q = blockIdx.x * QB + qs;
    const int r_begin = first_begin + blockIdx.y * range_len;
    const int r_end = min(r_limit, r_begin + range_len);
    const int out_chunk = out_chunk0 + blockIdx.y;

    float tau = tau_init ? tau_init[(int64_t)q * out_nchunks] : __builtin_inff();  // column 0 = the sample range
    if (h == 0) {
        kcnt[qs] = 0;
        tau_s[qs] = tau;
    }

    // this lane's half of its query row stays in registers for the whole sweep
    float bq[HK];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(Pq + (int64_t)q * KP + h * HK);
#pragma unroll
        for (int m = 0; m < HK / 4; ++m) {
            const f32x4 v = src[m];
            bq[4 * m + 0] = v.x;
            bq[4 * m + 1] = v.y;
            bq[4 * m + 2] = v.z;
            bq[4 * m + 3] = v.w;
        }
    }

    StageRegs<NST> st;
#define BMX_STAGE_LOAD(R0) stage_load(st, reinterpret_cast<const f32x4*>(Pr + (int64_t)(R0) * KP), tid);
#define BMX_STAGE_STORE(SEL) stage_store<NST, TILE_F4>(st, reinterpret_cast<f32x4*>(As + (SEL) * RT * KP), tid);

    BMX_STAGE_LOAD(r_begin)
    BMX_STAGE_STORE(0)
    __syncthreads();

    // lane-private pending list: slot base + kept region + this lane's half
    unsigned long long* pend = buf + qs * CAP + KS + h * PL;
    int mycnt = 0;
    int cur = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += RT) {
        const bool more = r0 + RT < r_end;
        if (more) BMX_STAGE_LOAD(r0 + RT)

        f32x16 acc[RT / 32];
#pragma unroll
        for (int t = 0; t < RT / 32; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

        const float* Ab = As + cur * RT * KP;
#pragma unroll
        for (int m = 0; m < HK / 4; ++m) {
            f32x4 a[RT / 32];
#pragma unroll
            for (int t = 0; t < RT / 32; ++t)
                a[t] = *reinterpret_cast<const f32x4*>(Ab + (t * 32 + j) * KP + h * HK + 4 * m);
#pragma unroll
            for (int t = 0; t < RT / 32; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, bq[4 * m + 0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, bq[4 * m + 1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, bq[4 * m + 2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, bq[4 * m + 3], acc[t], 0, 0, 0);
            }
        }

        // the other LDS buffer is free (its readers passed the last barrier): park the next tile there now so the
        // staging registers are dead during the selection below
        if (more) BMX_STAGE_STORE(cur ^ 1)

        // lane (j, h) now holds, for ITS query j, the values of references r0 + 32 t + (e&3) + 8 (e>>2) + 4 h
#pragma unroll
        for (int t = 0; t < RT / 32; ++t) {
            // cheap tile-level reject: nothing in this lane's 16 values beats the threshold
            float mn = acc[t][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mn = fminf(mn, acc[t][e]);
            if (__builtin_amdgcn_ballot_w64(mn < tau) == 0) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = acc[t][e];
                const bool pass = v < tau;
                if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;
                if (pass) {
                    const int ridx = r0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    pend[mycnt] = ((unsigned long long)f32_orderable(v) << 32) | (uint32_t)ridx;
                    ++mycnt;
                }
                unsigned long long fm = __builtin_amdgcn_ballot_w64(mycnt >= PL);
                if (fm) {
                    fm = (fm | (fm >> 32)) & 0xFFFFFFFFull;  // both K-halves of a query share one slot
                    while (fm) {
                        const int jj = __builtin_ctzll(fm);
                        fm &= fm - 1;
                        compact_slot<KS>(buf, kcnt, tau_s, wave * 32 + jj, jj, lane, mycnt);
                    }
                    tau = tau_s[qs];
                }
            }
        }

        __syncthreads();
        cur ^= 1;
    }

    // final compaction of every slot of this wave, then write candidates + threshold
    for (int jj = 0; jj < 32; ++jj) compact_slot<KS>(buf, kcnt, tau_s, wave * 32 + jj, jj, lane, mycnt);
    for (int jj = 0; jj < 32; ++jj) {
        const int s = wave * 32 + jj;
        const int qq = blockIdx.x * QB + s;
        const int n = kcnt[s];
        if (lane < KS) {
            const unsigned long long key = buf[s * CAP + lane];
            cand[((int64_t)qq * out_nchunks + out_chunk) * KS + lane] = lane < n ? (int32_t)(uint32_t)key : -1;
        }
        // a range that never filled its kept list rejected nothing below its starting threshold
        if (lane == 0) tau_out[(int64_t)qq * out_nchunks + out_chunk] = tau_s[s];
    }
#undef BMX_STAGE_LOAD
#undef BMX_STAGE_STORE
}

// ---------------------------------------------------------------------------------------------------
// 2b. wave-per-workgroup variant: no LDS staging, no barriers.  Each wave owns 32 queries and streams the reference
// tiles straight from L2 into its MFMA A-fragments (fragment-major prepared layout: every load is a coalesced 1 KiB
// read), double-buffered in registers.  Waves never wait for each other, so a wave that is compacting a candidate
// buffer does not stall its neighbours, and 3 waves per SIMD overlap selection with the matrix pipe.
// ---------------------------------------------------------------------------------------------------
template <int KP, int KS>
__global__ __launch_bounds__(64, 3) void knn_topk_w1(const float* __restrict__ Pq, const float* __restrict__ PrF,
                                                     int first_begin, int range_len, int r_limit, int out_chunk0,
                                                     int out_nchunks,
                                                     const unsigned long long* __restrict__ seed_in,
                                                     unsigned long long* __restrict__ seed_out,
                                                     int32_t* __restrict__ cand, float* __restrict__ tau_out) {
    // seed_in  (nullable): [nq_pad][KS + 1] keys of the sample range's kept list (+ its length) -- every range starts
    //                      from it, so its threshold tightens from the first tile on;
    // seed_out (nullable): this launch IS the sample range: write the kept list there instead of cand / tau_out.
    constexpr int CAP = KS + 2 * PL;
    constexpr int HK = KP / 2;
    constexpr int NM = HK / 4;  // 16-byte pieces per lane per 32-row tile
    static_assert(CAP <= 64, "one candidate per lane during compaction");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem);  // [32][CAP]
    int* kcnt = reinterpret_cast<int*>(buf + 32 * CAP);                     // [32]
    float* tau_s = reinterpret_cast<float*>(kcnt + 32);                     // [32]

    const int lane = threadIdx.x;
    const int j = lane & 31, h = lane >> 5;
    const int q = blockIdx.x * 32 + j;
    const int r_begin = first_begin + blockIdx.y * range_len;
    const int r_end = min(r_limit, r_begin + range_len);
    const int out_chunk = out_chunk0 + blockIdx.y;

    float tau = __builtin_inff();
    if (seed_in) {
        for (int jj = 0; jj < 32; ++jj) {
            const unsigned long long* sp = seed_in + ((int64_t)blockIdx.x * 32 + jj) * (KS + 1);
            const int n = (int)sp[KS];
            if (lane < KS) buf[jj * CAP + lane] = sp[lane];
            if (lane == 0) {
                kcnt[jj] = n;
                tau_s[jj] = n >= KS ? orderable_f32((uint32_t)(sp[KS - 1] >> 32)) : __builtin_inff();
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        tau = tau_s[j];
    } else if (h == 0) {
        kcnt[j] = 0;
        tau_s[j] = tau;
    }

    float bq[HK];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(Pq + (int64_t)q * KP + h * HK);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const f32x4 v = src[m];
            bq[4 * m + 0] = v.x;
            bq[4 * m + 1] = v.y;
            bq[4 * m + 2] = v.z;
            bq[4 * m + 3] = v.w;
        }
    }

    unsigned long long* pend = buf + j * CAP + KS + h * PL;
    int mycnt = 0;

    auto tile_ptr = [&](int r0) {
        return reinterpret_cast<const f32x4*>(PrF) + ((int64_t)(r0 >> 5) * NM) * 64 + lane;
    };
    // two register sets: the loads for tile t + 1 are issued before tile t computes (the last prefetch runs into
    // the tail padding of the prepared references)
    f32x4 a0[NM], a1[NM];
#define BMX_LOAD_TILE(A, R0)                                  \
    {                                                         \
        const f32x4* p_ = tile_ptr(R0);                       \
        _Pragma("unroll") for (int m = 0; m < NM; ++m) A[m] = p_[m * 64]; \
    }
    BMX_LOAD_TILE(a0, r_begin)

    auto tile = [&](const f32x4 (&a)[NM], int r0) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, bq[4 * m + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, bq[4 * m + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, bq[4 * m + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, bq[4 * m + 3], acc, 0, 0, 0);
        }
        float mn = acc[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mn = fminf(mn, acc[e]);
        if (__builtin_amdgcn_ballot_w64(mn < tau) == 0) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = acc[e];
            const bool pass = v < tau;
            if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;
            if (pass) {
                const int ridx = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                pend[mycnt] = ((unsigned long long)f32_orderable(v) << 32) | (uint32_t)ridx;
                ++mycnt;
            }
            unsigned long long fm = __builtin_amdgcn_ballot_w64(mycnt >= PL);
            if (fm) {
                fm = (fm | (fm >> 32)) & 0xFFFFFFFFull;
                while (fm) {
                    const int jj = __builtin_ctzll(fm);
                    fm &= fm - 1;
                    compact_slot<KS>(buf, kcnt, tau_s, jj, jj, lane, mycnt);
                }
                tau = tau_s[j];
            }
        }
    };

    for (int r0 = r_begin; r0 < r_end; r0 += 64) {
        BMX_LOAD_TILE(a1, r0 + 32)
        tile(a0, r0);
        BMX_LOAD_TILE(a0, r0 + 64)
        tile(a1, r0 + 32);
    }
#undef BMX_LOAD_TILE

    for (int jj = 0; jj < 32; ++jj) compact_slot<KS>(buf, kcnt, tau_s, jj, jj, lane, mycnt);
    for (int jj = 0; jj < 32; ++jj) {
        const int qq = blockIdx.x * 32 + jj;
        const int n = kcnt[jj];
        if (seed_out) {
            unsigned long long* sp = seed_out + (int64_t)qq * (KS + 1);
            if (lane < KS) sp[lane] = buf[jj * CAP + lane];
            if (lane == 0) sp[KS] = (unsigned long long)n;
            continue;
        }
        if (lane < KS) {
            const unsigned long long key = buf[jj * CAP + lane];
            cand[((int64_t)qq * out_nchunks + out_chunk) * KS + lane] = lane < n ? (int32_t)(uint32_t)key : -1;
        }
        if (lane == 0) tau_out[(int64_t)qq * out_nchunks + out_chunk] = tau_s[jj];
    }
}

// ---------------------------------------------------------------------------------------------------
// 3. refine: exact FP64 re-rank of the candidates + certification
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double exact_d2(const double* __restrict__ a, const double* __restrict__ b, int d) {
    double s = 0.0;
    int c = 0;
    if ((d & 1) == 0) {  // rows of an even number of doubles are 16-byte aligned: half as many load instructions
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* a2 = reinterpret_cast<const d2*>(a);
        const d2* b2 = reinterpret_cast<const d2*>(b);
        for (; c < d; c += 2) {
            const d2 x = a2[c >> 1], y = b2[c >> 1];
            const double t0 = x[0] - y[0];
            s += t0 * t0;  // compiled with -ffp-contract=off: the reference's left-to-right sum, bit for bit
            const double t1 = x[1] - y[1];
            s += t1 * t1;
        }
        return s;
    }
    for (; c < d; ++c) {
        const double t = a[c] - b[c];
        s += t * t;
    }
    return s;
}

__device__ __forceinline__ bool key_less(double da, int ia, double db, int ib) {
    return da < db || (da == db && ia < ib);
}

constexpr int REFINE_MAXM = 64 * 5;  // MAX_CHUNKS * 40

__global__ __launch_bounds__(256) void knn_refine(const double* __restrict__ X, const int32_t* __restrict__ ref_rows,
                                                  const double* __restrict__ Q, const int32_t* __restrict__ q_rows,
                                                  int nq, int d, int k, int KS, int nchunks, int dedupe,
                                                  double eps_k, double eps_qr, double eps_split,
                                                  const int32_t* __restrict__ cand, const float* __restrict__ cand_v,
                                                  const float* __restrict__ tau, const double* __restrict__ qn2,
                                                  const unsigned long long* __restrict__ max_rn2_bits,
                                                  int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                  int32_t* __restrict__ flagged, double* __restrict__ flag_bound) {
    __shared__ double sd[4][REFINE_MAXM];
    __shared__ int si[4][REFINE_MAXM];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;
    const int Mall = nchunks * KS;
    const double* qv = Q + (int64_t)(q_rows ? q_rows[q] : q) * d;
    // gather the valid candidates of all ranges into a dense list (ballot prefix), then work on that list only
    int M = 0;
    float* sv = reinterpret_cast<float*>(&sd[w][0]) + REFINE_MAXM;  // upper half of this wave's sd row: approx values
    for (int m0 = 0; m0 < Mall; m0 += 64) {
        const int m = m0 + lane;
        const int id = m < Mall ? cand[(int64_t)q * Mall + m] : -1;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(id >= 0);
        const int pos = M + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (id >= 0) {
            si[w][pos] = id;
            if (cand_v) sv[pos] = cand_v[(int64_t)q * Mall + m];
        }
        M += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // several ranges: only the KS best by approximate value can matter; the rest count as rejected with the
    // (KS+1)-th smallest approximate value as their bound, which enters the certificate below
    float tmerge = __builtin_inff();
    if (cand_v && M > KS) {
        constexpr int NU = (REFINE_MAXM + 63) / 64;
        int keep_pos[NU], ids[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int m = lane + 64 * u;
            keep_pos[u] = -1;
            ids[u] = 0;
            if (m < M) {
                const float vm = sv[m];
                const int im = si[w][m];
                int rank = 0;
                for (int f = 0; f < M; ++f) {
                    const float vf = sv[f];
                    rank += (vf < vm || (vf == vm && si[w][f] < im)) ? 1 : 0;
                }
                keep_pos[u] = rank < KS ? rank : -1;
                ids[u] = im;
                if (rank == KS) tmerge = vm;  // the first rejected one bounds all rejected ones from below
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (keep_pos[u] >= 0) si[w][keep_pos[u]] = ids[u];
        for (int o = 32; o > 0; o >>= 1) tmerge = fminf(tmerge, __shfl_xor(tmerge, o));
        M = KS;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    for (int m = lane; m < M; m += 64) {
        const int id = si[w][m];
        sd[w][m] = exact_d2(qv, X + (int64_t)(ref_rows ? ref_rows[id] : id) * d, d);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (dedupe && nchunks > 1) {
        // seeded ranges can carry the same sample reference in several lists: keep the first copy only
        for (int m = lane; m < M; m += 64) {
            const int im = si[w][m];
            bool dup = false;
            for (int f = 0; f < m; ++f) dup |= si[w][f] == im;
            if (dup) sd[w][m] = __builtin_inf();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int m = lane; m < M; m += 64)
            if (sd[w][m] == __builtin_inf()) si[w][m] = 0x7FFFFFFF;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    double kth = M >= k ? 0.0 : __builtin_inf();  // fewer than k candidates: never certified
    for (int m = lane; m < M; m += 64) {
        const double dm = sd[w][m];
        const int im = si[w][m];
        int rank = 0;
        for (int f = 0; f < M; ++f) rank += key_less(sd[w][f], si[w][f], dm, im) ? 1 : 0;
        if (rank < k && im != 0x7FFFFFFF) {
            idx_out[(int64_t)q * k + rank] = im;
            if (dist_out) dist_out[(int64_t)q * k + rank] = sqrt(dm);
        }
        if (rank == k - 1) kth = dm;
    }
    // certification: every rejected reference has  v >= tau_c, i.e. approx d2 >= tau_c + |q~|^2, and the f32 path
    // is within eps of the exact value, so the top k is proven when  kth < min_c tau_c + |q~|^2 - eps.
    float tmin = tmerge;
    for (int c = lane; c < nchunks; c += 64) tmin = fminf(tmin, tau[(int64_t)q * nchunks + c]);
    for (int o = 32; o > 0; o >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, o));
        kth = fmax(kth, __shfl_xor(kth, o));
    }
    if (lane == 0) {
        const double qn = sqrt(qn2[q]);
        const double rm = sqrt(__longlong_as_double((long long)*max_rn2_bits));
        const double u = 5.9604644775390625e-8;  // 2^-24
        // f32 rounding of the centred coordinates + accumulation over eps_k terms (+ the dropped split-bf16 terms)
        const double eps = 1.5 * (u * (2.0 * (qn + rm) * (qn + rm) + (eps_k + 1.0) * (rm * rm + eps_qr * qn * rm)) +
                                  eps_split * qn * rm);
        const bool proven = kth < (double)tmin + qn2[q] - eps;  // tmin = +inf when nothing was ever rejected
        if (!proven) {
            const int pos = atomicAdd(&flagged[0], 1);
            flagged[1 + pos] = q;
            flag_bound[pos] = kth;  // the true k-th neighbour is no farther than the k-th candidate
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 4. exact FP64 scan for flagged queries (or every query when the MFMA path does not apply), in batches:
//    knn_exact_dist   -- grid (reference blocks, queries of the batch): all exact squared distances, spread over the
//                        whole chip even when only a handful of queries are flagged;
//    knn_exact_select -- one workgroup per query: k rounds of block-wide (distance, index) minimum.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_exact_dist(const double* __restrict__ X, const int32_t* __restrict__ ref_rows,
                                                      int nr, const double* __restrict__ Q,
                                                      const int32_t* __restrict__ q_rows, int d,
                                                      const int32_t* __restrict__ flagged, int f0,
                                                      double* __restrict__ drow) {
    const int f = f0 + blockIdx.y;
    const int q = flagged ? flagged[1 + f] : f;
    const double* qv = Q + (int64_t)(q_rows ? q_rows[q] : q) * d;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < nr) drow[(int64_t)blockIdx.y * nr + r] = exact_d2(qv, X + (int64_t)(ref_rows ? ref_rows[r] : r) * d, d);
}

__global__ __launch_bounds__(256) void knn_exact_select(const double* __restrict__ drow, int nr, int k,
                                                        const int32_t* __restrict__ flagged, int f0,
                                                        int32_t* __restrict__ idx_out, double* __restrict__ dist_out) {
    __shared__ double rd[256];
    __shared__ int ri[256];
    const int tid = threadIdx.x;
    const int f = f0 + blockIdx.x;
    const int q = flagged ? flagged[1 + f] : f;
    const double* row = drow + (int64_t)blockIdx.x * nr;
    double last_d = -1.0;  // squared distances are >= 0
    int last_i = -1;
    for (int jdx = 0; jdx < k; ++jdx) {
        double bd = __builtin_inf();
        int bi = 0x7FFFFFFF;
        for (int r = tid; r < nr; r += 256) {
            const double v = row[r];
            if (key_less(last_d, last_i, v, r) && key_less(v, r, bd, bi)) {
                bd = v;
                bi = r;
            }
        }
        rd[tid] = bd;
        ri[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o && key_less(rd[tid + o], ri[tid + o], rd[tid], ri[tid])) {
                rd[tid] = rd[tid + o];
                ri[tid] = ri[tid + o];
            }
            __syncthreads();
        }
        last_d = rd[0];
        last_i = ri[0];
        if (tid == 0) {
            idx_out[(int64_t)q * k + jdx] = last_i;
            if (dist_out) dist_out[(int64_t)q * k + jdx] = sqrt(last_d);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// 4b. fast exact path for a FEW flagged queries: the k-th candidate distance bounds the true k-th neighbour from
//     above, so one pass that stages each reference tile once in LDS, evaluates it against every flagged query in
//     FP64 and keeps the references within that bound (a handful per query) replaces the full per-query rescan.
//     A query whose list overflows (massive exact ties) is handed to the full scan.
// ---------------------------------------------------------------------------------------------------
constexpr int XF_TILE = 64;    // reference rows per staged tile
constexpr int XF_CAP = 256;    // kept references per flagged query

__global__ __launch_bounds__(256) void knn_exact_filter(const double* __restrict__ X,
                                                        const int32_t* __restrict__ ref_rows, int nr,
                                                        const double* __restrict__ Q,
                                                        const int32_t* __restrict__ q_rows, int d,
                                                        const int32_t* __restrict__ flagged,
                                                        const double* __restrict__ flag_bound, int nflag,
                                                        int32_t* __restrict__ xcnt, double* __restrict__ xd,
                                                        int32_t* __restrict__ xi) {
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    double* xs = reinterpret_cast<double*>(smem_x);  // [XF_TILE][d + 1]  (+1: breaks the bank stride)
    const int ld = d + 1;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * XF_TILE;
    const int rows_here = min(XF_TILE, nr - r0);
    for (int e = tid; e < rows_here * d; e += 256) {
        const int rr = e / d, c = e - rr * d;
        const int64_t row = ref_rows ? ref_rows[r0 + rr] : r0 + rr;
        xs[rr * ld + c] = X[row * d + c];
    }
    __syncthreads();
    const int rr = tid & (XF_TILE - 1);
    if (rr >= rows_here) return;
    const double* xr = xs + rr * ld;
    for (int f = tid / XF_TILE; f < nflag; f += 256 / XF_TILE) {
        const int q = flagged[1 + f];
        const double* qv = Q + (int64_t)(q_rows ? q_rows[q] : q) * d;
        double s = 0.0;
        for (int c = 0; c < d; ++c) {
            const double t = qv[c] - xr[c];
            s += t * t;
        }
        if (s <= flag_bound[f]) {
            const int pos = atomicAdd(&xcnt[f], 1);
            if (pos < XF_CAP) {
                xd[(int64_t)f * XF_CAP + pos] = s;
                xi[(int64_t)f * XF_CAP + pos] = r0 + rr;
            }
        }
    }
}

// one wave per flagged query: exact (distance, index) ranking of its short list; overflowed lists go to `slow`
__global__ __launch_bounds__(256) void knn_exact_pick(const int32_t* __restrict__ flagged, int nflag, int k,
                                                      const int32_t* __restrict__ xcnt, const double* __restrict__ xd,
                                                      const int32_t* __restrict__ xi, int32_t* __restrict__ idx_out,
                                                      double* __restrict__ dist_out, int32_t* __restrict__ slow) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= nflag) return;
    const int q = flagged[1 + f];
    const int n = xcnt[f];
    if (n > XF_CAP || n < k) {  // n < k cannot happen (the k candidates themselves qualify); belt and braces
        if (lane == 0) {
            const int pos = atomicAdd(&slow[0], 1);
            slow[1 + pos] = q;
        }
        return;
    }
    const double* dd = xd + (int64_t)f * XF_CAP;
    const int32_t* ii = xi + (int64_t)f * XF_CAP;
    for (int m = lane; m < n; m += 64) {
        const double dm = dd[m];
        const int im = ii[m];
        int rank = 0;
        for (int t = 0; t < n; ++t) rank += key_less(dd[t], ii[t], dm, im) ? 1 : 0;
        if (rank < k) {
            idx_out[(int64_t)q * k + rank] = im;
            if (dist_out) dist_out[(int64_t)q * k + rank] = sqrt(dm);
        }
    }
}

__global__ void fill_u32(uint32_t* __restrict__ p, int n, uint32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void accumulate_flagged(const int32_t* __restrict__ flagged, unsigned long long* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total += (unsigned long long)flagged[0];
}

struct TopkLaunch {
    const float* pq;
    const float* pr;
    int nqb;          // query blocks
    int first_begin;  // first prepared reference row of range 0
    int range_len;    // rows per range (tile multiple)
    int nranges;
    int r_limit;      // end of the last range
    int out_chunk0;   // candidate-list column of range 0
    int out_nchunks;  // candidate-list columns in total
    const float* tau_init;
    int32_t* cand;
    float* tau;
    int variant;      // 0: workgroup-shared LDS staging, 1: wave-per-workgroup streaming
    int lds_pad;      // extra dynamic LDS requested by variant 1 to cap resident waves per CU
    const unsigned long long* seed_in = nullptr;  // variant 1
    unsigned long long* seed_out = nullptr;       // variant 1
};

template <int KP, int KS>
size_t topk_lds_bytes() {
    return (size_t)2 * RT * KP * 4 + (size_t)QB * (KS + 2 * PL) * 8 + QB * 4 + QB * 4;
}

template <int KP, int KS>
void launch_topk(hipStream_t stream, KnnWorkspace& ws, const TopkLaunch& L) {
    const size_t lds = topk_lds_bytes<KP, KS>();
    static bool attr_set = false;
    if (!attr_set) {
        BMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_topk_mfma<KP, KS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (ws.profile) {
        ev = ws.next_events();
        BMX_HIP(hipEventRecord(ev.first, stream));
    }
    if (L.variant == 1) {
        const size_t lds1 = (size_t)32 * (KS + 2 * PL) * 8 + 256 + (size_t)L.lds_pad;
        static size_t attr1 = 0;
        if (lds1 > attr1) {
            BMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_topk_w1<KP, KS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
            attr1 = lds1;
        }
        hipLaunchKernelGGL((knn_topk_w1<KP, KS>), dim3(L.nqb, L.nranges), dim3(64), lds1, stream, L.pq, L.pr,
                           L.first_begin, L.range_len, L.r_limit, L.out_chunk0, L.out_nchunks, L.seed_in, L.seed_out,
                           L.cand, L.tau);
    } else {
        hipLaunchKernelGGL((knn_topk_mfma<KP, KS>), dim3(L.nqb, L.nranges), dim3(THREADS), lds, stream, L.pq, L.pr,
                           L.first_begin, L.range_len, L.r_limit, L.out_chunk0, L.out_nchunks, L.tau_init, L.cand,
                           L.tau);
    }
    BMX_LAUNCH_CHECK();
    if (ws.profile) BMX_HIP(hipEventRecord(ev.second, stream));
}

template <int KS>
bool dispatch_kp(int KP, hipStream_t s, KnnWorkspace& ws, const TopkLaunch& L) {
    switch (KP) {
#define BMX_KP_CASE(V)                      \
    case V:                                 \
        launch_topk<V, KS>(s, ws, L);       \
        return true;
        BMX_KP_CASE(8)
        BMX_KP_CASE(16)
        BMX_KP_CASE(24)
        BMX_KP_CASE(32)
        BMX_KP_CASE(40)
        BMX_KP_CASE(56)
        BMX_KP_CASE(64)
        BMX_KP_CASE(80)
        BMX_KP_CASE(104)
        BMX_KP_CASE(128)
#undef BMX_KP_CASE
        default:
            return false;
    }
}

size_t topk_lds_for(int KP, int KS) { return (size_t)2 * RT * KP * 4 + (size_t)QB * (KS + 2 * PL) * 8 + QB * 8; }

int pick_kp(int d) {
    static const int opts[] = {8, 16, 24, 32, 40, 56, 64, 80, 104, 128};
    for (int o : opts)
        if (d + 1 <= o) return o;
    return 0;
}

}  // namespace

void knn_device(hipStream_t stream, KnnWorkspace& ws, const double* X, const int32_t* ref_rows, int nr,
                const double* Q, const int32_t* q_rows, int nq_total, int d, int k, int32_t* idx_out,
                double* dist_out, int q_begin, int q_end) {
    (void)nq_total;
    const int nq = q_end - q_begin;
    if (nq <= 0 || k <= 0) return;
    if (k > nr) throw Error(BMX_ERR_ARG, "kNN: k exceeds the number of reference cells");
    // sub-range of the query list
    const double* Qs = Q;
    const int32_t* qrs = q_rows;
    if (q_rows)
        qrs = q_rows + q_begin;
    else
        Qs = Q + (int64_t)q_begin * d;
    int32_t* io = idx_out + (int64_t)q_begin * k;
    double* dout = dist_out ? dist_out + (int64_t)q_begin * k : nullptr;

    const int KP = pick_kp(d);
    const int KS = k <= 20 ? 24 : (k <= 36 ? 40 : 0);
    const bool use_mfma = !ws.force_exact && KP != 0 && KS != 0 && nr > 2 * KS;

    int32_t* flagged = ws.flagged.reserve((size_t)nq + 1);
    BMX_HIP(hipMemsetAsync(flagged, 0, sizeof(int32_t), stream));

    if (use_mfma) {
        // candidate-pass variant: 2 = split-bf16 MFMA with an LDS ring (default), 1 = f32 MFMA, one wave per
        // workgroup, 0 = f32 MFMA with workgroup-shared LDS staging.  BMX_TOPK_VARIANT overrides (A/B runs).
        static const int requested = [] {
            const char* v = std::getenv("BMX_TOPK_VARIANT");
            return v ? std::atoi(v) : 2;
        }();
        const int NS = bf16_pick_ns(d);
        int variant = requested;
        if (variant == 2 && (NS == 0 || (KS == 40 && NS > 16))) variant = 1;
        ws.last_variant = variant;
        const int ncons = variant == 2 ? bf16_ncons(NS, KS) : 0;
        const int unit = variant == 2 ? 32 * ncons : (variant == 1 ? 32 : QB);  // queries per workgroup
        const int nq_pad = (int)round_up(nq, variant == 2 ? unit : 256);
        const int nqb = nq_pad / unit;
        const int rmul = variant == 2 ? 32 : RT;  // range lengths are multiples of the kernels' tile step

        // Reference ranges.  A short sample range [0, S) runs first; its kept list seeds every main range (variants
        // 1, 2) or at least hands it a valid starting threshold (variant 0), so selection is tight from the first
        // tile.  C ranges (and for variant 1 the resident workgroups per CU, W, capped through the LDS request) are
        // picked so that (query blocks x C) workgroups fill the resident slots in whole rounds; every extra range
        // repeats a little selection work, hence the small penalties.
        const int S = nr >= 32768 ? (std::getenv("BMX_SAMPLE") ? std::atoi(std::getenv("BMX_SAMPLE")) : 4096) : 0;
        int C = 1, W = 0;
        {
            int w_lo = 0, w_hi = 0, fixed_slots = 0;
            if (variant == 1) {  // VGPRs (100 up to KP 64, 168 above) and LDS bound the resident waves per CU
                w_hi = std::min(KP <= 64 ? 16 : 12, (160 * 1024) / (32 * (KS + 2 * PL) * 8 + 256));
                w_lo = std::min(8, w_hi);
            } else if (variant == 2) {
                fixed_slots = 256;  // the ring kernel takes a CU's whole LDS: one workgroup per CU
            } else {
                fixed_slots = topk_lds_for(KP, KS) <= 80 * 1024 ? 512 : 256;
            }
            double best = -1.0;
            for (int w = w_hi; w >= w_lo; --w)
                for (int c = 1; c <= MAX_CHUNKS - 1; ++c) {
                    if (c > 1 && (nr - S) / c < 2048) break;
                    const int slots = variant == 1 ? w * 256 : fixed_slots;
                    const double rounds = (double)nqb * c / slots;
                    const double eff = rounds >= 1.0 ? rounds / std::ceil(rounds) : rounds;
                    // measured on 100k x 100k: each extra range costs the split-bf16 kernel ~0.1 of a 100k-row sweep
                    // (selection restarts from the sample threshold, bigger refine) -- a fixed cost, so relatively
                    // less for longer reference sets; the f32 kernels ~0.015 of a sweep
                    const double per_range = variant == 2 ? 0.03 * std::min(1.0, 1.0e5 / (double)nr) : 0.015;
                    // a lone range that needs a second, partly filled round measured ~12 % slower than its round count says
                    const double lone = (variant == 2 && c == 1 && rounds > 1.0 && rounds < 2.0) ? 0.12 : 0.0;
                    const double score = eff - per_range * c - lone - 0.01 * (w_hi - w);
                    if (score > best) {
                        best = score;
                        C = c;
                        W = w;
                    }
                }
        }
        // Split-bf16 ring kernel: one workgroup per CU.  The query blocks that fill whole rounds of 256 workgroups
        // sweep the reference as ONE range (tightest thresholds, one list per query); only the remaining b blocks are
        // split into c ranges, chosen so that their b*c short items fill the last round evenly.  Measured work per
        // pair evaluation relative to one range (100k x 400k): 3 ranges 1.12, 5: 1.17, 7: 1.21 ~ 1 + 0.105 ln c.
        int n_full = 0;
        if (variant == 2) {
            const int a = nqb / 256, b = nqb % 256;
            n_full = a * 256;
            C = 1;
            if (b > 0) {
                double best = 1e30;
                for (int c = 1; c <= MAX_CHUNKS - 1; ++c) {
                    if (c > 1 && nr / c < 2048) break;
                    const double tail = std::ceil((double)b * c / 256.0) / c * (1.0 + 0.105 * std::log((double)c));
                    if (tail < best - 1e-9) {
                        best = tail;
                        C = c;
                    }
                }
            }
        }
        if (variant == 2 && std::getenv("BMX_SPLIT_C")) C = std::max(1, std::atoi(std::getenv("BMX_SPLIT_C")));
        if (std::getenv("BMX_FORCE_C")) {
            C = std::atoi(std::getenv("BMX_FORCE_C"));
            n_full = 0;
        }
        const int main_rows = variant == 2 ? nr : nr - S;  // the bf16 full pass rescans the sample rows
        const int chunk_len = (int)round_up(cdiv(main_rows, C), rmul);
        C = std::max(1, cdiv(main_rows, chunk_len));
        const int nr_pad = (variant == 2 ? 0 : S) + chunk_len * C;
        const bool seeded = variant == 1 && S > 0;
        const int nchunks = C + (S > 0 && variant == 0 ? 1 : 0);  // only variant 0 keeps the sample as its own column
        if (std::getenv("BMX_DEBUG"))
            fprintf(stderr, "[bmx] knn nq=%d nr=%d d=%d KS=%d variant=%d S=%d C=%d W=%d chunk=%d full-range blocks=%d of %d\n",
                    nq, nr, d, KS, variant, S, C, W, chunk_len, C > 1 ? n_full : nqb, nqb);
        int lds_pad = 0;
        if (variant == 1) {
            const int base = 32 * (KS + 2 * PL) * 8 + 256;
            lds_pad = std::max(0, (160 * 1024) / W - 512 - base);  // floor(160 KiB / request) == W
        }

        const int KPw = variant == 2 ? 8 * NS : KP;  // prepared row width in 4-byte words
        float* pq = ws.pq.reserve((size_t)nq_pad * KPw);
        float* pr = ws.pr.reserve((size_t)(nr_pad + 4 * RT) * KPw);  // + tail padding: the prefetches over-read
        double* qn2 = ws.qn2.reserve(nq_pad);
        double* rn2 = ws.rn2.reserve(nr_pad);
        double* mean = ws.mean.reserve((size_t)d + 2);
        unsigned long long* maxbits = reinterpret_cast<unsigned long long*>(mean + d);
        int32_t* cand = ws.cand.reserve((size_t)nq_pad * nchunks * KS);
        float* tau = ws.tau.reserve((size_t)nq_pad * nchunks);
        unsigned long long* seed =
            seeded ? reinterpret_cast<unsigned long long*>(ws.seed.reserve((size_t)nq_pad * (KS + 1))) : nullptr;

        // centre of the reference: any vector is valid (the error bound uses the norms actually obtained), a point
        // near the mean keeps it tight -- the mean of a strided sample of <= 16k rows costs next to nothing
        const int cstride = std::max(1, nr / 16384);
        const int ncs = cdiv(nr, cstride);
        const int rpb = 256;
        const int nb = cdiv(ncs, rpb);
        double* red = ws.red.reserve((size_t)nb * d);
        hipLaunchKernelGGL(colsum_partial, dim3(nb), dim3(256), 0, stream, X, ref_rows, ncs, d, rpb, cstride, red);
        BMX_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final, dim3(cdiv(d, 64)), dim3(64), 0, stream, red, nb, d, 1.0 / ncs, mean);
        BMX_LAUNCH_CHECK();
        BMX_HIP(hipMemsetAsync(maxbits, 0, sizeof(unsigned long long), stream));

        double eps_k, eps_qr, eps_split;
        const float* cand_v = nullptr;
        if (variant == 2) {
            bf16_prep(stream, X, ref_rows, nr, nr_pad, d, NS, mean, 0, reinterpret_cast<uint16_t*>(pr), rn2, maxbits,
                      ws.maxslots.reserve(64 * 16));
            bf16_prep(stream, Qs, qrs, nq, nq_pad, d, NS, mean, 1, reinterpret_cast<uint16_t*>(pq), qn2, maxbits,
                      ws.maxslots.p);
            // sample pass: threshold estimation over rows [0, S); full pass: every row, starting from that threshold
            uint32_t* tau_g = ws.tau_g.reserve(nq_pad);
            if (S == 0) {  // no sample: +inf everywhere (0xFF800000 is the orderable image of +inf)
                hipLaunchKernelGGL(fill_u32, dim3(cdiv(nq_pad, 256)), dim3(256), 0, stream, tau_g, nq_pad, 0xFF800000u);
                BMX_LAUNCH_CHECK();
            }
            Bf16Launch L{reinterpret_cast<const uint16_t*>(pq), reinterpret_cast<const uint16_t*>(pr), nqb, 0, S, 1, S,
                         0, nchunks, tau_g, 1, cand, nchunks > 1 ? ws.cand_v.reserve((size_t)nq_pad * nchunks * KS) : nullptr,
                         tau};
            cand_v = L.cand_v;
            bool ok = true;
            if (S > 0) ok = bf16_launch(stream, ws, NS, KS, L);
            L.first_begin = 0;
            L.range_len = chunk_len;
            L.nranges = C;
            L.n_full = C > 1 ? n_full : 0;
            L.r_limit = nr_pad;
            L.sample = 0;
            ok = ok && bf16_launch(stream, ws, NS, KS, L);
            if (!ok) throw Error(BMX_ERR_ARG, "kNN: unsupported padded dimension");
            eps_k = 16.0 * NS;                                   // f32 accumulation over the concatenated K
            eps_qr = 6.0;                                        // three product blocks, each <= 2 |q||r|
            eps_split = 3.03 * 2.0 * 1.52587890625e-05;          // dropped ql.rl, qh.r3, q3.rh: 3.03 * 2^-16 * 2|q||r|
        } else {
            hipLaunchKernelGGL(knn_prep, dim3(cdiv(nr_pad, 256)), dim3(256), 0, stream, X, ref_rows, nr, nr_pad, d, KP,
                               mean, 0, variant == 1 ? 1 : 0, pr, rn2, maxbits);
            BMX_LAUNCH_CHECK();
            hipLaunchKernelGGL(knn_prep, dim3(cdiv(nq_pad, 256)), dim3(256), 0, stream, Qs, qrs, nq, nq_pad, d, KP, mean,
                               1, 0, pq, qn2, maxbits);
            BMX_LAUNCH_CHECK();
            TopkLaunch L{pq, pr, nqb, 0, S, 1, S, 0, nchunks, nullptr, cand, tau, variant, lds_pad};
            if (seeded) L.seed_out = seed;
            bool ok = true;
            if (S > 0) ok = KS == 24 ? dispatch_kp<24>(KP, stream, ws, L) : dispatch_kp<40>(KP, stream, ws, L);
            L.first_begin = S;
            L.range_len = chunk_len;
            L.nranges = C;
            L.r_limit = nr_pad;
            L.out_chunk0 = S > 0 && !seeded ? 1 : 0;
            L.tau_init = S > 0 && !seeded ? tau : nullptr;  // column 0 of tau[q][nchunks]: read with stride nchunks
            L.seed_in = seed;
            L.seed_out = nullptr;
            ok = ok && (KS == 24 ? dispatch_kp<24>(KP, stream, ws, L) : dispatch_kp<40>(KP, stream, ws, L));
            if (!ok) throw Error(BMX_ERR_ARG, "kNN: unsupported padded dimension");
            eps_k = KP;
            eps_qr = 2.0;
            eps_split = 0.0;
        }

        hipLaunchKernelGGL(knn_refine, dim3(cdiv(nq, 4)), dim3(256), 0, stream, X, ref_rows, Qs, qrs, nq, d, k, KS,
                           nchunks, seeded ? 1 : 0, eps_k, eps_qr, eps_split, cand, cand_v, tau, qn2, maxbits, io, dout,
                           flagged,
                           ws.flag_bound.reserve((size_t)nq + 1));
        BMX_LAUNCH_CHECK();
        if (std::getenv("BMX_DEBUG")) {
            std::vector<int32_t> hc((size_t)nq * nchunks * KS);
            BMX_HIP(hipMemcpyAsync(hc.data(), cand, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            BMX_HIP(hipStreamSynchronize(stream));
            size_t valid = 0;
            for (int32_t v : hc) valid += v >= 0;
            fprintf(stderr, "[bmx] candidates per query after the top-k pass: %.1f (of %d slots)\n", (double)valid / nq,
                    nchunks * KS);
        }
    }

    // exact path: flagged queries, or everything when the MFMA path does not apply.  The number of flagged queries
    // decides the launch shape, so it is read back here (one small synchronisation per search).
    {
        int count = nq;
        if (use_mfma) {
            int32_t h = 0;
            BMX_HIP(hipMemcpyAsync(&h, flagged, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            BMX_HIP(hipStreamSynchronize(stream));
            count = h;
        }
        const int32_t* scan_list = use_mfma ? flagged : nullptr;
        if (count > 0 && use_mfma) {
            // few flagged queries: bounded filter pass, then rank the short lists; overflows fall through
            int32_t* xcnt = ws.xcnt.reserve((size_t)count);
            double* xd = ws.xd.reserve((size_t)count * XF_CAP);
            int32_t* xi = ws.xi.reserve((size_t)count * XF_CAP);
            int32_t* slow = ws.slow.reserve((size_t)count + 1);
            BMX_HIP(hipMemsetAsync(xcnt, 0, (size_t)count * sizeof(int32_t), stream));
            BMX_HIP(hipMemsetAsync(slow, 0, sizeof(int32_t), stream));
            const size_t lds = (size_t)XF_TILE * (d + 1) * sizeof(double);
            hipLaunchKernelGGL(knn_exact_filter, dim3(cdiv(nr, XF_TILE)), dim3(256), lds, stream, X, ref_rows, nr, Qs, qrs,
                               d, flagged, ws.flag_bound.p, count, xcnt, xd, xi);
            BMX_LAUNCH_CHECK();
            hipLaunchKernelGGL(knn_exact_pick, dim3(cdiv(count, 4)), dim3(256), 0, stream, flagged, count, k, xcnt, xd, xi,
                               io, dout, slow);
            BMX_LAUNCH_CHECK();
            int32_t h = 0;
            BMX_HIP(hipMemcpyAsync(&h, slow, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            BMX_HIP(hipStreamSynchronize(stream));
            count = h;
            scan_list = slow;
        }
        if (count > 0) {
            const size_t budget = (size_t)512 << 20;
            // grid.y carries the queries of a batch: at most 65535 of them
            const int batch =
                (int)std::min<size_t>({std::max<size_t>(1, budget / ((size_t)nr * 8)), (size_t)count, (size_t)65535});
            double* drow = ws.drow.reserve((size_t)batch * nr);
            for (int f0 = 0; f0 < count; f0 += batch) {
                const int nb = std::min(batch, count - f0);
                hipLaunchKernelGGL(knn_exact_dist, dim3(cdiv(nr, 256), nb), dim3(256), 0, stream, X, ref_rows, nr, Qs, qrs,
                                   d, scan_list, f0, drow);
                BMX_LAUNCH_CHECK();
                hipLaunchKernelGGL(knn_exact_select, dim3(nb), dim3(256), 0, stream, drow, nr, k, scan_list, f0, io,
                                   dout);
                BMX_LAUNCH_CHECK();
            }
        }
    }
    if (ws.flag_total && use_mfma) {
        hipLaunchKernelGGL(accumulate_flagged, dim3(1), dim3(64), 0, stream, flagged, ws.flag_total);
        BMX_LAUNCH_CHECK();
    }
}

}  // namespace bmx
