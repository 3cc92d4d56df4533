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

#include <algorithm>
#include <cmath>

namespace bmx {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int QB = 128;          // queries per workgroup: 4 waves x one 32-query MFMA column tile
constexpr int RT = 64;           // references per staged LDS tile (two 32-row MFMA tiles)
constexpr int THREADS = 256;
constexpr int SLACK = 24;        // candidate-buffer slots beyond KS
constexpr int MAX_CHUNKS = 8;

__device__ __forceinline__ uint32_t f32_orderable(float v) {
    uint32_t u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o) {
    uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}

// ---------------------------------------------------------------------------------------------------
// column sums over a row list (two deterministic stages)
// ---------------------------------------------------------------------------------------------------
__global__ void colsum_partial(const double* __restrict__ X, const int32_t* __restrict__ rows, int n, int d,
                               int rows_per_block, double* __restrict__ partial) {
    __shared__ double sm[4][64];
    const int c0 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    for (int cb = 0; cb < d; cb += 64) {
        const int c = cb + c0;
        double s = 0.0;
        if (c < d)
            for (int r = r0 + rl; r < r1; r += 4) {
                const int64_t row = rows ? rows[r] : r;
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
__global__ void knn_prep(const double* __restrict__ X, const int32_t* __restrict__ rows, int n, int n_pad, int d,
                         int KP, const double* __restrict__ mean, int is_query, float* __restrict__ P,
                         double* __restrict__ n2, unsigned long long* __restrict__ max_n2_bits) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    float* out = P + (int64_t)r * KP;
    if (r >= n) {
        for (int c = 0; c < KP; ++c) out[c] = 0.f;
        if (!is_query) out[d] = __builtin_inff();  // padded references can never pass a threshold
        return;
    }
    const int64_t row = rows ? rows[r] : r;
    const double* x = X + row * d;
    double s = 0.0;
    for (int c = 0; c < d; ++c) {
        const float f = (float)(x[c] - mean[c]);
        s += (double)f * (double)f;
        out[c] = is_query ? -2.f * f : f;
    }
    out[d] = is_query ? 1.f : (float)s;
    for (int c = d + 1; c < KP; ++c) out[c] = 0.f;
    n2[r] = s;
    if (!is_query) atomicMax(max_n2_bits, (unsigned long long)__double_as_longlong(s));
}

// ---------------------------------------------------------------------------------------------------
// 2. MFMA distance tiles + per-query threshold / buffer selection
// ---------------------------------------------------------------------------------------------------
// One wave compacts the candidate buffer of query slot `qs` (n <= 64 entries, one per lane): rank by counting over
// the unique 64-bit keys, keep the KS smallest in sorted order, publish the new threshold.
template <int KS, int CAP>
__device__ __forceinline__ void compact_slot(unsigned long long* buf, int* cnt, float* tau_s, int qs, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int n = cnt[qs];
    unsigned long long* b = buf + qs * CAP;
    const unsigned long long key = lane < n ? b[lane] : ~0ull;
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    int rank = 0;
    for (int f = 0; f < n; ++f) {
        const uint32_t flo = __builtin_amdgcn_readlane(klo, f);
        const uint32_t fhi = __builtin_amdgcn_readlane(khi, f);
        const unsigned long long fk = ((unsigned long long)fhi << 32) | flo;
        rank += fk < key ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < n && rank < KS) b[rank] = key;
    if (n >= KS && lane < n && rank == KS - 1) tau_s[qs] = orderable_f32(khi);
    if (lane == 0) cnt[qs] = n < KS ? n : KS;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

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

template <int KP, int KS>
__global__ __launch_bounds__(THREADS, 2) void knn_topk_mfma(const float* __restrict__ Pq, const float* __restrict__ Pr,
                                                            int nq_pad, int nr_pad, int chunk_len, int nchunks,
                                                            int32_t* __restrict__ cand, float* __restrict__ tau_out) {
    constexpr int CAP = KS + SLACK;
    constexpr int TRIG = CAP - 2;  // at most two lanes (the two K-halves of a query) append per register step
    constexpr int HK = KP / 2;     // K elements per lane half
    constexpr int TILE_F4 = RT * KP / 4;
    constexpr int NST = (TILE_F4 + THREADS - 1) / THREADS;
    static_assert(CAP <= 64, "one candidate per lane during compaction");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                                               // [2][RT][KP]
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem + 2 * RT * KP * 4);  // [QB][CAP]
    int* cnt = reinterpret_cast<int*>(buf + QB * CAP);                                        // [QB]
    float* tau_s = reinterpret_cast<float*>(cnt + QB);                                        // [QB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int qs = wave * 32 + j;
    const int q = blockIdx.x * QB + qs;
    const int chunk = blockIdx.y;
    const int r_begin = chunk * chunk_len;
    const int r_end = min(nr_pad, r_begin + chunk_len);

    if (tid < QB) {
        cnt[tid] = 0;
        tau_s[tid] = __builtin_inff();
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

    float tau = __builtin_inff();
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
                bool flush = false;
                if (pass) {
                    const int ridx = r0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int pos = atomicAdd(&cnt[qs], 1);
                    buf[qs * CAP + pos] = ((unsigned long long)f32_orderable(v) << 32) | (uint32_t)ridx;
                    flush = pos + 1 >= TRIG;
                }
                unsigned long long fm = __builtin_amdgcn_ballot_w64(flush);
                if (fm) {
                    fm = (fm | (fm >> 32)) & 0xFFFFFFFFull;  // both K-halves of a query share one slot
                    while (fm) {
                        const int jj = __builtin_ctzll(fm);
                        fm &= fm - 1;
                        compact_slot<KS, CAP>(buf, cnt, tau_s, wave * 32 + jj, lane);
                    }
                    tau = tau_s[qs];
                }
            }
        }

        __syncthreads();
        cur ^= 1;
    }

    // final compaction of every slot of this wave, then write candidates + threshold
    for (int jj = 0; jj < 32; ++jj) compact_slot<KS, CAP>(buf, cnt, tau_s, wave * 32 + jj, lane);
    for (int jj = 0; jj < 32; ++jj) {
        const int s = wave * 32 + jj;
        const int qq = blockIdx.x * QB + s;
        const int n = cnt[s];
        if (lane < KS) {
            const unsigned long long key = buf[s * CAP + lane];
            cand[((int64_t)qq * nchunks + chunk) * KS + lane] = lane < n ? (int32_t)(uint32_t)key : -1;
        }
        if (lane == 0) tau_out[(int64_t)qq * nchunks + chunk] = n >= KS ? tau_s[s] : __builtin_inff();
    }
    (void)nq_pad;
#undef BMX_STAGE_LOAD
#undef BMX_STAGE_STORE
}

// ---------------------------------------------------------------------------------------------------
// 3. refine: exact FP64 re-rank of the candidates + certification
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double exact_d2(const double* __restrict__ a, const double* __restrict__ b, int d) {
    double s = 0.0;
    for (int c = 0; c < d; ++c) {
        const double t = a[c] - b[c];
        s += t * t;  // compiled with -ffp-contract=off: the reference's left-to-right sum, bit for bit
    }
    return s;
}

__device__ __forceinline__ bool key_less(double da, int ia, double db, int ib) {
    return da < db || (da == db && ia < ib);
}

constexpr int REFINE_MAXM = 64 * 5;  // MAX_CHUNKS * 40

__global__ __launch_bounds__(256) void knn_refine(const double* __restrict__ X, const int32_t* __restrict__ ref_rows,
                                                  const double* __restrict__ Q, const int32_t* __restrict__ q_rows,
                                                  int nq, int d, int k, int KS, int nchunks, int KP,
                                                  const int32_t* __restrict__ cand, const float* __restrict__ tau,
                                                  const double* __restrict__ qn2,
                                                  const unsigned long long* __restrict__ max_rn2_bits,
                                                  int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                  int32_t* __restrict__ flagged) {
    __shared__ double sd[4][REFINE_MAXM];
    __shared__ int si[4][REFINE_MAXM];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;
    const int M = nchunks * KS;
    const double* qv = Q + (int64_t)(q_rows ? q_rows[q] : q) * d;
    for (int m = lane; m < M; m += 64) {
        const int id = cand[(int64_t)q * M + m];
        double d2 = __builtin_inf();
        if (id >= 0) d2 = exact_d2(qv, X + (int64_t)(ref_rows ? ref_rows[id] : id) * d, d);
        sd[w][m] = d2;
        si[w][m] = id >= 0 ? id : 0x7FFFFFFF;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double kth = 0.0;
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
    float tmin = __builtin_inff();
    for (int c = lane; c < nchunks; c += 64) tmin = fminf(tmin, tau[(int64_t)q * nchunks + c]);
    for (int o = 32; o > 0; o >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, o));
        kth = fmax(kth, __shfl_xor(kth, o));
    }
    if (lane == 0) {
        const double qn = sqrt(qn2[q]);
        const double rm = sqrt(__longlong_as_double((long long)*max_rn2_bits));
        const double u = 5.9604644775390625e-8;  // 2^-24
        const double eps = 1.5 * u * (2.0 * (qn + rm) * (qn + rm) + (KP + 1.0) * (rm * rm + 2.0 * qn * rm));
        const bool proven = kth < (double)tmin + qn2[q] - eps;  // tmin = +inf when nothing was ever rejected
        if (!proven) {
            const int pos = atomicAdd(&flagged[0], 1);
            flagged[1 + pos] = q;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 4. exact FP64 scan for flagged queries (or every query when the MFMA path does not apply)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_exact(const double* __restrict__ X, const int32_t* __restrict__ ref_rows,
                                                 int nr, const double* __restrict__ Q,
                                                 const int32_t* __restrict__ q_rows, int nq, int d, int k,
                                                 const int32_t* __restrict__ flagged, int all_queries,
                                                 double* __restrict__ drow, int32_t* __restrict__ idx_out,
                                                 double* __restrict__ dist_out) {
    __shared__ double rd[256];
    __shared__ int ri[256];
    const int tid = threadIdx.x;
    const int count = all_queries ? nq : flagged[0];
    double* row = drow + (int64_t)blockIdx.x * nr;
    for (int f = blockIdx.x; f < count; f += gridDim.x) {
        const int q = all_queries ? f : flagged[1 + f];
        const double* qv = Q + (int64_t)(q_rows ? q_rows[q] : q) * d;
        for (int r = tid; r < nr; r += 256) row[r] = exact_d2(qv, X + (int64_t)(ref_rows ? ref_rows[r] : r) * d, d);
        __syncthreads();
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
}

__global__ void accumulate_flagged(const int32_t* __restrict__ flagged, unsigned long long* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total += (unsigned long long)flagged[0];
}

template <int KP, int KS>
void launch_topk(hipStream_t stream, const float* pq, const float* pr, int nq_pad, int nr_pad, int chunk_len,
                 int nchunks, int32_t* cand, float* tau) {
    constexpr int CAP = KS + SLACK;
    const size_t lds = (size_t)2 * RT * KP * 4 + (size_t)QB * CAP * 8 + QB * 4 + QB * 4;
    static bool attr_set = false;
    if (!attr_set) {
        BMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_topk_mfma<KP, KS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(nq_pad / QB, nchunks);
    hipLaunchKernelGGL((knn_topk_mfma<KP, KS>), grid, dim3(THREADS), lds, stream, pq, pr, nq_pad, nr_pad, chunk_len,
                       nchunks, cand, tau);
    BMX_LAUNCH_CHECK();
}

template <int KS>
bool dispatch_kp(int KP, hipStream_t s, const float* pq, const float* pr, int nq_pad, int nr_pad, int cl, int nc,
                 int32_t* cand, float* tau) {
    switch (KP) {
#define BMX_KP_CASE(V)                                                  \
    case V:                                                             \
        launch_topk<V, KS>(s, pq, pr, nq_pad, nr_pad, cl, nc, cand, tau); \
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

    ws.topk_launched = false;
    int32_t* flagged = ws.flagged.reserve((size_t)nq + 1);
    BMX_HIP(hipMemsetAsync(flagged, 0, sizeof(int32_t), stream));

    if (use_mfma) {
        const int nq_pad = (int)round_up(nq, QB);
        // reference chunks: enough workgroups to fill 256 CUs x 2 several times over, chunk length a tile multiple
        int nchunks = 1;
        const int nqb = nq_pad / QB;
        while (nchunks < MAX_CHUNKS && (int64_t)nqb * nchunks < 1024 && (int64_t)nr / (nchunks * 2) >= 4096) nchunks *= 2;
        const int chunk_len = (int)round_up(cdiv(nr, nchunks), RT);
        nchunks = cdiv(nr, chunk_len);
        const int nr_pad = chunk_len * nchunks;

        float* pq = ws.pq.reserve((size_t)nq_pad * KP);
        float* pr = ws.pr.reserve((size_t)(nr_pad + RT) * KP);  // + one tile: the staging loads over-read
        double* qn2 = ws.qn2.reserve(nq_pad);
        double* rn2 = ws.rn2.reserve(nr_pad);
        double* mean = ws.mean.reserve((size_t)d + 2);
        unsigned long long* maxbits = reinterpret_cast<unsigned long long*>(mean + d);
        int32_t* cand = ws.cand.reserve((size_t)nq_pad * nchunks * KS);
        float* tau = ws.tau.reserve((size_t)nq_pad * nchunks);

        // reference mean (any centre is valid; the mean keeps the f32 error bound tight)
        const int rpb = 1024;
        const int nb = cdiv(nr, rpb);
        double* red = ws.red.reserve((size_t)nb * d);
        hipLaunchKernelGGL(colsum_partial, dim3(nb), dim3(256), 0, stream, X, ref_rows, nr, d, rpb, red);
        BMX_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final, dim3(cdiv(d, 64)), dim3(64), 0, stream, red, nb, d, 1.0 / nr, mean);
        BMX_LAUNCH_CHECK();
        BMX_HIP(hipMemsetAsync(maxbits, 0, sizeof(unsigned long long), stream));

        hipLaunchKernelGGL(knn_prep, dim3(cdiv(nr_pad, 256)), dim3(256), 0, stream, X, ref_rows, nr, nr_pad, d, KP, mean,
                           0, pr, rn2, maxbits);
        BMX_LAUNCH_CHECK();
        hipLaunchKernelGGL(knn_prep, dim3(cdiv(nq_pad, 256)), dim3(256), 0, stream, Qs, qrs, nq, nq_pad, d, KP, mean, 1,
                           pq, qn2, maxbits);
        BMX_LAUNCH_CHECK();

        if (ws.ev_begin) BMX_HIP(hipEventRecord(ws.ev_begin, stream));
        bool ok = KS == 24 ? dispatch_kp<24>(KP, stream, pq, pr, nq_pad, nr_pad, chunk_len, nchunks, cand, tau)
                           : dispatch_kp<40>(KP, stream, pq, pr, nq_pad, nr_pad, chunk_len, nchunks, cand, tau);
        if (!ok) throw Error(BMX_ERR_ARG, "kNN: unsupported padded dimension");
        if (ws.ev_end) BMX_HIP(hipEventRecord(ws.ev_end, stream));
        ws.topk_launched = true;

        hipLaunchKernelGGL(knn_refine, dim3(cdiv(nq, 4)), dim3(256), 0, stream, X, ref_rows, Qs, qrs, nq, d, k, KS,
                           nchunks, KP, cand, tau, qn2, maxbits, io, dout, flagged);
        BMX_LAUNCH_CHECK();
    }

    // exact path: flagged queries, or everything when the MFMA path does not apply
    {
        size_t budget = (size_t)512 << 20;
        int blocks = (int)std::min<size_t>(256, std::max<size_t>(8, budget / ((size_t)nr * 8)));
        blocks = std::min(blocks, std::max(1, nq));
        double* drow = ws.drow.reserve((size_t)blocks * nr);
        hipLaunchKernelGGL(knn_exact, dim3(blocks), dim3(256), 0, stream, X, ref_rows, nr, Qs, qrs, nq, d, k, flagged,
                           use_mfma ? 0 : 1, drow, io, dout);
        BMX_LAUNCH_CHECK();
    }
    if (ws.flag_total && use_mfma) {
        hipLaunchKernelGGL(accumulate_flagged, dim3(1), dim3(64), 0, stream, flagged, ws.flag_total);
        BMX_LAUNCH_CHECK();
    }
}

}  // namespace bmx
