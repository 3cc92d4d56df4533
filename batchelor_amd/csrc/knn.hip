// Exact k-nearest-neighbour search on MI355X (gfx950) -- the >95 % hot spot of fastMNN / reducedMNN.
//
// Replaces the two BiocNeighbors::queryKNN calls inside findMutualNN (R/MNN_tree.R:129) and the one inside
// .tricube_weighted_correction (R/fastMNN.R:605).  Contract: exact Euclidean kNN, ascending distance; ties broken by
// lowest index (upstream leaves ties unpinned).
//
// A search goes through tiers; each tier only sees the queries the one before could not certify:
//   1. knn_topk_f16 (knn_f16.hip): single fp16 product per coordinate, K = d + 3 columns, KS = 32 (k <= 20) or 48
//      (k <= 36) candidates per query and reference range;
//   2. knn_topk_bf16 (knn_bf16.hip): three bf16 products per coordinate (f32-grade), KS = 24 / 40;
//   3. knn_exact_filter / knn_exact_pick: one FP64 sweep of the references against the handful of queries left, bounded
//      by each query's k-th candidate distance;
//   4. knn_exact_dist / knn_exact_select: full FP64 scan (massive exact ties, k > 36, tiny inputs, d > 127).
// After tiers 1 and 2, knn_refine computes the FP64 distances of the candidates in the reference's summation order
// (left to right over the dimensions, no FMA contraction), ranks them by (distance, index), and checks rigorously
// that no reference the candidate pass rejected can enter the top k given the pass's error bound; a query for which
// that cannot be shown is flagged for the next tier.
// Result: indices are exactly those of an FP64 brute-force search with (distance, index) ordering.
#include "bmx_common.hpp"
#include "knn_select.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace bmx {

int f16_pick_ns(int d, int KS);
void f16_prep_all(hipStream_t stream, const double* X, const int32_t* rrows, int nr, int nr_pad, const double* Q,
                  const int32_t* qrows, int nq, int nq_pad, int d, int NS, const double* mean, uint16_t* Pr, uint16_t* Pq,
                  double* rn2, double* qn2, unsigned long long* maxbits, unsigned long long* slots, int32_t* flagged0,
                  float* margin, const sel::PassEps& pe, const float* seed_d2, uint32_t* tau_seed, uint32_t* tau_init);
bool f16_launch(hipStream_t stream, KnnWorkspace& ws, int NS, int KS, const Bf16Launch& L);
void f16_sample_merge(hipStream_t stream, const float* lists, int nranges, int KS, int nq, int k, const float* margin,
                      uint32_t* tau_g);

namespace {

using namespace sel;

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
// FP64 helpers shared by the refine and exact kernels
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double exact_d2(const double* __restrict__ a, const double* __restrict__ b, int d) {
    double s = 0.0;
    int c = 0;
    if ((d & 1) == 0) {  // rows of an even number of doubles are 16-byte aligned: half as many load instructions
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* a2 = reinterpret_cast<const d2*>(a);
        const d2* b2 = reinterpret_cast<const d2*>(b);
        // eight 16-byte pieces of the (randomly placed) row b in flight at a time; the sum itself stays strictly left
        // to right (compiled with -ffp-contract=off: the reference's sum, bit for bit)
        for (; c + 16 <= d; c += 16) {
            d2 y[8], x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = b2[(c >> 1) + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = a2[(c >> 1) + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double t0 = x[i][0] - y[i][0];
                s += t0 * t0;
                const double t1 = x[i][1] - y[i][1];
                s += t1 * t1;
            }
        }
        for (; c < d; c += 2) {
            const d2 x = a2[c >> 1], y = b2[c >> 1];
            const double t0 = x[0] - y[0];
            s += t0 * t0;
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

// ---------------------------------------------------------------------------------------------------
// refine: exact FP64 re-rank of the candidates + certification.  One wave per query.
//   1. the candidates of all reference ranges are gathered into one dense list;
//   2. they are ranked by their approximate values: only the KS best go on (the first one cut bounds all the others
//      from below and enters the certificate), and of those only the ones within twice the error bound of the k-th
//      best -- a candidate further out is provably farther than k others, so its exact distance is never needed;
//   3. FP64 distances (the CPU's summation order, bit for bit), exact (distance, index) ranking;
//   4. certificate: every reference the candidate pass rejected had an approximate value >= tau, i.e. an exact
//      squared distance >= tau + |q~|^2 - eps, so the top k is proven when the k-th exact distance is below that.
// Values of a scaled pass (fp16 tier: coordinates times the power of two s) are brought back with 1 / s^2.
// ---------------------------------------------------------------------------------------------------
constexpr int REFINE_MAXM = MAX_CHUNKS * 48;

// seeded search: tau_g = min(what is there, the seed's threshold) -- the form for passes that do not take the fused prep
__global__ void seed_tau_kernel(const float* __restrict__ seed_d2, const double* __restrict__ qn2,
                                const unsigned long long* __restrict__ max_rn2_bits, int nq, int nq_pad, PassEps pe,
                                uint32_t* __restrict__ tau_g) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq_pad) return;
    uint32_t o = f32_orderable(-__builtin_inff());  // padded queries: nothing passes
    if (q < nq) o = pass_seed_tau((double)seed_d2[q], qn2[q], __longlong_as_double((long long)*max_rn2_bits), pe);
    tau_g[q] = q < nq ? min(tau_g[q], o) : o;  // the image is order-preserving: min of images = image of the min
}

template <int REFINE_NC>  // pieces of 8 doubles (one 16-byte load per lane of a quad) a row may have; 0: lane-per-row gather
__global__ __launch_bounds__(256) void knn_refine(const double* __restrict__ X, const int32_t* __restrict__ ref_rows,
                                                  const double* __restrict__ Q, const int32_t* __restrict__ q_rows,
                                                  int nq, int d, int k, int KS, int nchunks, double eps_k, double eps_qr,
                                                  double eps_split, double eps_den, int scaled,
                                                  const int32_t* __restrict__ cand, const float* __restrict__ cand_v,
                                                  const float* __restrict__ tau, const double* __restrict__ qn2,
                                                  const unsigned long long* __restrict__ max_rn2_bits,
                                                  const float* __restrict__ seed_d2, int32_t* __restrict__ idx_out,
                                                  double* __restrict__ dist_out, int32_t* __restrict__ flagged,
                                                  double* __restrict__ flag_bound,
                                                  unsigned long long* __restrict__ zero_slots, int q0,
                                                  double* __restrict__ kth_out, int sq = 0) {
    __shared__ __attribute__((aligned(16))) double sd[4][REFINE_MAXM];
    __shared__ int si[4][REFINE_MAXM];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // (the prep kernels' 64 slot maxima have been folded by now: left zeroed for the next search's norm pass)
    if (zero_slots && blockIdx.x == 0 && threadIdx.x < 64) zero_slots[(size_t)threadIdx.x * 16] = 0ull;
    const int q = q0 + blockIdx.x * 4 + w;  // (queries [q0, nq): the ones in front go through knn_refine_half)
    if (q >= nq) return;
    const int Mall = nchunks * KS;
    const double* qv = Q + (int64_t)(q_rows ? q_rows[q] : q) * d;
    // error bound of the candidate pass for this query (unscaled units)
    const double max_rn2 = __longlong_as_double((long long)*max_rn2_bits);
    const double s = scaled ? pass_scale(max_rn2) : 1.0;
    const double s2inv = 1.0 / (s * s);
    const double eps = pass_eps(sqrt(qn2[q]), sqrt(max_rn2), s, eps_k, eps_qr, eps_split, eps_den);
    // 1. dense list of the valid candidates (ballot prefix)
    int M = 0;
    // this wave's sd row doubles as the list of rank keys until the exact distances go there: (order-preserving image
    // of the approximate value) << 32 | index -- unique, and ordered like (value, index)
    unsigned long long* sk = reinterpret_cast<unsigned long long*>(&sd[w][0]);
    for (int m0 = 0; m0 < Mall; m0 += 64) {
        const int m = m0 + lane;
        const int id = m < Mall ? cand[(int64_t)q * Mall + m] : -1;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(id >= 0);
        const int pos = M + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (id >= 0) {
            si[w][pos] = id;
            if (cand_v) sk[pos] = ((unsigned long long)f32_orderable(cand_v[(int64_t)q * Mall + m]) << 32) | (uint32_t)id;
        }
        M += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // 2. rank by approximate value
    float tmerge = __builtin_inff();
    if (cand_v && M > k) {
        constexpr int NU = (REFINE_MAXM + 63) / 64;
        int rk[NU], ids[NU];
        float vs[NU];
        float vk = __builtin_inff();  // k-th smallest approximate value
#pragma unroll
        for (int uu = 0; uu < NU; ++uu) {
            const int m = lane + 64 * uu;
            rk[uu] = 0x7FFFFFFF;
            ids[uu] = 0;
            vs[uu] = 0.f;
            if (m < M) {
                const unsigned long long km = sk[m];
                const float vm = orderable_f32((uint32_t)(km >> 32));
                int rank = 0;
                int f = 0;
                for (; f + 2 <= M; f += 2) {  // (16-byte reads: the row is 16-byte aligned, f even)
                    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                    const u64x2 kf = *reinterpret_cast<const u64x2*>(sk + f);
                    rank += (kf[0] < km ? 1 : 0) + (kf[1] < km ? 1 : 0);
                }
                if (f < M) rank += sk[f] < km ? 1 : 0;
                rk[uu] = rank;
                ids[uu] = (int)(uint32_t)km;
                vs[uu] = vm;
                if (rank == k - 1) vk = vm;
                if (rank == KS) tmerge = vm;  // the first one cut by rank bounds all the others cut from below
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            tmerge = fminf(tmerge, __shfl_xor(tmerge, o));
            vk = fminf(vk, __shfl_xor(vk, o));
        }
        // a candidate whose approximate value exceeds the k-th best by more than twice the bound is farther (exactly)
        // than each of the k best: it cannot be among the k nearest, whatever else happens
        const float cut = (float)((double)vk + 2.0 * eps * (s * s) * 1.0000002 + 1e-30);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int Mn = 0;
#pragma unroll
        for (int uu = 0; uu < NU; ++uu) {
            if (uu * 64 < M) {
                const bool keep = rk[uu] < KS && vs[uu] <= cut;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
                if (keep) si[w][Mn + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = ids[uu];
                Mn += __builtin_popcountll(mask);
            }
        }
        M = Mn;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // 3. exact distances and ranks
    if constexpr (REFINE_NC > 0) {
        // Four lanes per candidate row: their 16-byte loads are 64 contiguous bytes of the (randomly placed) row, so
        // every cache line of it is asked for twice instead of eight times as with a lane per row.  The sum stays the
        // reference's: one running value per row, handed from lane to lane of the quad (DPP) every two elements,
        // strictly left to right (compiled with -ffp-contract=off: bit for bit the CPU's sum).
        typedef double d2 __attribute__((ext_vector_type(2)));
        const int g = lane >> 2, p = lane & 3;
        const int np = d >> 1;  // 16-byte pieces per row
        const d2* q2 = reinterpret_cast<const d2*>(qv);
        d2 x[REFINE_NC > 0 ? REFINE_NC : 1];
#pragma unroll
        for (int c = 0; c < REFINE_NC; ++c) {
            x[c] = d2{0.0, 0.0};
            if (4 * c + p < np) x[c] = q2[4 * c + p];
        }
        const int p_last = (np - 1) & 3;
        for (int r0 = 0; r0 < M; r0 += 16) {
            const int m = r0 + g;
            const int id = si[w][m < M ? m : 0];
            const d2* row2 = reinterpret_cast<const d2*>(X + (int64_t)(ref_rows ? ref_rows[id] : id) * d);
            d2 y[REFINE_NC > 0 ? REFINE_NC : 1];
#pragma unroll
            for (int c = 0; c < REFINE_NC; ++c) {
                y[c] = d2{0.0, 0.0};
                if (4 * c + p < np) y[c] = row2[4 * c + p];
            }
            // the squares: every lane its own two per piece, all at once; only the additions have an order
#pragma unroll
            for (int c = 0; c < REFINE_NC; ++c) {
                const double t0 = x[c][0] - y[c][0], t1 = x[c][1] - y[c][1];
                y[c][0] = t0 * t0;
                y[c][1] = t1 * t1;
            }
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < REFINE_NC; ++c) {
                if (4 * c < np) {  // (wave-uniform)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) {
                        // the running sum as the previous lane of the quad has it (lane 0: lane 3's, from the piece before)
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)bits, 0x93, 0xF, 0xF, false);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(bits >> 32), 0x93, 0xF, 0xF, false);
                        const double prev = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                        // (0 + x is x exactly: the first element needs no case of its own -- prev is lane 3's untouched 0)
                        double run = prev + y[c][0];
                        run += y[c][1];
                        if (p == ph && 4 * c + ph < np) acc = run;
                    }
                }
            }
            if (p == p_last && m < M) sd[w][m] = acc;
        }
    } else {
        for (int m = lane; m < M; m += 64) {
            const int id = si[w][m];
            sd[w][m] = exact_d2(qv, X + (int64_t)(ref_rows ? ref_rows[id] : id) * d, d);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // fewer than k candidates: never certified -- unless the search was seeded, where the row is complete as soon as
    // nothing within the seed distance can have been rejected: the seed then plays the k-th distance's part
    const double seed = seed_d2 ? (double)seed_d2[q] : __builtin_inf();
    double kth = M >= k ? 0.0 : seed;
    // (an unseeded row short of candidates is never certified and is rewritten by the exact sweep -- but an optimistic run
    // reads it before the host has seen the flag: its unused places hold a valid position, not whatever was there)
    for (int m = M + lane; m < k; m += 64) idx_out[(int64_t)q * k + m] = seed_d2 ? -1 : 0;
    for (int m = lane; m < M; m += 64) {
        const double dm = sd[w][m];
        const int im = si[w][m];
        int rank = 0;
        for (int f = 0; f < M; ++f) rank += key_less(sd[w][f], si[w][f], dm, im) ? 1 : 0;
        if (rank < k) {
            idx_out[(int64_t)q * k + rank] = im;
            if (dist_out) dist_out[(int64_t)q * k + rank] = sq ? dm : sqrt(dm);  // (sq: squared, for the partitioned search's merge)
        }
        if (rank == k - 1) kth = dm;
    }
    // 4. certificate
    float tmin = tmerge;
    for (int c = lane; c < nchunks; c += 64) tmin = fminf(tmin, tau[(int64_t)q * nchunks + c]);
    for (int o = 32; o > 0; o >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, o));
        kth = fmax(kth, __shfl_xor(kth, o));
    }
    const double kth_full = kth;  // the row's k-th distance (squared) where it has k entries
    // a seeded row only has to be right up to the seed distance: entries beyond it are of no use to the caller
    kth = fmin(kth, seed);
    if (lane == 0) {
        const bool proven = kth < (double)tmin * s2inv + qn2[q] - eps;  // tmin = +inf when nothing was ever rejected
        if (!proven) {
            const int pos = atomicAdd(&flagged[0], 1);
            flagged[1 + pos] = q;
            flag_bound[pos] = kth;  // the true k-th neighbour is no farther than the k-th candidate
        }
        // the largest distance of a certified FULL row (what the intersection's probe rejects against without reading the
        // row); +inf where the row is short or may still be rewritten: the probe then reads the row
        if (kth_out) kth_out[q] = (proven && M >= k) ? sqrt(kth_full) : __builtin_inf();
    }
}

// ---------------------------------------------------------------------------------------------------
// The same for queries with ONE list of at most 32 candidates (the query blocks that swept the whole reference as a single
// range with KS = 32: two thirds to nine tenths of a search's queries): HALF a wave per query.  A query's candidates fill
// at most 32 lanes, so a whole wave spent most of the gather, both ranking loops and the certificate on idle lanes; with
// two queries per wave those parts cost half as much per query (the FP64 distance chain -- a quad of lanes per candidate
// row, 8 rows per half and trip -- costs the same).  Same arithmetic, same order, same outputs as knn_refine.
// Layout: query q's list is cand[q * stride .. + 32) (stride = nchunks * KS: the other ranges' columns of such a query hold
// nothing), its threshold tau[q * nchunks].
// ---------------------------------------------------------------------------------------------------
template <int REFINE_NC>
__global__ __launch_bounds__(256) void knn_refine_half(const double* __restrict__ X, const int32_t* __restrict__ ref_rows,
                                                       const double* __restrict__ Q, const int32_t* __restrict__ q_rows,
                                                       int nq, int d, int k, int nchunks, double eps_k, double eps_qr,
                                                       double eps_split, double eps_den, int scaled,
                                                       const int32_t* __restrict__ cand, const float* __restrict__ cand_v,
                                                       const float* __restrict__ tau, const double* __restrict__ qn2,
                                                       const unsigned long long* __restrict__ max_rn2_bits,
                                                       const float* __restrict__ seed_d2, int32_t* __restrict__ idx_out,
                                                       double* __restrict__ dist_out, int32_t* __restrict__ flagged,
                                                       double* __restrict__ flag_bound,
                                                       unsigned long long* __restrict__ zero_slots,
                                                       double* __restrict__ kth_out, int sq = 0) {
    constexpr int KS = 32;
    __shared__ __attribute__((aligned(16))) double sd[8][KS];
    __shared__ int si[8][KS];
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int w = (threadIdx.x >> 6) * 2 + half;  // the half-wave's row of the LDS arrays
    if (zero_slots && blockIdx.x == 0 && threadIdx.x < 64) zero_slots[(size_t)threadIdx.x * 16] = 0ull;
    const int qraw = blockIdx.x * 8 + w;
    const bool live = qraw < nq;
    const int q = live ? qraw : nq - 1;  // (an idle half follows the last query and writes nothing: shuffles stay convergent)
    const int64_t stride = (int64_t)nchunks * KS;
    const double* qv = Q + (int64_t)(q_rows ? q_rows[q] : q) * d;
    const double max_rn2 = __longlong_as_double((long long)*max_rn2_bits);
    const double s = scaled ? pass_scale(max_rn2) : 1.0;
    const double s2inv = 1.0 / (s * s);
    const double eps = pass_eps(sqrt(qn2[q]), sqrt(max_rn2), s, eps_k, eps_qr, eps_split, eps_den);
    auto half_ballot = [&](bool p) { return (uint32_t)(__builtin_amdgcn_ballot_w64(p) >> (half << 5)); };
    // 1. dense list of the valid candidates
    unsigned long long* sk = reinterpret_cast<unsigned long long*>(&sd[w][0]);
    int M;
    {
        const int id = cand[(int64_t)q * stride + hl];
        const uint32_t mask = half_ballot(id >= 0);
        const int pos = __builtin_popcount(mask & ((1u << hl) - 1u));
        if (id >= 0) {
            si[w][pos] = id;
            if (cand_v) sk[pos] = ((unsigned long long)f32_orderable(cand_v[(int64_t)q * stride + hl]) << 32) | (uint32_t)id;
        }
        M = __builtin_popcount(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // 2. rank by approximate value; only the candidates within twice the error bound of the k-th best go on
    const int Mo = max(M, __shfl_xor(M, 32));  // (loops run to the longer of the wave's two lists)
    if (cand_v) {
        const bool rankit = M > k;
        int rank = 0x7FFFFFFF, id = 0;
        float vm = 0.f, vk = __builtin_inff();
        if (hl < M) {
            const unsigned long long km = sk[hl];
            vm = orderable_f32((uint32_t)(km >> 32));
            id = (int)(uint32_t)km;
            rank = 0;
        }
        for (int f = 0; f + 2 <= Mo; f += 2) {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            const u64x2 kf = *reinterpret_cast<const u64x2*>(sk + f);
            if (hl < M) {
                const unsigned long long km = ((unsigned long long)f32_orderable(vm) << 32) | (uint32_t)id;
                rank += (f < M && kf[0] < km ? 1 : 0) + (f + 1 < M && kf[1] < km ? 1 : 0);
            }
        }
        if ((Mo & 1) && hl < M && Mo - 1 < M) {
            const unsigned long long km = ((unsigned long long)f32_orderable(vm) << 32) | (uint32_t)id;
            rank += sk[Mo - 1] < km ? 1 : 0;
        }
        if (hl < M && rank == k - 1) vk = vm;
        for (int o = 16; o > 0; o >>= 1) vk = fminf(vk, __shfl_xor(vk, o));
        const float cut = (float)((double)vk + 2.0 * eps * (s * s) * 1.0000002 + 1e-30);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (rankit) {
            const bool keep = hl < M && vm <= cut;
            const uint32_t mask = half_ballot(keep);
            if (keep) si[w][__builtin_popcount(mask & ((1u << hl) - 1u))] = id;
            M = __builtin_popcount(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // 3. exact distances: a quad of lanes per candidate row, 8 rows per half and trip (see knn_refine)
    const int M3 = max(M, __shfl_xor(M, 32));
    if constexpr (REFINE_NC > 0) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const int g = hl >> 2, p = hl & 3;
        const int np = d >> 1;
        const d2* q2 = reinterpret_cast<const d2*>(qv);
        d2 x[REFINE_NC];
#pragma unroll
        for (int c = 0; c < REFINE_NC; ++c) {
            x[c] = d2{0.0, 0.0};
            if (4 * c + p < np) x[c] = q2[4 * c + p];
        }
        const int p_last = (np - 1) & 3;
        for (int r0 = 0; r0 < M3; r0 += 8) {
            const int m = r0 + g;
            const int id = M > 0 ? si[w][m < M ? m : 0] : 0;  // (the other half's list may be the longer one: a valid row)
            const d2* row2 = reinterpret_cast<const d2*>(X + (int64_t)(ref_rows ? ref_rows[id] : id) * d);
            d2 y[REFINE_NC];
#pragma unroll
            for (int c = 0; c < REFINE_NC; ++c) {
                y[c] = d2{0.0, 0.0};
                if (4 * c + p < np) y[c] = row2[4 * c + p];
            }
#pragma unroll
            for (int c = 0; c < REFINE_NC; ++c) {
                const double t0 = x[c][0] - y[c][0], t1 = x[c][1] - y[c][1];
                y[c][0] = t0 * t0;
                y[c][1] = t1 * t1;
            }
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < REFINE_NC; ++c) {
                if (4 * c < np) {  // (wave-uniform)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) {
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)bits, 0x93, 0xF, 0xF, false);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(bits >> 32), 0x93, 0xF, 0xF, false);
                        const double prev = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                        double run = prev + y[c][0];
                        run += y[c][1];
                        if (p == ph && 4 * c + ph < np) acc = run;
                    }
                }
            }
            if (p == p_last && m < M) sd[w][m] = acc;
        }
    } else {
        if (hl < M) {
            const int id = si[w][hl];
            sd[w][hl] = exact_d2(qv, X + (int64_t)(ref_rows ? ref_rows[id] : id) * d, d);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const double seed = seed_d2 ? (double)seed_d2[q] : __builtin_inf();
    double kth = M >= k ? 0.0 : seed;
    if (live)
        for (int m = M + hl; m < k; m += 32) idx_out[(int64_t)q * k + m] = seed_d2 ? -1 : 0;
    {
        const double dm = hl < M ? sd[w][hl] : 0.0;
        const int im = hl < M ? si[w][hl] : 0;
        int rank = 0;
        for (int f = 0; f < M3; ++f) {
            if (f < M) rank += key_less(sd[w][f], si[w][f], dm, im) ? 1 : 0;
        }
        if (hl < M && rank < k && live) {
            idx_out[(int64_t)q * k + rank] = im;
            if (dist_out) dist_out[(int64_t)q * k + rank] = sq ? dm : sqrt(dm);  // (sq: squared, for the partitioned search's merge)
        }
        if (hl < M && rank == k - 1) kth = dm;
    }
    // 4. certificate
    for (int o = 16; o > 0; o >>= 1) kth = fmax(kth, __shfl_xor(kth, o));
    const double kth_full = kth;
    kth = fmin(kth, seed);
    if (hl == 0 && live) {
        const bool proven = kth < (double)tau[(int64_t)q * nchunks] * s2inv + qn2[q] - eps;
        if (!proven) {
            const int pos = atomicAdd(&flagged[0], 1);
            flagged[1 + pos] = q;
            flag_bound[pos] = kth;
        }
        if (kth_out) kth_out[q] = (proven && M >= k) ? sqrt(kth_full) : __builtin_inf();
    }
}

// ---------------------------------------------------------------------------------------------------
// 4. exact FP64 scan for flagged queries (or every query when the MFMA path does not apply), in batches:
//    knn_exact_dist   -- grid (tiles of 64 references, tiles of 64 queries of the batch): all exact squared distances, spread
//                        over the whole chip even when only a handful of queries are flagged;
//    knn_exact_select -- one workgroup per query: k rounds of block-wide (distance, index) minimum.
// ---------------------------------------------------------------------------------------------------
// A tile of 64 references x 64 queries of the batch per workgroup, the rows staged through the LDS 32 columns at a time; a thread
// holds 4 x 4 pairs.  Every pair's sum runs over the columns left to right, one subtraction, one product, one addition each
// (compiled with -ffp-contract=off): exact_d2's value bit for bit.  (Until late in round 6 a thread took ONE pair and read both
// its rows itself: 1.2 KB of L2 traffic a pair at 150 columns, 555 ms for 20 000 queries against 100 000 cells.)
constexpr int XDT = 64;   // rows of a tile, both ways
constexpr int XDC = 32;   // columns staged at a time
__global__ __launch_bounds__(256) void knn_exact_dist(const double* __restrict__ X, const int32_t* __restrict__ ref_rows,
                                                      int nr, const double* __restrict__ Q,
                                                      const int32_t* __restrict__ q_rows, int d,
                                                      const int32_t* __restrict__ flagged, int f0, int nb,
                                                      double* __restrict__ drow, int dev_cap = 0) {
    // (dev_cap > 0: the number of listed queries is on the device -- flagged[0], at most dev_cap of them are taken --, the grid
    // is made for dev_cap and the tiles beyond the count end here)
    if (dev_cap > 0) nb = min(flagged[0], dev_cap);
    const int q0 = blockIdx.y * XDT, r0 = blockIdx.x * XDT;
    if (q0 >= nb) return;
    __shared__ double Qs_[XDT][XDC + 1], Xs_[XDT][XDC + 1];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int lr = tid >> 2, lc = (tid & 3) * 8;  // this thread stages 8 columns of row lr of either tile
    const double* qp = nullptr;
    const double* xp = nullptr;
    if (q0 + lr < nb) {
        const int f = f0 + q0 + lr;
        const int q = flagged ? flagged[1 + f] : f;
        qp = Q + (int64_t)(q_rows ? q_rows[q] : q) * d;
    }
    if (r0 + lr < nr) xp = X + (int64_t)(ref_rows ? ref_rows[r0 + lr] : r0 + lr) * d;
    double s[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) s[i][j2] = 0.0;
    for (int c0 = 0; c0 < d; c0 += XDC) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + lc + e;
            Qs_[lr][lc + e] = (qp && c < d) ? qp[c] : 0.0;
            Xs_[lr][lc + e] = (xp && c < d) ? xp[c] : 0.0;
        }
        __syncthreads();
        const int cmax = d - c0 < XDC ? d - c0 : XDC;
        for (int c = 0; c < cmax; ++c) {
            double qv[4], xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[i] = Qs_[ty * 4 + i][c];
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) xv[j2] = Xs_[tx * 4 + j2][c];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const double t = qv[i] - xv[j2];
                    s[i][j2] += t * t;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qi = q0 + ty * 4 + i;
        if (qi >= nb) continue;
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            const int r = r0 + tx * 4 + j2;
            if (r < nr) drow[(int64_t)qi * nr + r] = s[i][j2];
        }
    }
}

// k <= 256: TWO sweeps of the row instead of k.  The k-th smallest of the 256 threads' own minima bounds the k-th smallest of the
// row from above (k different entries lie at or below it); the entries at or below that bound -- about k (1 + k / 256) of them --
// are collected in the LDS and sorted by (distance, position).  More than XS2_CAP of them (a row of equal distances): the k
// rounds below.
constexpr int XS2_CAP = 2048;
__device__ __forceinline__ void knn_exact_select_rounds(const double* __restrict__ row, int nr, int k, int q,
                                                        int32_t* __restrict__ idx_out, double* __restrict__ dist_out, int sq,
                                                        double* rd, int* ri);

__global__ __launch_bounds__(256) void knn_exact_select(const double* __restrict__ drow, int nr, int k,
                                                        const int32_t* __restrict__ flagged, int f0,
                                                        int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                        int dev_cap = 0, int sq = 0) {
    __shared__ double sd[XS2_CAP];
    __shared__ int si[XS2_CAP];
    __shared__ double tau_sh;
    __shared__ int cnt_sh;
    if (dev_cap > 0 && (int)blockIdx.x >= min(flagged[0], dev_cap)) return;  // (see knn_exact_dist)
    const int tid = threadIdx.x;
    const int f = f0 + blockIdx.x;
    const int q = flagged ? flagged[1 + f] : f;
    const double* row = drow + (int64_t)blockIdx.x * nr;
    if (k <= 256) {
        double m = __builtin_inf();
        for (int r = tid; r < nr; r += 256) m = fmin(m, row[r]);
        sd[tid] = m;
        if (tid == 0) cnt_sh = 0;
        __syncthreads();
        int rank = 0;
        for (int t = 0; t < 256; ++t) rank += (sd[t] < m || (sd[t] == m && t < tid)) ? 1 : 0;
        if (rank == k - 1) tau_sh = m;  // (k <= nr: at least k threads hold an entry)
        __syncthreads();
        const double tau = tau_sh;
        __syncthreads();  // (sd is reused)
        for (int r = tid; r < nr; r += 256) {
            const double v = row[r];
            if (v <= tau) {
                const int at = atomicAdd(&cnt_sh, 1);
                if (at < XS2_CAP) {
                    sd[at] = v;
                    si[at] = r;
                }
            }
        }
        __syncthreads();
        const int n = cnt_sh;
        if (n <= XS2_CAP) {
            int np2 = 64;
            while (np2 < n) np2 <<= 1;
            for (int i = n + tid; i < np2; i += 256) {
                sd[i] = __builtin_inf();
                si[i] = 0x7FFFFFFF;
            }
            __syncthreads();
            for (int size = 2; size <= np2; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int t = tid; t < (np2 >> 1); t += 256) {
                        const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                        const double a = sd[i], b = sd[j];
                        const int ai = si[i], bi = si[j];
                        if (key_less(b, bi, a, ai) == ((i & size) == 0)) {
                            sd[i] = b;
                            sd[j] = a;
                            si[i] = bi;
                            si[j] = ai;
                        }
                    }
                    __syncthreads();
                }
            for (int j = tid; j < k; j += 256) {
                idx_out[(int64_t)q * k + j] = si[j];
                if (dist_out) dist_out[(int64_t)q * k + j] = sq ? sd[j] : sqrt(sd[j]);
            }
            return;
        }
        __syncthreads();
    }
    knn_exact_select_rounds(row, nr, k, q, idx_out, dist_out, sq, sd, si);
}

__device__ __forceinline__ void knn_exact_select_rounds(const double* __restrict__ row, int nr, int k, int q,
                                                        int32_t* __restrict__ idx_out, double* __restrict__ dist_out, int sq,
                                                        double* rd, int* ri) {
    const int tid = threadIdx.x;
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
            if (dist_out) dist_out[(int64_t)q * k + jdx] = sq ? last_d : sqrt(last_d);
        }
        __syncthreads();
    }
}

// The same selection for k > 64 (prop.k runs; the k rounds above are 69 ms a query at k = 1 000 over 100 000 references,
// however few queries there are -- a query is one workgroup).  Squared distances are >= 0, so they order like their bit
// patterns: a bisection over the patterns finds the k-th smallest value T (<= 63 counting sweeps of the row, which stays in
// L2), a bisection over the positions finds how far into the references equal to T the list reaches (ties rank by position,
// as above), the k chosen entries are compacted into LDS and sorted by (distance, position).
constexpr int XSB_T = 1024;
constexpr int XSB_MAXK = 8192;

__device__ __forceinline__ int xsb_block_sum(int v, int* sh_part) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();  // (sh_part is read by everyone at the end of the previous call)
    if ((threadIdx.x & 63) == 0) sh_part[threadIdx.x >> 6] = v;
    __syncthreads();
    int s = 0;
    for (int w = 0; w < XSB_T / 64; ++w) s += sh_part[w];
    return s;
}

__global__ __launch_bounds__(XSB_T) void knn_exact_select_big(const double* __restrict__ drow, int nr, int k, int np2,
                                                              const int32_t* __restrict__ flagged, int f0,
                                                              int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                              int dev_cap = 0, int sq = 0) {
    extern __shared__ double xsb_d[];       // [np2] distances, then [np2] positions
    if (dev_cap > 0 && (int)blockIdx.x >= min(flagged[0], dev_cap)) return;  // (see knn_exact_dist)
    int* xsb_i = reinterpret_cast<int*>(xsb_d + np2);
    __shared__ int sh_part[XSB_T / 64];
    __shared__ int sh_fill;
    const int tid = threadIdx.x;
    const int f = f0 + blockIdx.x;
    const int q = flagged ? flagged[1 + f] : f;
    const double* row = drow + (int64_t)blockIdx.x * nr;
    // the k-th smallest pattern (NaN patterns lie above +inf's and are never counted; the caller has k <= nr finite rows)
    unsigned long long lo = 0, hi = 0x7FF0000000000000ull;
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int r = tid; r < nr; r += XSB_T) c += (unsigned long long)__double_as_longlong(row[r]) <= mid ? 1 : 0;
        if (xsb_block_sum(c, sh_part) >= k) hi = mid;
        else lo = mid + 1;
    }
    const unsigned long long T = lo;
    int c = 0;
    for (int r = tid; r < nr; r += XSB_T) c += (unsigned long long)__double_as_longlong(row[r]) < T ? 1 : 0;
    const int need = k - xsb_block_sum(c, sh_part);  // >= 1 entries equal to T, the first in position order
    int plo = 0, phi = nr - 1;  // smallest position P with `need` entries equal to T at or before it
    while (plo < phi) {
        const int mid = plo + ((phi - plo) >> 1);
        int e = 0;
        for (int r = tid; r <= mid; r += XSB_T) e += (unsigned long long)__double_as_longlong(row[r]) == T ? 1 : 0;
        if (xsb_block_sum(e, sh_part) >= need) phi = mid;
        else plo = mid + 1;
    }
    const int P = plo;
    if (tid == 0) sh_fill = 0;
    for (int i = tid; i < np2; i += XSB_T) {
        xsb_d[i] = __builtin_inf();
        xsb_i[i] = 0x7FFFFFFF;
    }
    __syncthreads();
    for (int r = tid; r < nr; r += XSB_T) {
        const double v = row[r];
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        if (b < T || (b == T && r <= P)) {
            const int at = atomicAdd(&sh_fill, 1);
            if (at < np2) {
                xsb_d[at] = v;
                xsb_i[at] = r;
            }
        }
    }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (np2 >> 1); t += XSB_T) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const double a = xsb_d[i], b = xsb_d[j];
                const int ai = xsb_i[i], bi = xsb_i[j];
                if (key_less(b, bi, a, ai) == ((i & size) == 0)) {
                    xsb_d[i] = b;
                    xsb_d[j] = a;
                    xsb_i[i] = bi;
                    xsb_i[j] = ai;
                }
            }
            __syncthreads();
        }
    for (int j = tid; j < k; j += XSB_T) {
        idx_out[(int64_t)q * k + j] = xsb_i[j];
        if (dist_out) dist_out[(int64_t)q * k + j] = sq ? xsb_d[j] : sqrt(xsb_d[j]);
    }
}

// ---------------------------------------------------------------------------------------------------
// 4b. fast exact path for a FEW flagged queries: the k-th candidate distance bounds the true k-th neighbour from
//     above, so one pass that stages each reference tile once in LDS, evaluates it against every flagged query in
//     FP64 and keeps the references within that bound (a handful per query) replaces the full per-query rescan.
//     A query whose list overflows (massive exact ties) is handed to the full scan.
// ---------------------------------------------------------------------------------------------------
constexpr int XF_TILE = 64;    // reference rows per staged tile
constexpr int XF_QCH = 16;     // flagged queries staged at a time
constexpr int XF_CAP = 256;    // kept references per flagged query

__global__ __launch_bounds__(256) void knn_exact_filter(const double* __restrict__ X,
                                                        const int32_t* __restrict__ ref_rows, int nr,
                                                        const double* __restrict__ Q,
                                                        const int32_t* __restrict__ q_rows, int d,
                                                        const int32_t* __restrict__ flagged,
                                                        const double* __restrict__ flag_bound, int nflag, int dev_cap,
                                                        int32_t* __restrict__ xcnt, double* __restrict__ xd,
                                                        int32_t* __restrict__ xi) {
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    if (dev_cap > 0) {  // launched without the host knowing the count: nothing flagged (the rule) -> nothing staged
        nflag = flagged[0];
        if (nflag <= 0 || nflag > dev_cap) return;  // (more than the cap: knn_exact_pick raises the run's invalid flag)
    }
    // Tiles of 64 reference rows through the LDS: a wave copies a row per instruction (lanes along the row: 16-byte pieces
    // where the rows allow, one coalesced read of its 8 d bytes), then every thread takes one row of the tile against every
    // flagged query -- exact_d2's sum, left to right.  (Round 3 staged element by element with an integer division per
    // double: 380 us for a sweep of 700 000 rows that moves 280 MB; a thread per row straight from global memory: worse,
    // 64 cache lines per load instruction.)
    double* xs = reinterpret_cast<double*>(smem_x);  // [XF_TILE][d + 1]  (+1: breaks the bank stride)
    const int ld = d + 1;
    double* qs = xs + XF_TILE * ld;                  // [XF_QCH][d + 1] flagged queries
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool wide = (d & 1) == 0;
    const int np = wide ? d >> 1 : d;  // pieces per row
    for (int r0 = blockIdx.x * XF_TILE; r0 < nr; r0 += gridDim.x * XF_TILE) {
        const int rows_here = min(XF_TILE, nr - r0);
        if (wide && np <= 64) {
            // all 16 rows of this wave asked for before the first is stored: their round trips overlap
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 v[XF_TILE / 4];
#pragma unroll
            for (int i = 0; i < XF_TILE / 4; ++i) {
                const int rr = w + 4 * i;
                v[i] = d2{0.0, 0.0};
                if (rr < rows_here && lane < np)
                    v[i] = reinterpret_cast<const d2*>(X + (int64_t)(ref_rows ? ref_rows[r0 + rr] : r0 + rr) * d)[lane];
            }
#pragma unroll
            for (int i = 0; i < XF_TILE / 4; ++i) {
                const int rr = w + 4 * i;
                if (rr < rows_here && lane < np) {
                    xs[rr * ld + 2 * lane] = v[i][0];
                    xs[rr * ld + 2 * lane + 1] = v[i][1];
                }
            }
        } else {
            for (int rr = w; rr < rows_here; rr += 4) {
                const double* src = X + (int64_t)(ref_rows ? ref_rows[r0 + rr] : r0 + rr) * d;
                for (int p = lane; p < np; p += 64) {
                    if (wide) {
                        typedef double d2 __attribute__((ext_vector_type(2)));
                        const d2 v = reinterpret_cast<const d2*>(src)[p];
                        xs[rr * ld + 2 * p] = v[0];
                        xs[rr * ld + 2 * p + 1] = v[1];
                    } else {
                        xs[rr * ld + p] = src[p];
                    }
                }
            }
        }
        __syncthreads();
        const int rr = tid & (XF_TILE - 1);
        const double* xr = xs + rr * ld;
        // the flagged queries, XF_QCH at a time, through the LDS as well: read from global memory inside the sum, every one of
        // the d steps of every thread's chain waited for an L2 round trip (0.8 ms per config-3 step for ~25 queries)
        for (int f0 = 0; f0 < nflag; f0 += XF_QCH) {
            const int nq_here = min(XF_QCH, nflag - f0);
            for (int e = tid; e < nq_here * d; e += 256) {
                const int fq = e / d, c = e - fq * d;
                const int q = flagged[1 + f0 + fq];
                qs[fq * ld + c] = Q[(int64_t)(q_rows ? q_rows[q] : q) * d + c];
            }
            __syncthreads();
            for (int fq = tid / XF_TILE; fq < nq_here && rr < rows_here; fq += 256 / XF_TILE) {
                const double* qv = qs + fq * ld;
                double s = 0.0;
                for (int c = 0; c < d; ++c) {
                    const double t = qv[c] - xr[c];
                    s += t * t;
                }
                const int f = f0 + fq;
                if (s <= flag_bound[f]) {
                    const int pos = atomicAdd(&xcnt[f], 1);
                    if (pos < XF_CAP) {
                        xd[(int64_t)f * XF_CAP + pos] = s;
                        xi[(int64_t)f * XF_CAP + pos] = r0 + rr;
                    }
                }
            }
            __syncthreads();
        }
        __syncthreads();  // the tile is restaged
    }
}

// one wave per flagged query: exact (distance, index) ranking of its short list; overflowed lists go to `slow`.
// dev_cap > 0: launched without the host knowing the count (optimistic run): the count is flagged[0]; a count beyond the
// cap or a list that overflowed cannot be handled here and raises opt[0] (the engine then repeats the run with host-checked
// searches); opt[1] accumulates the queries that came this way; the counters of the lists are left zeroed for the next use.
__global__ __launch_bounds__(256) void knn_exact_pick(const int32_t* __restrict__ flagged, int nflag, int dev_cap, int k,
                                                      int seeded, int32_t* __restrict__ xcnt, const double* __restrict__ xd,
                                                      const int32_t* __restrict__ xi, int32_t* __restrict__ idx_out,
                                                      double* __restrict__ dist_out, int32_t* __restrict__ slow,
                                                      int32_t* __restrict__ opt, int sq = 0) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (dev_cap > 0) {
        nflag = flagged[0];
        if (blockIdx.x == 0 && threadIdx.x == 0 && nflag > 0) {
            atomicAdd(&opt[1], nflag);
            if (nflag > dev_cap) opt[0] = 1;
        }
        if (nflag > dev_cap) {
            // (the run is given up -- but the searches queued behind this one still run, and their sweeps count into these
            // slots: left as they are, a later row would hold a reference twice, its merge (lk_merge_wave) two candidates of one
            // rank and an entry nobody wrote)
            if (f < dev_cap && lane == 0) xcnt[f] = 0;
            return;
        }
    }
    if (f >= nflag) return;
    const int q = flagged[1 + f];
    const int n = xcnt[f];
    if (dev_cap > 0 && lane == 0) xcnt[f] = 0;
    // fewer than k within the bound: only a seeded row (exactly the references within its seed distance: a short row,
    // padded with -1, is what the caller asked for); otherwise it cannot happen (the k candidates themselves qualify)
    const bool short_ok = seeded && n < k && n <= XF_CAP;
    if (n > XF_CAP || (n < k && !short_ok)) {
        if (lane == 0) {
            if (dev_cap > 0) {
                opt[0] = 1;
            } else {
                const int pos = atomicAdd(&slow[0], 1);
                slow[1 + pos] = q;
            }
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
            if (dist_out) dist_out[(int64_t)q * k + rank] = sq ? dm : sqrt(dm);
        }
    }
    if (short_ok)
        for (int m = n + lane; m < k; m += 64) idx_out[(int64_t)q * k + m] = -1;
}

__global__ void fill_f64(double* __restrict__ p, int n, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void fill_u32(uint32_t* __restrict__ p, int n, uint32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}


// rows of the flagged queries in the caller's query list: out[f] = rows ? rows[flagged[1 + f]] : flagged[1 + f]
__global__ void flagged_rows(const int32_t* __restrict__ flagged, int n, const int32_t* __restrict__ rows,
                             int32_t* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) {
        const int q = flagged[1 + f];
        out[f] = rows ? rows[q] : q;
    }
}

// results of a search over the flagged queries back into their rows: out[flagged[1 + f]][j] = sub[f][j]
__global__ void scatter_flagged(const int32_t* __restrict__ flagged, int n, int k, const int32_t* __restrict__ sub_idx,
                                const double* __restrict__ sub_dist, int32_t* __restrict__ idx_out,
                                double* __restrict__ dist_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * k) return;
    const int f = e / k, j = e - f * k;
    const int64_t o = (int64_t)flagged[1 + f] * k + j;
    idx_out[o] = sub_idx[e];
    if (dist_out) dist_out[o] = sub_dist[e];
}

// Developer experiment (testing hook "tau_replay", EXPERIMENTS.md round 6): what the full pass would cost if every query
// STARTED from the threshold it ends with.  1: every search of a run records its queries' final thresholds (the smallest over
// its reference ranges: each is at least the k-th best value of ITS range plus the margin, hence of the whole reference);
// 2: the same sequence of searches starts its full passes from the recording.  Any valid threshold leaves the results as they
// are; only the number of survivors changes.
__global__ void tau_record_kernel(const float* __restrict__ tau, int nchunks, int nq, uint32_t* __restrict__ rec) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float t = __builtin_inff();
    for (int c = 0; c < nchunks; ++c) t = fminf(t, tau[(int64_t)q * nchunks + c]);
    rec[q] = f32_orderable(t);
}
__global__ void tau_apply_kernel(const uint32_t* __restrict__ rec, int nq, uint32_t* __restrict__ tau_g) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) tau_g[q] = min(tau_g[q], rec[q]);
}

#ifndef BMX_SEEDED_SAMPLE
#define BMX_SEEDED_SAMPLE 4096
#endif
struct Tier {
    int id;  // 1 = fp16 single product, 2 = split bf16
    int NS, KS;
};

// the candidate tiers that take this shape, cheapest first
int candidate_tiers(int d, int k, int nr, Tier out[2]) {
    const int only = dev_knobs().knn_tier;  // testing hook: 1 / 2 = that tier only, 3 = exact scan only
    int n = 0;
    if (k > 36) return 0;
    const int KS1 = k <= 20 ? BMX_KS1 : 48, KS2 = k <= 20 ? 24 : 40;
    const int ns1 = f16_pick_ns(d, KS1);
    if (ns1 && nr > 2 * KS1 && (only == 0 || only == 1)) out[n++] = Tier{1, ns1, KS1};
    const int ns2 = bf16_pick_ns(d);
    if (ns2 && !(KS2 == 40 && ns2 > 16) && nr > 2 * KS2 && (only == 0 || only == 2)) out[n++] = Tier{2, ns2, KS2};
    return n;
}

// One candidate pass + refine over the queries (Qs, qrs)[0, nq): certified rows of io / dout are final; the others
// are listed in `flagged` (count in flagged[0]) with their k-th candidate distance in flag_bound.
void candidate_pass(hipStream_t stream, KnnWorkspace& ws, const Tier& T, const double* X, const int32_t* ref_rows, int nr,
                    const double* Qs, const int32_t* qrs, int nq, int d, int k, int32_t* io, double* dout,
                    int32_t* flagged, double* flag_bound, const float* seed_d2, const double* centre, double* kth_out) {
    const int NS = T.NS, KS = T.KS;
    ws.last_variant = T.id == 1 ? 3 : 2;
    // queries per workgroup: 8 consumer waves of 32 in the fp16 kernel; 8 or 4 (long rows, long lists) in the bf16 kernel
    const int unit = T.id == 1 ? 256 : 32 * bf16_ncons(NS, KS);
    const int nq_pad = (int)round_up(nq, unit);
    const int nqb = nq_pad / unit;

    // Reference ranges.  A short sample range [0, S) runs first and hands every query a valid starting threshold, so
    // selection is tight from the first tile of the full pass.  One workgroup per CU: the query blocks that fill whole
    // rounds of 256 workgroups sweep the reference as ONE range (tightest thresholds, one list per query); only the
    // remaining b blocks are split into c ranges, chosen so that their b * c short items fill the last round evenly.
    // Measured work per pair evaluation relative to one range (100k x 400k): 3 ranges 1.12, 5: 1.17, 7: 1.21.
    // Sample size: each sampled row costs every query a filter-only tile visit, each row NOT sampled costs KS / row
    // more candidates to append and select (the harmonic tail of the running threshold) -- with the fp16 kernel's
    // cheap tiles the balance is flat between 16k and 32k rows at 50 PCs and lower for longer rows (a sampled tile
    // costs the full matrix work).  The fp16 kernel samples small references too (a quarter of the rows): its sweep
    // hands every survivor to another wave, which makes a start without thresholds expensive
    const bool no_margin_knob = dev_knobs().no_margin != 0;
    int S_auto = nr >= 32768 ? 4096 : 0;
    if (T.id == 1 && nr >= 4096) S_auto = std::max(1024, std::min(nr / 4, NS <= 4 ? 24576 : 12288));
    // (a seeded search samples too, but a sixth of the rows: the odd query whose seed is loose -- a left cell listed by
    // one far-away right cell -- then starts from a sampled threshold instead of none; the tighter of the two counts)
    if (seed_d2 && T.id == 1) S_auto = std::min(S_auto, BMX_SEEDED_SAMPLE);
    int S = (int)round_up(dev_knobs().sample >= 0 ? dev_knobs().sample : S_auto, T.id == 1 ? f16_rows_per_slot(NS, KS) : 64);
    // A search with few query blocks (a rank's slice of a multi-GPU run: 12 500 queries = 49 blocks for 256 CUs) would sweep
    // the sample with a fifth of the chip, every block over all of it: the sample is split into CS ranges instead, block x
    // range items fill the CUs, each range of at least 24 ring slots (a lane keeps the KS / 2 <= 24 smallest of its per-slot
    // minima) hands its queries a threshold of its own and the tightest one counts (atomicMin on the shared word).
    // Round 6: the ranges leave their lists and the k-th smallest of the union is taken (f16_sample_merge): the threshold of the
    // unsplit sample, so the split is on by default (testing hook "sample_split": 0 never, n that many ranges).
    int CS = 1;
    if (T.id == 1 && S > 0 && nqb < 128 && dev_knobs().sample_split != 0 && !no_margin_knob) {
        const int rs = f16_rows_per_slot(NS, KS);
        CS = std::max(1, std::min({256 / std::max(nqb, 1), 8, S / (8 * rs)}));
        if (dev_knobs().sample_split > 0) CS = std::max(1, std::min({dev_knobs().sample_split, S / rs, 512 / KS}));
    }
    int C = 1, n_full = 0;
    {
        const int a = nqb / 256, b = nqb % 256;
        n_full = a * 256;
        if (b > 0) {
            double best = 1e30;
            for (int c = 1; c <= MAX_CHUNKS - 1; ++c) {
                if (c > 1 && nr / c < 2048) break;
                const double tailc = std::ceil((double)b * c / 256.0) / c * (1.0 + 0.105 * std::log((double)c));
                if (tailc < best - 1e-9) {
                    best = tailc;
                    C = c;
                }
            }
        }
    }
    if (dev_knobs().split_c > 0) C = std::max(1, dev_knobs().split_c);
    if (dev_knobs().force_c > 0) {
        C = std::max(1, std::min(MAX_CHUNKS - 1, dev_knobs().force_c));
        n_full = 0;
    }
    const int rmul = T.id == 1 ? f16_rows_per_slot(NS, KS) : 32;  // the fp16 ring hands two (or four) tiles over at a time
    const int chunk_len = (int)round_up(cdiv(nr, C), rmul);
    C = std::max(1, cdiv(nr, chunk_len));
    const int nr_pad = chunk_len * C;
    const int nchunks = C;
    S = std::min(S, nr_pad / rmul * rmul);  // (a forced sample size beyond the reference: the prepared image ends at nr_pad)
    if (debug_prints())
        fprintf(stderr, "[bmx] knn tier %d: nq=%d nr=%d d=%d NS=%d KS=%d S=%d C=%d chunk=%d full-range blocks=%d of %d\n", T.id,
                nq, nr, d, NS, KS, S, C, chunk_len, C > 1 ? n_full : nqb, nqb);

    const int KPw = 8 * NS;  // prepared row width in 4-byte words
    float* pq = ws.pq.reserve((size_t)nq_pad * KPw);
    float* pr = ws.pr.reserve((size_t)(nr_pad + 256) * KPw);  // + tail padding: the producers' prefetches over-read
    double* qn2 = ws.qn2.reserve(nq_pad);
    double* rn2 = ws.rn2.reserve(nr_pad);
    double* mean = ws.mean.reserve((size_t)d + 2);
    unsigned long long* maxbits = reinterpret_cast<unsigned long long*>(mean + d);
    int32_t* cand = ws.cand.reserve((size_t)nq_pad * nchunks * KS);
    float* cand_v = ws.cand_v.reserve((size_t)nq_pad * nchunks * KS);
    float* tau = ws.tau.reserve((size_t)nq_pad * nchunks);
    uint32_t* tau_g = ws.tau_g.reserve(nq_pad);
    unsigned long long* slots = ws.maxslots.reserve(64 * 16);

    // centre of the reference: any vector is valid (the error bound uses the norms actually obtained), a point
    // near the mean keeps it tight -- the mean of a strided sample of <= 16k rows costs next to nothing
    // (the merge engine already holds the column mean of the reference's node and hands it in: nothing to compute)
    if (!centre) {
        const int cstride = std::max(1, nr / 16384);
        const int ncs = cdiv(nr, cstride);
        const int rpb = 256;
        const int nb = cdiv(ncs, rpb);
        double* red = ws.red.reserve((size_t)nb * d);
        hipLaunchKernelGGL(colsum_partial, dim3(nb), dim3(256), 0, stream, X, ref_rows, ncs, d, rpb, cstride, red);
        BMX_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final, dim3(cdiv(d, 64)), dim3(64), 0, stream, red, nb, d, 1.0 / ncs, mean);
        BMX_LAUNCH_CHECK();
        centre = mean;
    }
    PassEps pe;
    if (T.id == 1) {
        pe.eps_k = 16.0 * NS;                                              // f32 accumulation over the K columns
        pe.eps_qr = 2.0;                                                   // one product block, <= 2 |q||r|
        pe.eps_split = 2.0 * (0.0009765625 + 2.384185791015625e-07);       // both operands rounded to fp16: 2^-10 (1 + 2^-12) * 2|q||r|
        pe.eps_den = 6.103515625e-05 * (std::sqrt((double)d) + 3.0);       // 2^-14 per flushed input, scaled units
        pe.scaled = 1;
    } else {
        pe.eps_k = 16.0 * NS;                                              // f32 accumulation over the concatenated K
        pe.eps_qr = 6.0;                                                   // three product blocks, each <= 2 |q||r|
        pe.eps_split = 1.5 * 3.03 * 2.0 * 1.52587890625e-05;               // dropped ql.rl, qh.r3, q3.rh: 3.03 * 2^-16 * 2|q||r|
        pe.eps_den = 0.0;
        pe.scaled = 0;
    }
    const double eps_k = pe.eps_k, eps_qr = pe.eps_qr, eps_split = pe.eps_split, eps_den = pe.eps_den;
    Bf16Launch L{reinterpret_cast<const uint16_t*>(pq), reinterpret_cast<const uint16_t*>(pr), nqb, 0, S, 1, S, 0, nchunks,
                 tau_g, 1, cand, cand_v, tau};
    const bool no_margin = dev_knobs().no_margin != 0;  // testing hook: the KS-th-best cut only
    auto go = [&](const Bf16Launch& l) {
        return T.id == 1 ? f16_launch(stream, ws, NS, KS, l) : bf16_launch(stream, ws, NS, KS, l);
    };
    bool ok = true;
    unsigned long long* zero_slots = nullptr;
    if (T.id == 1) {
        // fp16 tier: references (norms, then image) and queries (image, norm, margin, seed threshold) in TWO launches; the
        // slot maxima are folded by the second one, which also resets the flagged-query counter (knn_f16.hip)
        if (!ws.slots_clean) BMX_HIP(hipMemsetAsync(slots, 0, 64 * 16 * sizeof(unsigned long long), stream));
        float* margin = no_margin ? nullptr : ws.margin.reserve(nq_pad);
        uint32_t* tau_seed = seed_d2 ? ws.tau_seed.reserve(nq_pad) : nullptr;
        // without a sample pass the thresholds start at the seed (or at +inf): prep writes them
        f16_prep_all(stream, X, ref_rows, nr, nr_pad, Qs, qrs, nq, nq_pad, d, NS, centre, reinterpret_cast<uint16_t*>(pr),
                     reinterpret_cast<uint16_t*>(pq), rn2, qn2, maxbits, slots, flagged, margin, pe, seed_d2, tau_seed,
                     S == 0 || CS > 1 ? tau_g : nullptr);
        if (margin) {
            L.margin = margin;
            L.k = k;
        }
        L.tau_seed = tau_seed;
        zero_slots = slots;  // knn_refine leaves them zeroed for the next search
        ws.slots_clean = false;
        if (S > 0) {
            float* lists = nullptr;
            if (CS > 1) {  // block x range items over the sample rows [0, S)
                const int rs = f16_rows_per_slot(NS, KS);
                L.range_len = (int)round_up(cdiv(S, CS), rs);
                L.nranges = cdiv(S, L.range_len);
                if (L.nranges > 1 && margin) {
                    lists = ws.samp_lists.reserve((size_t)nq_pad * L.nranges * KS);
                    L.cand_v = lists;  // (a sample pass has no other use for it)
                }
            }
            ok = go(L);
            if (lists) f16_sample_merge(stream, lists, L.nranges, KS, nq, k, margin, tau_g);
            L.cand_v = cand_v;
        }
    } else {
        ws.slots_clean = false;
        BMX_HIP(hipMemsetAsync(maxbits, 0, sizeof(unsigned long long), stream));
        bf16_prep(stream, X, ref_rows, nr, nr_pad, d, NS, centre, 0, reinterpret_cast<uint16_t*>(pr), rn2, maxbits, slots);
        bf16_prep(stream, Qs, qrs, nq, nq_pad, d, NS, centre, 1, reinterpret_cast<uint16_t*>(pq), qn2, maxbits, slots);
        // sample pass: threshold estimation over rows [0, S); full pass: every row, starting from that threshold
        if (S == 0) {  // no sample: +inf everywhere (0xFF800000 is the orderable image of +inf)
            hipLaunchKernelGGL(fill_u32, dim3(cdiv(nq_pad, 256)), dim3(256), 0, stream, tau_g, nq_pad, 0xFF800000u);
            BMX_LAUNCH_CHECK();
        }
        if (S > 0) ok = go(L);
        if (seed_d2) {  // tau_g = min(sampled threshold, seed threshold)
            hipLaunchKernelGGL(seed_tau_kernel, dim3(cdiv(nq_pad, 256)), dim3(256), 0, stream, seed_d2, qn2, maxbits, nq, nq_pad,
                               pe, tau_g);
            BMX_LAUNCH_CHECK();
        }
        BMX_HIP(hipMemsetAsync(flagged, 0, sizeof(int32_t), stream));
    }
    L.first_begin = 0;
    L.range_len = chunk_len;
    L.nranges = C;
    L.n_full = C > 1 ? n_full : 0;
    L.r_limit = nr_pad;
    L.sample = 0;
    const int replay = T.id == 1 ? dev_knobs().tau_replay : 0;
    if (replay == 2 && ws.replay_idx < ws.tau_rec.size() && ws.tau_rec[ws.replay_idx].cap >= (size_t)nq_pad) {
        hipLaunchKernelGGL(tau_apply_kernel, dim3(cdiv(nq_pad, 256)), dim3(256), 0, stream,
                           (const uint32_t*)ws.tau_rec[ws.replay_idx].p, nq_pad, tau_g);
        BMX_LAUNCH_CHECK();
    }
    ok = ok && go(L);
    if (replay == 1) {
        if (ws.tau_rec.size() <= ws.replay_idx) ws.tau_rec.resize(ws.replay_idx + 1);
        hipLaunchKernelGGL(tau_record_kernel, dim3(cdiv(nq_pad, 256)), dim3(256), 0, stream, (const float*)tau, nchunks, nq_pad,
                           ws.tau_rec[ws.replay_idx].reserve((size_t)nq_pad));
        BMX_LAUNCH_CHECK();
    }
    if (replay) ++ws.replay_idx;
    if (!ok) throw Error(BMX_ERR_ARG, "kNN: unsupported padded dimension");
    {
        // rows of an even number of doubles are 16-byte aligned: the quad gather, instantiated for the row length.  The queries
        // of the whole-range blocks (one list of <= 32 candidates each) take half a wave each, the others a wave
        const int need = (d & 1) ? 0 : cdiv(d, 8);
        const int unit_q = T.id == 1 ? 256 : 32 * bf16_ncons(NS, KS);
        const int nq_half = KS == 32 && !dev_knobs().refine_wave ? (C > 1 ? std::min(nq, n_full * unit_q) : nq) : 0;
        const int sq = ws.dist_squared ? 1 : 0;
#define BMX_REFINE(NC)                                                                                                              \
    do {                                                                                                                            \
        if (nq_half > 0)                                                                                                            \
            hipLaunchKernelGGL(knn_refine_half<NC>, dim3(cdiv(nq_half, 8)), dim3(256), 0, stream, X, ref_rows, Qs, qrs, nq_half, d, k, \
                               nchunks, eps_k, eps_qr, eps_split, eps_den, T.id == 1 ? 1 : 0, cand, cand_v, tau, qn2, maxbits,       \
                               seed_d2, io, dout, flagged, flag_bound, zero_slots, kth_out, sq);                                    \
        if (nq > nq_half)                                                                                                           \
            hipLaunchKernelGGL(knn_refine<NC>, dim3(cdiv(nq - nq_half, 4)), dim3(256), 0, stream, X, ref_rows, Qs, qrs, nq, d, k, KS,  \
                               nchunks, eps_k, eps_qr, eps_split, eps_den, T.id == 1 ? 1 : 0, cand, cand_v, tau, qn2, maxbits,       \
                               seed_d2, io, dout, flagged, flag_bound, nq_half > 0 ? nullptr : zero_slots, nq_half, kth_out, sq);   \
    } while (0)
        if (need == 0 || need > 16) BMX_REFINE(0);
        else if (need <= 2) BMX_REFINE(2);
        else if (need <= 4) BMX_REFINE(4);
        else if (need <= 7) BMX_REFINE(7);
        else if (need <= 10) BMX_REFINE(10);
        else if (need <= 13) BMX_REFINE(13);
        else BMX_REFINE(16);
#undef BMX_REFINE
    }
    BMX_LAUNCH_CHECK();
    if (zero_slots) ws.slots_clean = true;
    if (debug_prints()) {
        std::vector<int32_t> hc((size_t)nq * nchunks * KS);
        BMX_HIP(hipMemcpyAsync(hc.data(), cand, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        ws.sync(stream);
        size_t valid = 0;
        for (int32_t v : hc) valid += v >= 0;
        fprintf(stderr, "[bmx] candidates per query after the top-k pass: %.1f (of %d slots)\n", (double)valid / nq,
                nchunks * KS);
    }
}

// one device word to the host: stored into pinned memory by a one-thread kernel, sequence number last, and the host spins
// on that word (no copy to schedule, signal and poll through the runtime: ~15 us instead of ~100 of idle GPU per read-back)
__global__ void publish_word(const int32_t* __restrict__ dev, int32_t* host, int seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t v = *dev;
    __hip_atomic_store(&host[0], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int read_count(hipStream_t stream, KnnWorkspace& ws, const int32_t* dev) {
    int32_t* h = reinterpret_cast<int32_t*>(ws.pinned_words());  // word 0 of the workspace's pinned block: [value, sequence]
    if (ws.read_seq == 0) h[1] = 0;  // (a pooled block may hold an earlier owner's numbers)
    const int seq = ++ws.read_seq;
    hipLaunchKernelGGL(publish_word, dim3(1), dim3(64), 0, stream, dev, h, seq);
    BMX_LAUNCH_CHECK();
    const double budget = ws.wd_budget_s;
    const auto t0 = std::chrono::steady_clock::now();
    volatile int32_t* vp = h;
    unsigned spins = 0;
    while (vp[1] != seq) {
        if ((++spins & 0x3FFu) == 0) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (budget > 0.0 && el > budget)
                throw WatchdogTimeout("watchdog: the GPU work queued on the engine's stream did not finish in time; the engine is "
                                      "dead (restart the process)");
            if (el > 0.5) {
                const hipError_t e = hipStreamQuery(stream);
                (void)hipGetLastError();
                if (e != hipSuccess && e != hipErrorNotReady)
                    throw Error(BMX_ERR_HIP, std::string("the search's stream failed: ") + hipGetErrorString(e));
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return h[0];
}

}  // namespace

DevKnobs& dev_knobs() {
    static DevKnobs k;
    return k;
}

bool debug_prints() {  // BMX_DEBUG=1: the shape decisions of every search (slow: lists are copied back for their counts)
    static const bool on = std::getenv("BMX_DEBUG") != nullptr && std::getenv("BMX_DEBUG")[0] != 't';
    return on;
}

bool debug_timings() {  // BMX_DEBUG=t (or 1): host-side timings of the one-shot call
    static const bool on = std::getenv("BMX_DEBUG") != nullptr;
    return on;
}

void guarded_stream_sync(hipStream_t stream, double budget_s) {
    if (!(budget_s > 0.0)) {
        BMX_HIP(hipStreamSynchronize(stream));
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            throw Error(BMX_ERR_HIP, std::string("hipStreamQuery failed: ") + hipGetErrorString(e));
        }
        (void)hipGetLastError();  // hipErrorNotReady is sticky for hipGetLastError otherwise
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > budget_s) {
            char msg[200];
            std::snprintf(msg, sizeof(msg),
                          "watchdog: the GPU work queued on the engine's stream did not finish within %.1f s; the engine is "
                          "dead (restart the process)", budget_s);
            throw WatchdogTimeout(msg);
        }
        if (el > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(100));  // short waits spin, long ones sleep
    }
}

namespace {

// Exact FP64 search: `count` queries listed in scan-list form (list[1 + f], nullptr = queries 0 .. count-1).  With
// bounds (the k-th candidate distance of each listed query) a bounded filter pass runs first; what overflows it, or
// everything when there are no bounds, takes the full scan.
void exact_search(hipStream_t stream, KnnWorkspace& ws, const double* X, const int32_t* ref_rows, int nr, const double* Qs,
                  const int32_t* qrs, int d, int k, int32_t* io, double* dout, const int32_t* list,
                  const double* bounds, int count, bool seeded = false) {
    if (count <= 0) return;
    ws.last_exact += count;
    const int32_t* scan_list = list;
    if (list && bounds) {
        int32_t* xcnt = ws.xcnt.reserve((size_t)count);
        double* xd = ws.xd.reserve((size_t)count * XF_CAP);
        int32_t* xi = ws.xi.reserve((size_t)count * XF_CAP);
        int32_t* slow = ws.slow.reserve((size_t)count + 1);
        ws.xcnt_clear = false;  // (the lists' counters are left as the sweep filled them)
        BMX_HIP(hipMemsetAsync(xcnt, 0, (size_t)count * sizeof(int32_t), stream));
        BMX_HIP(hipMemsetAsync(slow, 0, sizeof(int32_t), stream));
        const size_t lds = (size_t)(XF_TILE + XF_QCH) * (d + 1) * sizeof(double);
        ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_exact_filter), lds);
        hipLaunchKernelGGL(knn_exact_filter, dim3(std::min(cdiv(nr, XF_TILE), 2048)), dim3(256), lds, stream, X, ref_rows, nr, Qs, qrs,
                           d, list, bounds, count, 0, xcnt, xd, xi);
        BMX_LAUNCH_CHECK();
        hipLaunchKernelGGL(knn_exact_pick, dim3(cdiv(count, 4)), dim3(256), 0, stream, list, count, 0, k, seeded ? 1 : 0, xcnt,
                           xd, xi, io, dout, slow, (int32_t*)nullptr, ws.dist_squared ? 1 : 0);
        BMX_LAUNCH_CHECK();
        count = read_count(stream, ws, slow);
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
            hipLaunchKernelGGL(knn_exact_dist, dim3(cdiv(nr, XDT), cdiv(nb, XDT)), dim3(256), 0, stream, X, ref_rows, nr, Qs, qrs,
                               d, scan_list, f0, nb, drow);
            BMX_LAUNCH_CHECK();
            if (k > 256 && k <= XSB_MAXK) {
                int np2 = 128;
                while (np2 < k) np2 <<= 1;
                const size_t lds = (size_t)np2 * (sizeof(double) + sizeof(int));
                ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_exact_select_big), lds);
                hipLaunchKernelGGL(knn_exact_select_big, dim3(nb), dim3(XSB_T), lds, stream, drow, nr, k, np2, scan_list, f0, io,
                                   dout, 0, ws.dist_squared ? 1 : 0);
            } else {
                hipLaunchKernelGGL(knn_exact_select, dim3(nb), dim3(256), 0, stream, drow, nr, k, scan_list, f0, io, dout, 0,
                                   ws.dist_squared ? 1 : 0);
            }
            BMX_LAUNCH_CHECK();
        }
    }
}

// queries (Qs, qrs)[0, nq) through the tiers tiers[t ...]
void search_tiers(hipStream_t stream, KnnWorkspace& ws, const Tier* tiers, int ntiers, int t, const double* X,
                  const int32_t* ref_rows, int nr, const double* Qs, const int32_t* qrs, int nq, int d, int k, int32_t* io,
                  double* dout, const float* seed_d2 = nullptr, const double* centre = nullptr, double* kth_out = nullptr) {
    if (t >= ntiers) {
        exact_search(stream, ws, X, ref_rows, nr, Qs, qrs, d, k, io, dout, nullptr, nullptr, nq);
        return;
    }
    int32_t* flagged = ws.flagged_t[t].reserve((size_t)nq + 1);
    double* bound = ws.flag_bound_t[t].reserve((size_t)nq + 1);
    // (what a seeded pass cannot settle goes on unseeded: the full k nearest serve the caller just as well)
    // (kth_out: the first tier's re-rank writes it for the rows it certifies; a row any later tier or the exact sweep rewrites
    // keeps +inf there, which only costs the intersection's probe a row read)
    candidate_pass(stream, ws, tiers[t], X, ref_rows, nr, Qs, qrs, nq, d, k, io, dout, flagged, bound, seed_d2, centre,
                   t == 0 ? kth_out : nullptr);
    if (ws.optimistic && t == 0) {
        // Optimistic run (the merge engine): no read-back inside a search.  Practically every query is certified by the
        // first tier (config 3: 1-6 of 3 million per step are not), so the bounded FP64 sweep for the few that are not is
        // launched unconditionally with the count left on the device -- with nothing flagged its workgroups end at once.
        // More than OPT_CAP flagged queries, or a list that overflows (massive exact ties), raises opt_state[0]; the engine
        // looks at it at its next wait and repeats the run with host-checked searches.
        int32_t* opt = ws.opt_state_ptr(stream);
        int32_t* xcnt = ws.xcnt.reserve((size_t)KnnWorkspace::OPT_CAP);
        if (!ws.xcnt_clear) {
            BMX_HIP(hipMemsetAsync(xcnt, 0, (size_t)KnnWorkspace::OPT_CAP * sizeof(int32_t), stream));
            ws.xcnt_clear = true;
        }
        double* xd = ws.xd.reserve((size_t)KnnWorkspace::OPT_CAP * XF_CAP);
        int32_t* xi = ws.xi.reserve((size_t)KnnWorkspace::OPT_CAP * XF_CAP);
        const size_t lds = (size_t)(XF_TILE + XF_QCH) * (d + 1) * sizeof(double);
        ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_exact_filter), lds);
        hipLaunchKernelGGL(knn_exact_filter, dim3(std::min(cdiv(nr, XF_TILE), 2048)), dim3(256), lds, stream, X, ref_rows, nr, Qs, qrs,
                           d, flagged, bound, 0, KnnWorkspace::OPT_CAP, xcnt, xd, xi);
        BMX_LAUNCH_CHECK();
        hipLaunchKernelGGL(knn_exact_pick, dim3(KnnWorkspace::OPT_CAP / 4), dim3(256), 0, stream, flagged, 0,
                           KnnWorkspace::OPT_CAP, k, seed_d2 ? 1 : 0, xcnt, xd, xi, io, dout, (int32_t*)nullptr, opt,
                           ws.dist_squared ? 1 : 0);
        BMX_LAUNCH_CHECK();
        return;
    }
    // the number of uncertified queries decides what is launched next, so it is read back (one small synchronisation)
    const int count = read_count(stream, ws, flagged);
    if (count == 0) return;
    ws.last_flagged_tier[t] += count;
    // a few hundred leftovers are cheaper in one bounded FP64 sweep than in another candidate pass (prep of the whole
    // reference + a launch that cannot fill the chip)
    if (t + 1 < ntiers && count >= 256) {
        // the next candidate tier sees only these queries: a row list into the caller's query matrix, results into a
        // scratch block that is scattered back afterwards
        int32_t* rows2 = ws.sub_rows[t].reserve((size_t)count);
        hipLaunchKernelGGL(flagged_rows, dim3(cdiv(count, 256)), dim3(256), 0, stream, flagged, count, qrs, rows2);
        BMX_LAUNCH_CHECK();
        int32_t* sub_idx = ws.sub_idx[t].reserve((size_t)count * k);
        double* sub_dist = dout ? ws.sub_dist[t].reserve((size_t)count * k) : nullptr;
        search_tiers(stream, ws, tiers, ntiers, t + 1, X, ref_rows, nr, Qs, rows2, count, d, k, sub_idx, sub_dist, nullptr, centre);
        hipLaunchKernelGGL(scatter_flagged, dim3(cdiv((int64_t)count * k, 256)), dim3(256), 0, stream, flagged, count, k,
                           sub_idx, sub_dist, io, dout);
        BMX_LAUNCH_CHECK();
    } else {
        exact_search(stream, ws, X, ref_rows, nr, Qs, qrs, d, k, io, dout, flagged, bound, count, seed_d2 != nullptr);
    }
}

// ---------------------------------------------------------------------------------------------------
// k beyond what the candidate tiers' lists hold (k > 36: `prop.k` of R/MNN_tree.R:140-146, or simply k = 50).
// The reference is dealt into P strided partitions (row j of the list goes to partition j mod P: independent of how the
// caller ordered its cells), every partition is searched for its 36 (or 20) nearest with the whole certified cascade above, and
// the P x 36 candidates of a query are merged exactly: squared distances recomputed in the reference's order of operations
// (sum over the dimensions of (q - x)^2, no contraction), sorted by (distance, position) -- the exact search's own order --, the first
// k are the answer PROVIDED no partition's list ends inside them: a partition whose 36th neighbour ranks behind the k-th
// merged candidate cannot hold anything nearer that it did not list.  With P = ceil(k / 16) a partition holds 16 of the k
// on average; a query for which one holds more than 35 fails the test and goes to the exact scan.
// Cost: the matrix work of ONE search at k = 36 (P searches over 1 / P of the reference each) + P preparations.
// ---------------------------------------------------------------------------------------------------
constexpr int LK_MAXE = 2048;  // candidates merged per query

// seed[q] = the first partition's kp-th distance squared, rounded up to f32 (+inf where that row was not certified): sqrt,
// then squared again -- never below the true value
__global__ void lk_seed_kernel(const double* __restrict__ kth, int nq, float* __restrict__ seed) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const double dd = kth[q] * kth[q] * (1.0 + 1e-15);
    float f = (float)dd;
    if ((double)f < dd) f = __uint_as_float(__float_as_uint(f) + 1u);  // dd >= 0: the next float up
    seed[q] = dd < __builtin_inf() ? f : __builtin_inff();
}

// rows[off_p + j] = the (p + j P)-th row of the caller's reference list, partition-major
__global__ void lk_partition_rows(const int32_t* __restrict__ ref_rows, int nr, int P, int32_t* __restrict__ rows) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nr) return;
    const int p = g % P, j = g / P;
    // partitions 0 .. (nr % P) - 1 hold one row more: offset of partition p = p * (nr / P) + min(p, nr % P)
    const int base = nr / P, rem = nr % P;
    rows[(int64_t)p * base + (p < rem ? p : rem) + j] = ref_rows ? ref_rows[g] : g;
}

// kp: neighbours asked of every partition (sub_idx [P][nq][kp])
__global__ __launch_bounds__(256) void lk_merge(const double* __restrict__ X, const int32_t* __restrict__ rows, int nr, int P,
                                                int kp, const double* __restrict__ Q, const int32_t* __restrict__ qrs, int nq,
                                                int d, int k, const int32_t* __restrict__ sub_idx, int32_t* __restrict__ idx_out,
                                                double* __restrict__ dist_out, int32_t* __restrict__ flagged,
                                                int32_t* __restrict__ opt, double* __restrict__ kth_out,
                                                const float* __restrict__ seed_d2, const double* __restrict__ sub_d2) {
    __shared__ double kd[LK_MAXE];
    __shared__ int32_t ki[LK_MAXE];
    __shared__ int sh_fail;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int E = P * kp;
    int np2 = 256;
    while (np2 < E) np2 <<= 1;
    const double* qv = Q + (int64_t)(qrs ? qrs[q] : q) * d;
    const int base = nr / P, rem = nr % P;
    for (int e = tid; e < np2; e += 256) {
        double d2 = __builtin_inf();
        int g = 0x7fffffff;
        if (e < E) {
            const int p = e / kp, j = e - p * kp;
            const int l = sub_idx[((int64_t)p * nq + q) * kp + j];  // position inside partition p
            if (l >= 0) {
                if (sub_d2) {  // (the partition's search left the very sum below, squared: nothing to gather)
                    d2 = sub_d2[((int64_t)p * nq + q) * kp + j];
                } else {
                    const double* x = X + (int64_t)rows[(int64_t)p * base + (p < rem ? p : rem) + l] * d;
                    double s = 0.0;
                    for (int c = 0; c < d; ++c) {
                        const double t = qv[c] - x[c];
                        s += t * t;
                    }
                    d2 = s;
                }
                g = l * P + p;  // its position in the caller's reference list
            }
        }
        kd[e] = d2;
        ki[e] = g;
    }
    if (tid == 0) sh_fail = 0;
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const double a = kd[i], b = kd[ixj];
                    const int ia = ki[i], ib = ki[ixj];
                    const bool gt = a > b || (a == b && ia > ib);
                    if (((i & kk) == 0) ? gt : !gt) {
                        kd[i] = b;
                        kd[ixj] = a;
                        ki[i] = ib;
                        ki[ixj] = ia;
                    }
                }
            }
            __syncthreads();
        }
    for (int r = tid; r < k; r += 256) {
        idx_out[(int64_t)q * k + r] = ki[r];
        if (dist_out) dist_out[(int64_t)q * k + r] = sqrt(kd[r]);
    }
    // no partition's list may end inside the first k: its last entry is the kp-th of its partition -- found among the sorted
    // candidates by its position (unique) -- and must rank at k or later
    for (int r = tid; r < k; r += 256) {
        const int g = ki[r], p = g % P, l = g / P;
        if (sub_idx[((int64_t)p * nq + q) * kp + kp - 1] == l) sh_fail = 1;
    }
    // partitions searched within the seed distance (large_k_search) are complete up to it only: the k-th must lie inside
    if (tid == 0 && seed_d2 && !(kd[k - 1] < (double)seed_d2[q])) sh_fail = 1;
    __syncthreads();
    // (a row that fails is rewritten by an exact path -- unless an optimistic run has more of them than it takes on the device,
    // and then the kernels queued behind this search still read the row before the engine starts over: every entry a valid
    // position.  Seeded partitions can leave fewer than k candidates, i.e. empty places among the first k.)
    if (sh_fail)
        for (int r = tid; r < k; r += 256) idx_out[(int64_t)q * k + r] = 0;
    // (the intersection's probe skips a row whose k-th distance it knows to be too small: exact here when the merge stands)
    if (tid == 0 && kth_out) kth_out[q] = sh_fail ? __builtin_inf() : sqrt(kd[k - 1]);
    if (tid == 0 && sh_fail) {
        if (opt) atomicOr(opt, 1);  // optimistic run: the engine repeats it with host-checked searches
        else flagged[1 + atomicAdd(&flagged[0], 1)] = q;
    }
}

// The same merge for up to 512 candidates a query (k <= 224 with lists of 36), one WAVE per query and no sort: the partitions'
// lists arrive sorted by (squared distance, position) -- the order the merged list wants --, so a candidate's place in the
// merged list is the number of candidates in front of it: its place in its own list plus, for every other partition, a binary
// search.  (A 256-thread bitonic sort per query took 3 ms a search at config 2, a third of the whole.)
__global__ __launch_bounds__(256) void lk_merge_wave(const double* __restrict__ X, const int32_t* __restrict__ rows, int nr, int P,
                                                     int kp, const double* __restrict__ Q, const int32_t* __restrict__ qrs,
                                                     int nq, int d, int k, const int32_t* __restrict__ sub_idx,
                                                     int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                     int32_t* __restrict__ flagged, int32_t* __restrict__ opt,
                                                     double* __restrict__ kth_out, const float* __restrict__ seed_d2,
                                                     const double* __restrict__ sub_d2) {
    __shared__ double kd_[4][512];
    __shared__ int32_t ki_[4][512];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;  // (whole waves: nothing below synchronises the block)
    double* kd = kd_[w];
    int32_t* ki = ki_[w];
    const int E = P * kp;
    const double* qv = Q + (int64_t)(qrs ? qrs[q] : q) * d;
    const int base = nr / P, rem = nr % P;
    for (int e = lane; e < E; e += 64) {
        const int p = e / kp, j = e - p * kp;
        const int l = sub_idx[((int64_t)p * nq + q) * kp + j];
        if (sub_d2) {  // (the partition's search left the very sum below, squared: nothing to gather)
            kd[e] = l >= 0 ? sub_d2[((int64_t)p * nq + q) * kp + j] : __builtin_inf();
            ki[e] = l >= 0 ? l * P + p : 0x7fffffff;
            continue;
        }
        const double* x = X + (int64_t)rows[(int64_t)p * base + (p < rem ? p : rem) + (l >= 0 ? l : 0)] * d;
        // (eight coordinates asked for at a time -- a random row: the round trips are what this costs --, summed in order)
        double s = 0.0;
        int c = 0;
        for (; c + 8 <= d; c += 8) {
            double xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = x[c + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double t = qv[c + u] - xv[u];
                s += t * t;
            }
        }
        for (; c < d; ++c) {
            const double t = qv[c] - x[c];
            s += t * t;
        }
        kd[e] = l >= 0 ? s : __builtin_inf();
        ki[e] = l >= 0 ? l * P + p : 0x7fffffff;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // (every entry of the row a valid position whatever the lists hold: lists spoilt by a search that an optimistic run has
    // already given up on must not leave an entry unwritten for the kernels queued behind it)
    for (int r = lane; r < k; r += 64) idx_out[(int64_t)q * k + r] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    bool fail = false;
    double kthv = -1.0;  // the k-th merged candidate's squared distance, on the lane that ranks it
    for (int e = lane; e < E; e += 64) {
        const int p = e / kp, j = e - p * kp;
        const double de = kd[e];
        const int ge = ki[e];
        int rank = j;
        for (int p2 = 0; p2 < P; ++p2) {
            if (p2 == p) continue;
            int lo = 0, hi = kp;  // entries of list p2 in front of (de, ge)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const double dm = kd[p2 * kp + mid];
                const int gm = ki[p2 * kp + mid];
                if (dm < de || (dm == de && gm < ge)) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            idx_out[(int64_t)q * k + rank] = ge;
            if (dist_out) dist_out[(int64_t)q * k + rank] = sqrt(de);
            fail |= j == kp - 1;  // a partition's list ends inside the first k
            if (rank == k - 1) kthv = de;
        }
    }
    bool any_fail = __builtin_amdgcn_ballot_w64(fail) != 0;
    for (int o = 32; o > 0; o >>= 1) kthv = fmax(kthv, __shfl_xor(kthv, o));
    // partitions searched within the seed distance (large_k_search) are complete up to it only: the k-th must lie inside
    if (seed_d2 && !(kthv >= 0.0 && kthv < (double)seed_d2[q])) any_fail = true;
    if (any_fail)  // (see lk_merge: every entry of a failed row a valid position; later in the wave's program order than the ranks' stores)
        for (int r = lane; r < k; r += 64) idx_out[(int64_t)q * k + r] = 0;
    if (kth_out && lane == 0) kth_out[q] = any_fail || kthv < 0.0 ? __builtin_inf() : sqrt(kthv);
    if (any_fail && lane == 0) {
        if (opt) atomicOr(opt, 1);
        else flagged[1 + atomicAdd(&flagged[0], 1)] = q;
    }
}

// The merge for k beyond ~900 (`prop.k` = 0.05 of 100 000 cells is k = 5 000, R/MNN_tree.R:140-146): up to LKB_MAXE candidates a
// query, one workgroup per query (LKB_T threads: 256 up to 3 584 candidates, 512 up to 7 168, 1 024 beyond).  No rank counting (E x P binary searches would be 2e7 steps a query at
// P = 358) and no sort of all E: every thread keeps its <= 14 candidates in registers; the k-th smallest (squared distance,
// position) is found by bisection on the distances' bit patterns -- non-negative doubles order like unsigned integers; a round is
// a count and a block reduction --, the k selected candidates are compacted into the LDS and only THEY are sorted (bitonic
// network over the next power of two: 12 bytes an entry, 96 KB at k = 5 000).  The certificate is the partitioned search's: no
// partition's last entry may be among the selected.
constexpr int LKB_PER = 14;                    // candidates a thread holds
constexpr int LKB_MAXE = 1024 * LKB_PER;       // 14 336 (1 024 threads; up to 3 584 candidates: 256 threads -- a block-wide
                                               // barrier of 4 waves instead of 16, ~200 of them a query, and eight queries a CU)
constexpr int LKB_MAXK = 8192;                 // the sort's entries: 12 B x 8 192 = 96 KB of LDS

template <int LKB_T>
__global__ __launch_bounds__(LKB_T) void lk_merge_big(const double* __restrict__ X, const int32_t* __restrict__ rows, int nr, int P,
                                                      int kp, const double* __restrict__ Q, const int32_t* __restrict__ qrs,
                                                      int nq, int d, int k, const int32_t* __restrict__ sub_idx,
                                                      int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                      int32_t* __restrict__ flagged, int32_t* __restrict__ opt,
                                                      double* __restrict__ kth_out, const float* __restrict__ seed_d2,
                                                      const double* __restrict__ sub_d2) {
    extern __shared__ __attribute__((aligned(16))) char lkb_smem[];
    __shared__ int sh_cnt[LKB_T / 64], sh_cb[2][LKB_T / 64];
    __shared__ int sh_fail, sh_base;
    const int E = P * kp, q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    double* kd = reinterpret_cast<double*>(lkb_smem);    // [np2] the selected candidates, then sorted
    int32_t* ki = reinterpret_cast<int32_t*>(kd + np2);  // [np2]
    const double* qv = Q + (int64_t)(qrs ? qrs[q] : q) * d;
    const int base = nr / P, rem = nr % P;
    // ---- exact squared distances in the reference's order of operations; this thread's candidates stay in registers
    unsigned long long mine[LKB_PER];
    int gmine[LKB_PER];
#pragma unroll
    for (int u = 0; u < LKB_PER; ++u) {
        const int e = tid + u * LKB_T;
        unsigned long long bits = 0x7FF0000000000000ull;  // +inf: beyond E, or an empty place of a list
        int g = 0x7fffffff;
        if (e < E) {
            const int p = e / kp, j = e - p * kp;
            const int l = sub_idx[((int64_t)p * nq + q) * kp + j];
            if (l >= 0 && sub_d2) {  // (the partition's search left the very sum below, squared: nothing to gather)
                bits = (unsigned long long)__double_as_longlong(sub_d2[((int64_t)p * nq + q) * kp + j]);
                g = l * P + p;
            } else if (l >= 0) {
                const double* x = X + (int64_t)rows[(int64_t)p * base + (p < rem ? p : rem) + l] * d;
                double s_ = 0.0;
                int c = 0;
                for (; c + 8 <= d; c += 8) {
                    double xv[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) xv[t] = x[c + t];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const double dd = qv[c + t] - xv[t];
                        s_ += dd * dd;
                    }
                }
                for (; c < d; ++c) {
                    const double dd = qv[c] - x[c];
                    s_ += dd * dd;
                }
                bits = (unsigned long long)__double_as_longlong(s_);
                g = l * P + p;  // its position in the caller's reference list
            }
        }
        mine[u] = bits;
        gmine[u] = g;
    }
    // block-wide count of this thread's candidates that satisfy pred
    // (a wave's count: one ballot per candidate place, counted on the scalar side; the waves' counts meet in one of two LDS rows
    // in turn -- one barrier a round: a row is written again two rounds later, behind the barrier of the round in between)
    int cb_par = 0;
    auto count_block = [&](auto pred) -> int {
        int c = 0;
#pragma unroll
        for (int u = 0; u < LKB_PER; ++u) c += __popcll(__builtin_amdgcn_ballot_w64(pred(mine[u], gmine[u])));
        if (lane == 0) sh_cb[cb_par][w] = c;
        __syncthreads();
        int t = 0;
        for (int ww = 0; ww < LKB_T / 64; ++ww) t += sh_cb[cb_par][ww];
        cb_par ^= 1;
        return t;
    };
    // ---- the k-th smallest distance: the smallest bit pattern with at least k candidates at or below it
    unsigned long long vk = 0;
    for (int b = 62; b >= 0; --b) {  // (bit 63 is the sign: never set)
        const unsigned long long trial = vk | ((1ull << b) - 1ull);  // the largest value with the bits decided so far and this bit clear
        const int c = count_block([&](unsigned long long v, int) { return v <= trial; });
        if (c < k) vk |= 1ull << b;
    }
    // partitions searched within the seed distance (large_k_search) are complete up to it only: the k-th must lie inside.
    // (Fewer than k candidates in all -- seeded lists may be short -- leaves vk beyond +inf: nothing to select, the query goes
    // to the exact scan.)  vk is the same on every thread.
    if (vk >= 0x7FF0000000000000ull || (seed_d2 && !(__longlong_as_double((long long)vk) < (double)seed_d2[q]))) {
        for (int r = tid; r < k; r += LKB_T) idx_out[(int64_t)q * k + r] = 0;  // (see lk_merge: every entry a valid position)
        if (tid == 0) {
            if (kth_out) kth_out[q] = __builtin_inf();
            if (opt) atomicOr(opt, 1);
            else flagged[1 + atomicAdd(&flagged[0], 1)] = q;
        }
        return;
    }
    // ... and among the candidates AT that distance the m with the smallest positions (exact duplicates: rare)
    const int below = count_block([&](unsigned long long v, int) { return v < vk; });
    const int ties = count_block([&](unsigned long long v, int) { return v == vk; });
    int gk = 0x7fffffff;
    if (ties > k - below) {
        int lo = 0;  // the smallest position bound with at least k - below ties at or below it
        for (int b = 30; b >= 0; --b) {
            const int trial = lo | ((1 << b) - 1);
            const int c = count_block([&](unsigned long long v, int g) { return v == vk && g <= trial; });
            if (c < k - below) lo |= 1 << b;
        }
        gk = lo;
    }
    auto selected = [&](unsigned long long v, int g) { return v < vk || (v == vk && g <= gk); };
    // ---- the certificate: no partition's last entry among the selected (it sits at e = p kp + kp - 1)
    if (tid == 0) sh_fail = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < LKB_PER; ++u) {
        const int e = tid + u * LKB_T;
        if (e < E && (e % kp) == kp - 1 && selected(mine[u], gmine[u])) sh_fail = 1;
    }
    // ---- the selected to the front (k of them: vk is finite whenever k <= the candidates there are), sorted
    if (tid == 0) sh_base = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < LKB_PER; ++u) {
        const bool s1 = selected(mine[u], gmine[u]);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(s1);
        if (lane == 0) sh_cnt[w] = __popcll(m);
        __syncthreads();
        int off = sh_base;
        for (int ww = 0; ww < w; ++ww) off += sh_cnt[ww];
        if (s1) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            kd[pos] = __longlong_as_double((long long)mine[u]);
            ki[pos] = gmine[u];
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int ww = 0; ww < LKB_T / 64; ++ww) t += sh_cnt[ww];
            sh_base += t;
        }
        __syncthreads();
    }
    for (int i = k + tid; i < np2; i += LKB_T) {
        kd[i] = __builtin_inf();
        ki[i] = 0x7fffffff;
    }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += LKB_T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const double a = kd[i], b = kd[ixj];
                    const int ia = ki[i], ib = ki[ixj];
                    const bool gt = a > b || (a == b && ia > ib);
                    if (((i & kk) == 0) ? gt : !gt) {
                        kd[i] = b;
                        kd[ixj] = a;
                        ki[i] = ib;
                        ki[ixj] = ia;
                    }
                }
            }
            __syncthreads();
        }
    for (int r = tid; r < k; r += LKB_T) {
        idx_out[(int64_t)q * k + r] = ki[r];
        if (dist_out) dist_out[(int64_t)q * k + r] = sqrt(kd[r]);
    }
    if (tid == 0) {
        if (kth_out) kth_out[q] = sh_fail ? __builtin_inf() : sqrt(kd[k - 1]);
        if (sh_fail) {
            if (opt) atomicOr(opt, 1);
            else flagged[1 + atomicAdd(&flagged[0], 1)] = q;
        }
    }
}

// optimistic run: the queries a merge could not certify are on the device (flagged[0] of them).  Up to LK_OPT_CAP take the full
// FP64 scan there and then (launched whatever the count: with none flagged its workgroups end at once); more raise the run's
// flag (opt[0]) and the engine repeats the run with host-checked searches.  opt[1] counts the queries that took an exact path.
constexpr int LK_OPT_CAP = 32;
__global__ void lk_opt_overflow(const int32_t* __restrict__ flagged, int32_t* __restrict__ opt) {
    const int c = flagged[0];
    if (c > LK_OPT_CAP) atomicOr(&opt[0], 1);
    if (c > 0) atomicAdd(&opt[1], c < LK_OPT_CAP ? c : LK_OPT_CAP);
}

// false: the shape does not suit the partitioned search (too few reference cells a partition, too many candidates)
bool large_k_search(hipStream_t stream, KnnWorkspace& ws, const double* X, const int32_t* ref_rows, int nr, const double* Qs,
                    const int32_t* qrs, int nq, int d, int k, int32_t* io, double* dout, const double* centre,
                    double* kth_out) {
    // neighbours asked of a partition: 36 where a tier holds that many (rows of up to 61 columns), else 20 (up to 125); a
    // partition holds 0.45 of that of a query's k on average, so that one holding all of it is rare
    Tier tiers[2];
    int kp = 36;
    if (candidate_tiers(d, kp, std::max(nr / cdiv(k, 16), 1), tiers) == 0) kp = 20;
    int P = std::max(2, cdiv(k, kp == 36 ? 16 : 9));
    // beyond what the rank-counting merges take (2 048 candidates): the big merge, with partitions dealt a little finer -- 14
    // (8) of a query's k each on average -- so that a partition holding all kp of them is rarer still (one such query would
    // send an optimistic engine run back to its start)
    if ((int64_t)P * kp > LK_MAXE) P = cdiv(k, kp == 36 ? 14 : 8);
    if ((int64_t)P * kp > LKB_MAXE || k > LKB_MAXK || nr / P < 4 * kp) return false;
    if (candidate_tiers(d, kp, nr / P, tiers) == 0) return false;
    int32_t* rows = ws.lk_rows.reserve((size_t)nr);
    int32_t* sub = ws.lk_idx.reserve((size_t)P * nq * kp);
    hipLaunchKernelGGL(lk_partition_rows, dim3(cdiv(nr, 256)), dim3(256), 0, stream, ref_rows, nr, P, rows);
    BMX_LAUNCH_CHECK();
    const int base = nr / P, rem = nr % P;
    // The partitions are strided samples of one reference: a query's kp-th distance in the first is about its kp-th distance in
    // every other.  So the first is searched plainly and its (certified) kp-th distance seeds the rest: their thresholds start
    // there instead of at +inf -- a partition of ~1 400 cells is otherwise one long warm-up, every early value a survivor for the
    // service waves (1.34 ms a launch at k = 1 000 against 13 us of matrix work).  A seeded list is complete up to the seed only;
    // the merges' certificate asks the k-th merged distance to lie inside it.
    // The partitions' searches leave their exact SQUARED distances (the reference's sum, left to right -- what the merges used to
    // gather every candidate's row for: 2 592 rows of 400 bytes a query at k = 1 000, 49 ms of a 100 ms search).
    const size_t n_sub = (size_t)P * nq * kp;
    double* sub_d2 = n_sub * sizeof(double) <= ((size_t)16 << 30) ? ws.lk_d2.reserve(n_sub) : nullptr;
    struct SquaredScope {
        KnnWorkspace& w;
        ~SquaredScope() { w.dist_squared = false; }
    } squared_scope{ws};
    ws.dist_squared = sub_d2 != nullptr;
    const bool seeded = dev_knobs().lk_seed != 0 && P >= 3;
    double* kth0 = seeded ? ws.lk_kth.reserve((size_t)nq) : nullptr;
    float* seed = seeded ? ws.lk_seed.reserve((size_t)nq) : nullptr;
    for (int p = 0; p < P; ++p) {
        const int n_p = base + (p < rem ? 1 : 0);
        const int nt = candidate_tiers(d, kp, n_p, tiers);
        search_tiers(stream, ws, tiers, nt, 0, X, rows + (int64_t)p * base + std::min(p, rem), n_p, Qs, qrs, nq, d, kp,
                     sub + (int64_t)p * nq * kp, sub_d2 ? sub_d2 + (int64_t)p * nq * kp : nullptr, p > 0 ? seed : nullptr, centre,
                     p == 0 ? kth0 : nullptr);
        if (p == 0 && seeded) {
            hipLaunchKernelGGL(lk_seed_kernel, dim3(cdiv(nq, 256)), dim3(256), 0, stream, (const double*)kth0, nq, seed);
            BMX_LAUNCH_CHECK();
        }
    }
    ws.dist_squared = false;  // (what follows writes the caller's distances)
    int32_t* flagged = ws.flagged_t[0].reserve((size_t)nq + 1);
    int32_t* opt = ws.optimistic ? ws.opt_state_ptr(stream) : nullptr;
    BMX_HIP(hipMemsetAsync(flagged, 0, sizeof(int32_t), stream));
    int32_t* const no_opt = nullptr;  // (the merges list what they cannot certify in either kind of run)
    if (P * kp > 1024) {  // (the bisection merge sorts the k selected, lk_merge all the candidates: measured equal at 684 candidates
                          // a query (k = 300), 88 against 100 ms a config-2 step at 1 152 (k = 500))
        size_t np2 = 1;
        while (np2 < (size_t)k) np2 <<= 1;
        const size_t lds = np2 * 12;
        if (P * kp <= 256 * LKB_PER) {
            ensure_dynamic_lds(reinterpret_cast<const void*>(&lk_merge_big<256>), lds);
            hipLaunchKernelGGL(lk_merge_big<256>, dim3(nq), dim3(256), lds, stream, X, (const int32_t*)rows, nr, P, kp, Qs, qrs, nq, d,
                               k, (const int32_t*)sub, io, dout, flagged, no_opt, kth_out, (const float*)seed, (const double*)sub_d2);
        } else if (P * kp <= 512 * LKB_PER) {
            ensure_dynamic_lds(reinterpret_cast<const void*>(&lk_merge_big<512>), lds);
            hipLaunchKernelGGL(lk_merge_big<512>, dim3(nq), dim3(512), lds, stream, X, (const int32_t*)rows, nr, P, kp, Qs, qrs, nq, d,
                               k, (const int32_t*)sub, io, dout, flagged, no_opt, kth_out, (const float*)seed, (const double*)sub_d2);
        } else {
            ensure_dynamic_lds(reinterpret_cast<const void*>(&lk_merge_big<1024>), lds);
            hipLaunchKernelGGL(lk_merge_big<1024>, dim3(nq), dim3(1024), lds, stream, X, (const int32_t*)rows, nr, P, kp, Qs, qrs, nq,
                               d, k, (const int32_t*)sub, io, dout, flagged, no_opt, kth_out, (const float*)seed,
                               (const double*)sub_d2);
        }
    } else if (P * kp <= 512)
        hipLaunchKernelGGL(lk_merge_wave, dim3(cdiv(nq, 4)), dim3(256), 0, stream, X, (const int32_t*)rows, nr, P, kp, Qs, qrs, nq,
                           d, k, (const int32_t*)sub, io, dout, flagged, no_opt, kth_out, (const float*)seed, (const double*)sub_d2);
    else
        hipLaunchKernelGGL(lk_merge, dim3(nq), dim3(256), 0, stream, X, (const int32_t*)rows, nr, P, kp, Qs, qrs, nq, d, k,
                           (const int32_t*)sub, io, dout, flagged, no_opt, kth_out, (const float*)seed, (const double*)sub_d2);
    BMX_LAUNCH_CHECK();
    if (!opt) {
        const int count = read_count(stream, ws, flagged);
        if (count > 0) exact_search(stream, ws, X, ref_rows, nr, Qs, qrs, d, k, io, dout, flagged, nullptr, count);
    } else {
        // (one query whose first partition holds 36 of its k used to send the whole run back to its start: 7 of 300 000 queries a
        // step did at k = 1 000, every step ran twice)
        hipLaunchKernelGGL(lk_opt_overflow, dim3(1), dim3(1), 0, stream, (const int32_t*)flagged, opt);
        BMX_LAUNCH_CHECK();
        double* drow = ws.drow.reserve((size_t)LK_OPT_CAP * nr);
        hipLaunchKernelGGL(knn_exact_dist, dim3(cdiv(nr, XDT), cdiv(LK_OPT_CAP, XDT)), dim3(256), 0, stream, X, ref_rows, nr, Qs,
                           qrs, d, (const int32_t*)flagged, 0, 0, drow, LK_OPT_CAP);
        BMX_LAUNCH_CHECK();
        if (k > 256 && k <= XSB_MAXK) {
            int np2 = 128;
            while (np2 < k) np2 <<= 1;
            const size_t lds = (size_t)np2 * (sizeof(double) + sizeof(int));
            ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_exact_select_big), lds);
            hipLaunchKernelGGL(knn_exact_select_big, dim3(LK_OPT_CAP), dim3(XSB_T), lds, stream, (const double*)drow, nr, k, np2,
                               (const int32_t*)flagged, 0, io, dout, LK_OPT_CAP);
        } else {
            hipLaunchKernelGGL(knn_exact_select, dim3(LK_OPT_CAP), dim3(256), 0, stream, (const double*)drow, nr, k,
                               (const int32_t*)flagged, 0, io, dout, LK_OPT_CAP);
        }
        BMX_LAUNCH_CHECK();
    }
    return true;
}

}  // namespace

void knn_device(hipStream_t stream, KnnWorkspace& ws, const double* X, const int32_t* ref_rows, int nr,
                const double* Q, const int32_t* q_rows, int nq_total, int d, int k, int32_t* idx_out,
                double* dist_out, int q_begin, int q_end, const float* seed_d2, const double* centre, double* kth_out) {
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

    Tier tiers[2];
    const int ntiers = ws.force_exact ? 0 : candidate_tiers(d, k, nr, tiers);
    ws.last_exact = 0;
    ws.last_flagged_tier[0] = ws.last_flagged_tier[1] = 0;
    if (kth_out && ntiers == 0) {  // no candidate tier takes the shape: every row read by the probe
        hipLaunchKernelGGL(fill_f64, dim3(cdiv(nq, 256)), dim3(256), 0, stream, kth_out + q_begin, nq, __builtin_inf());
        BMX_LAUNCH_CHECK();
    }
    // (k beyond the tiers' lists: P partitions of the reference at k = 36 each, merged exactly; a seeded search goes unseeded --
    // the full k nearest serve its caller just as well)
    // (also k in (20, 36] where no tier holds lists that long: rows of more than 61 columns, BASELINE config 5's 100 PCs)
    if (ntiers == 0 && !ws.force_exact && k > 20 && dev_knobs().knn_tier != 3 &&
        large_k_search(stream, ws, X, ref_rows, nr, Qs, qrs, nq, d, k, io, dout, centre, kth_out ? kth_out + q_begin : nullptr)) {
        ws.exact_total += ws.last_exact;
        return;
    }
    search_tiers(stream, ws, tiers, ntiers, 0, X, ref_rows, nr, Qs, qrs, nq, d, k, io, dout,
                 seed_d2 ? seed_d2 + q_begin : nullptr, centre, kth_out ? kth_out + q_begin : nullptr);
    if (ntiers > 0) ws.exact_total += ws.last_exact;
    if (ntiers > 0) ws.tier2_total += ws.last_flagged_tier[0] * (ntiers > 1 ? 1 : 0);
}

}  // namespace bmx
