// HIP versions of the two classic-mnnCorrect natives that stay registered in batchelor's .Call table:
//   smooth_gaussian_kernel (src/smooth_gaussian_kernel.cpp:11-118)  and
//   adjust_shift_variance  (src/adjust_shift_variance.cpp:30-164).
// FP64 throughout.
// smooth_gaussian_kernel: the reference's sequential log-space accumulations become max-shifted (online log-sum-exp)
// reductions; results agree to rounding.
// adjust_shift_variance comes in two forms.  Its quantile walk (sorted cumulative log-sum against a target that is
// itself a log-sum) decides on last-bit differences whenever the weights are concentrated -- in a few per cent of the
// cells of the reference's own test data -- so a form that sums in any other order picks a different CELL there, not a
// rounded value.  asv_exact_kernel therefore repeats the reference's order of operations literally (distances in
// parallel, then the sequential logspace_add chains in restrict order, a lexicographic sort of (projection, weight),
// the sequential walk), with the bit-reproducible exp / log1p of portable_math.hpp: bit for bit what a CPU
// following the same order of operations with the same arithmetic gets (the tests' checker does).  Its sequential chains cost O(cells x (nr1 + nr2)) dependent steps, fine up to ~1e5 restricted cells;
// beyond that asv_kernel (parallel sums + sort-free weighted-quantile bisection) takes over and agrees except on those
// ill-conditioned cells (tests/testthat/test-mnn-correct.R:141,396-399 acknowledge the effect upstream).
#include "bmx_ops.hpp"
#include "portable_math.hpp"

#include <algorithm>
#include <cstdlib>

namespace bmx {
namespace {

constexpr int T = 256;

__device__ __forceinline__ double block_max(double v, double* sm) {
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int o = T / 2; o > 0; o >>= 1) {
        if (tid < o) sm[tid] = fmax(sm[tid], sm[tid + o]);
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_sum(double v, double* sm) {
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int o = T / 2; o > 0; o >>= 1) {
        if (tid < o) sm[tid] += sm[tid + o];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------------
// smooth_gaussian_kernel (src/smooth_gaussian_kernel.cpp:11-118) as ONE streaming kernel per output tile, in the
// manner of a flash-attention forward pass:
//     out[:, c] = sum_i averaged[:, i] p_ic / sum_i p_ic,   p_ic = exp(-|m_i - x_c|^2 / sigma2 - density_i)
// with m_i = mat[:, index_i] the MNN-involved cells.  A workgroup owns 64 cells x 128 genes of `out`, walks the MNN
// cells in tiles of 64 and keeps a running maximum / sum per cell (online softmax), so every (MNN cell, cell)
// distance is formed once per gene tile and no weight ever touches memory.
// Both products run on the FP64 matrix cores (v_mfma_f64_16x16x4_f64):
//   scores  S[i][c] = m_i . x_c over the distance genes (A = MNN tile, B = cell tile; |m|^2 + |x|^2 - 2 S after);
//   output  out[c][g] += P[c][i] averaged[i][g] -- the score accumulator IS this product's A operand: register reg of
//           row tile t at lane l holds i = 16 t + 4 reg + (l >> 4), exactly k-step 4 t + reg's element (l >> 4), and its
//           column c = l & 15 is the row the A operand wants, so P goes from the first product into the second without
//           leaving the registers.
// density_i = log sum_j exp(-|m_i - m_j|^2 / sigma2) is the same kernel without the second product (its log-sum-exp
// per "cell" over the MNN cells, the cells being the MNN cells themselves).
// ---------------------------------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void fill_nan(double* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __builtin_nan("");
}

__global__ void row_norms2(const double* __restrict__ X, int64_t n, int gd, double* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n) return;
    double s = 0.0;
    for (int k = lane; k < gd; k += 64) s += X[r * gd + k] * X[r * gd + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[r] = s;
}

constexpr int SG_KC = 32;    // distance genes staged per step
constexpr int SG_GT = 128;   // output genes per workgroup

template <bool APPLY>
__global__ __launch_bounds__(256) void sgk_flash(const double* __restrict__ X, int gd, const double* __restrict__ xn2,
                                                 const int32_t* __restrict__ cells, int64_t ncells,
                                                 const int32_t* __restrict__ index, int U, double inv_s2,
                                                 const double* __restrict__ dens, const double* __restrict__ Av, int g,
                                                 double* __restrict__ out, double* __restrict__ lse) {
    constexpr int P = SG_KC + 2;        // pitch 34: conflict-free fragment reads (see pca.hip)
    constexpr int PA = SG_GT + 16;      // pitch 144 = 16 mod 32
    __shared__ double ms[64 * P];
    __shared__ double xs[64 * P];
    __shared__ double as_[APPLY ? 64 * PA : 1];
    __shared__ double mi2[64], di[64], fac[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int g0 = blockIdx.y * SG_GT;
    const int cl = 16 * w + (lane & 15);  // this lane's column (cell) of the score tile
    const int64_t cme = c0 + cl;
    const int64_t crow = cme < ncells ? (cells ? cells[cme] : cme) : -1;
    const double cn2 = crow >= 0 ? xn2[crow] : 0.0;
    double m_run = -__builtin_inf(), l_run = 0.0;  // l_run: this lane's share of the column sum
    d4 acc[APPLY ? SG_GT / 16 : 1];
#pragma unroll
    for (int t = 0; t < (APPLY ? SG_GT / 16 : 1); ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    const int lr = tid >> 2, seg = (tid & 3) * 8;
    const int64_t xrow = c0 + lr < ncells ? (cells ? cells[c0 + lr] : c0 + lr) : -1;
    for (int i0 = 0; i0 < U; i0 += 64) {
        // ---- scores: S_t[reg] = m_i . x_c, i = i0 + 16 t + 4 reg + (lane >> 4), c = cl
        d4 S[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) S[t] = d4{0.0, 0.0, 0.0, 0.0};
        const int64_t mrow = i0 + lr < U ? index[i0 + lr] : -1;
        if (tid < 64) {
            const int64_t r = i0 + tid < U ? index[i0 + tid] : -1;
            mi2[tid] = r >= 0 ? xn2[r] : 0.0;
            di[tid] = (r >= 0 && dens) ? dens[i0 + tid] : 0.0;
        }
        for (int k0 = 0; k0 < gd; k0 += SG_KC) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + seg + e;
                ms[lr * P + seg + e] = (mrow >= 0 && k < gd) ? X[mrow * gd + k] : 0.0;
                xs[lr * P + seg + e] = (xrow >= 0 && k < gd) ? X[xrow * gd + k] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < SG_KC / 4; ++kk) {
                const double b = xs[cl * P + 4 * kk + (lane >> 4)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double a = ms[(16 * t + (lane & 15)) * P + 4 * kk + (lane >> 4)];
                    S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, S[t], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        if constexpr (APPLY) {
            // stage the averaged vectors of this MNN tile: as_[i][gene]
            for (int e = tid; e < 64 * SG_GT; e += 256) {
                const int ii = e / SG_GT, gg = e - ii * SG_GT;
                as_[ii * PA + gg] = (i0 + ii < U && g0 + gg < g) ? Av[(int64_t)(i0 + ii) * g + g0 + gg] : 0.0;
            }
        }
        // ---- log-weights and the online softmax of this lane's column
        double tmax = -__builtin_inf();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ii = 16 * t + 4 * reg + (lane >> 4);
                const double d2 = mi2[ii] + cn2 - 2.0 * S[t][reg];
                const double v = i0 + ii < U ? -(d2 > 0.0 ? d2 : 0.0) * inv_s2 - di[ii] : -__builtin_inf();
                S[t][reg] = v;
                tmax = fmax(tmax, v);
            }
        tmax = fmax(tmax, __shfl_xor(tmax, 16));
        tmax = fmax(tmax, __shfl_xor(tmax, 32));
        const double m_new = fmax(m_run, tmax);  // finite: the tile holds at least one MNN cell
        const double scale = exp(m_run - m_new);  // exp(-inf) = 0 on the first tile
        double psum = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double pv = exp(S[t][reg] - m_new);
                S[t][reg] = pv;
                psum += pv;
            }
        l_run = l_run * scale + psum;
        m_run = m_new;
        if constexpr (APPLY) {
            if ((lane >> 4) == 0) fac[cl] = scale;
            __syncthreads();  // fac and as_ visible
            // the output rows of this lane are the cells 16 w + (lane >> 4) + 4 reg: rescale, then accumulate
            double f[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) f[reg] = fac[16 * w + (lane >> 4) + 4 * reg];
#pragma unroll
            for (int gt = 0; gt < SG_GT / 16; ++gt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[gt][reg] *= f[reg];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int kk = 4 * t + reg;  // k-step: its element (lane >> 4) is MNN cell 4 kk + (lane >> 4) of the tile
                    const double a = S[t][reg];
#pragma unroll
                    for (int gt = 0; gt < SG_GT / 16; ++gt) {
                        const double b = as_[(4 * kk + (lane >> 4)) * PA + 16 * gt + (lane & 15)];
                        acc[gt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[gt], 0, 0, 0);
                    }
                }
        }
        __syncthreads();  // mi2 / di / as_ / fac are rewritten by the next tile
    }
    // column sum over the four lanes that share a column
    double l_tot = l_run + __shfl_xor(l_run, 16);
    l_tot += __shfl_xor(l_tot, 32);
    if constexpr (APPLY) {
        if ((lane >> 4) == 0) fac[cl] = 1.0 / l_tot;
        __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t c = c0 + 16 * w + (lane >> 4) + 4 * reg;
            if (c >= ncells) continue;
            const double inv = fac[16 * w + (lane >> 4) + 4 * reg];
#pragma unroll
            for (int gt = 0; gt < SG_GT / 16; ++gt) {
                const int gg = g0 + 16 * gt + (lane & 15);
                if (gg < g) out[c * g + gg] = acc[gt][reg] * inv;
            }
        }
    } else {
        if ((lane >> 4) == 0 && cme < ncells) lse[cme] = m_run + log(l_tot);
    }
}

// ---------------------------------------------------------------------------------------------------
// adjust_shift_variance: one workgroup per cell of data2 (grid-stride), scratch = (proj, weight) of restrict1
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long f64_orderable(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return u ^ ((u >> 63) ? ~0ull : 0x8000000000000000ull);
}
__device__ __forceinline__ double orderable_f64(unsigned long long o) {
    unsigned long long u = o ^ ((o >> 63) ? 0x8000000000000000ull : ~0ull);
    return __longlong_as_double((long long)u);
}

__global__ __launch_bounds__(T) void asv_kernel(const double* __restrict__ data1, int g, int n1,
                                                const double* __restrict__ data2, int n2,
                                                const double* __restrict__ vect, int64_t vs_cell, int64_t vs_x,
                                                double sigma2, const int32_t* __restrict__ r1, int nr1,
                                                const int32_t* __restrict__ r2, int nr2, double* __restrict__ out,
                                                double* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* grad = reinterpret_cast<double*>(smem_raw);  // [g]
    double* cur = grad + g;                              // [g]
    __shared__ double sm[T];
    __shared__ double sh_l2, sh_proj;
    const int tid = threadIdx.x;
    double* proj1 = scratch + (int64_t)blockIdx.x * 2 * nr1;
    double* lw1 = proj1 + nr1;

    for (int cell = blockIdx.x; cell < n2; cell += gridDim.x) {
        // unit gradient and own projection (adjust_shift_variance.cpp:57-70)
        for (int x = tid; x < g; x += T) {
            grad[x] = vect[(int64_t)x * vs_x + (int64_t)cell * vs_cell];
            cur[x] = data2[(int64_t)cell * g + x];
        }
        __syncthreads();
        if (tid == 0) {
            double l2 = 0.0;
            for (int x = 0; x < g; ++x) l2 += grad[x] * grad[x];
            sh_l2 = sqrt(l2);
        }
        __syncthreads();
        const double l2 = sh_l2;
        if (l2 != 0.0)
            for (int x = tid; x < g; x += T) grad[x] /= l2;
        __syncthreads();
        if (tid == 0) {
            double p = 0.0;
            for (int x = 0; x < g; ++x) p += grad[x] * cur[x];
            sh_proj = p;
        }
        __syncthreads();
        const double curproj = sh_proj;

        // own-batch cumulative probability (:74-112): two passes, max then sums
        auto pair_stats = [&](const double* other, double& proj, double& lw) {
            double pr = 0.0, sc = 0.0;
            for (int x = 0; x < g; ++x) {
                pr += grad[x] * other[x];
                sc += (cur[x] - other[x]) * grad[x];
            }
            double dist = 0.0;
            for (int x = 0; x < g; ++x) {
                const double w = (cur[x] - other[x]) - sc * grad[x];
                dist += w * w;
            }
            proj = pr;
            lw = -dist / sigma2;
        };
        double mx = -__builtin_inf();
        for (int s = tid; s < nr2; s += T) {
            const int same = r2[s];
            double pr, lw;
            if (same == cell)
                lw = 0.0;
            else
                pair_stats(data2 + (int64_t)same * g, pr, lw);
            mx = fmax(mx, lw);
        }
        mx = block_max(mx, sm);
        double below = 0.0, all = 0.0;
        for (int s = tid; s < nr2; s += T) {
            const int same = r2[s];
            double pr = 0.0, lw = 0.0;
            bool add = true;
            if (same != cell) {
                pair_stats(data2 + (int64_t)same * g, pr, lw);
                add = !(pr > curproj);
            }
            const double w = exp(lw - mx);
            all += w;
            if (add) below += w;
        }
        below = block_sum(below, sm);
        all = block_sum(all, sm);
        // prob2 (log) = log(below) - log(all); with nothing added the reference's prob2 stays 0 before the subtraction
        const double prob2 = (nr2 > 0 ? (below > 0.0 ? mx + log(below) : 0.0) - (mx + log(all)) : 0.0);

        // reference batch: projections and log-weights (:115-135)
        double mx1 = -__builtin_inf();
        for (int o = tid; o < nr1; o += T) {
            double pr, lw;
            pair_stats(data1 + (int64_t)r1[o] * g, pr, lw);
            proj1[o] = pr;
            lw1[o] = lw;
            mx1 = fmax(mx1, lw);
        }
        mx1 = block_max(mx1, sm);
        double tot1 = 0.0;
        for (int o = tid; o < nr1; o += T) {
            const double w = exp(lw1[o] - mx1);
            lw1[o] = w;  // now a linear weight relative to the maximum
            tot1 += w;
        }
        tot1 = block_sum(tot1, sm);

        double ref_quan = __builtin_nan("");
        if (nr1 > 0) {
            // smallest projection whose cumulative weight reaches exp(prob2) * total (:138-157); found by bisection on
            // the order-preserving integer image of the projections (sort-free, deterministic reductions)
            const double target = exp(prob2) * tot1;
            unsigned long long lo = 0ull, hi = ~0ull;  // invariant: cum(<= hi) >= target or hi is the fallback
            // does any prefix reach the target at all?  (default: last element, :141)
            double mxp = -__builtin_inf();
            for (int o = tid; o < nr1; o += T) mxp = fmax(mxp, proj1[o]);
            mxp = block_max(mxp, sm);
            hi = f64_orderable(mxp);
            while (lo < hi) {
                const unsigned long long mid = lo + (hi - lo) / 2;
                double cum = 0.0;
                for (int o = tid; o < nr1; o += T)
                    if (f64_orderable(proj1[o]) <= mid) cum += lw1[o];
                cum = block_sum(cum, sm);
                if (cum >= target)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            // hi is now the smallest key with cum >= target (or the maximum); snap to the data value at / above it
            double best = __builtin_inf();
            for (int o = tid; o < nr1; o += T)
                if (f64_orderable(proj1[o]) >= hi) best = fmin(best, proj1[o]);
            best = -block_max(-best, sm);
            ref_quan = best;
        }
        if (tid == 0) out[cell] = (ref_quan - curproj) / l2;  // :160
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------
// adjust_shift_variance, literal order of operations (src/adjust_shift_variance.cpp:52-161), one workgroup per cell.
// scratch per workgroup: lw2 [nr2], add2 [nr2] (1.0 / 0.0), key [npad] x 2 (projection, log-weight; npad = nr1 rounded
// up to a power of two, padded with +inf so that the padding sorts last).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_less(double p0, double w0, double p1, double w1) {
    return p0 < p1 || (!(p1 < p0) && w0 < w1);  // std::pair<double, double> operator<
}

__global__ __launch_bounds__(T) void asv_exact_kernel(const double* __restrict__ data1, int g, const double* __restrict__ data2,
                                                      int n2, const double* __restrict__ vect, int64_t vs_cell,
                                                      int64_t vs_x, double sigma2, const int32_t* __restrict__ r1, int nr1,
                                                      const int32_t* __restrict__ r2, int nr2, int npad,
                                                      double* __restrict__ out, double* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* grad = reinterpret_cast<double*>(smem_raw);  // [g]
    double* cur = grad + g;                              // [g]
    __shared__ double sh_l2, sh_proj, sh_prob2, sh_tot2, sh_tot1;
    const int tid = threadIdx.x;
    double* lw2 = scratch + (int64_t)blockIdx.x * (2 * (int64_t)nr2 + 2 * (int64_t)npad);
    double* add2 = lw2 + nr2;
    double* kp = add2 + nr2;  // projections
    double* kw = kp + npad;   // log-weights

    for (int cell = blockIdx.x; cell < n2; cell += gridDim.x) {
        for (int x = tid; x < g; x += T) {
            grad[x] = vect[(int64_t)x * vs_x + (int64_t)cell * vs_cell];
            cur[x] = data2[(int64_t)cell * g + x];
        }
        __syncthreads();
        if (tid == 0) {  // :57-70
            double l2 = 0.0;
            for (int x = 0; x < g; ++x) l2 += grad[x] * grad[x];
            sh_l2 = sqrt(l2);
        }
        __syncthreads();
        const double l2 = sh_l2;
        if (l2 != 0.0)
            for (int x = tid; x < g; x += T) grad[x] /= l2;
        __syncthreads();
        if (tid == 0) {
            double p = 0.0;
            for (int x = 0; x < g; ++x) p += grad[x] * cur[x];
            sh_proj = p;
        }
        __syncthreads();
        const double curproj = sh_proj;
        auto pair_stats = [&](const double* other, double& proj, double& lw) {  // :9-27 + the projection
            double pr = 0.0, sc = 0.0;
            for (int x = 0; x < g; ++x) pr += grad[x] * other[x];
            for (int x = 0; x < g; ++x) sc += (cur[x] - other[x]) * grad[x];
            double dist = 0.0;
            for (int x = 0; x < g; ++x) {
                const double w = (cur[x] - other[x]) - sc * grad[x];
                dist += w * w;
            }
            proj = pr;
            lw = -dist / sigma2;
        };
        // every pair's projection and log-weight, in parallel (each value is computed exactly as the reference does)
        for (int s = tid; s < nr2; s += T) {
            const int same = r2[s];
            double pr = 0.0, lw = 0.0;
            bool add = true;
            if (same != cell) {
                pair_stats(data2 + (int64_t)same * g, pr, lw);
                add = !(pr > curproj);
            }
            lw2[s] = lw;
            add2[s] = add ? 1.0 : 0.0;
        }
        for (int o = tid; o < npad; o += T) {
            double pr = __builtin_inf(), lw = __builtin_inf();
            if (o < nr1) pair_stats(data1 + (int64_t)r1[o] * g, pr, lw);
            kp[o] = pr;
            kw[o] = lw;
        }
        __syncthreads();
        // the three sequential log-sum chains, in restrict order (:74-112, :117-131), one wave each
        if (tid == 0) {
            double prob2 = 0.0, tot2 = 0.0;
            bool first_p = true;
            for (int s = 0; s < nr2; ++s) {
                const double lp = lw2[s];
                if (add2[s] != 0.0) {
                    prob2 = first_p ? lp : bmx_pm_logspace_add(prob2, lp);
                    first_p = false;
                }
                tot2 = s == 0 ? lp : bmx_pm_logspace_add(tot2, lp);
            }
            sh_prob2 = prob2;
            sh_tot2 = tot2;
        } else if (tid == 64) {
            double tot1 = 0.0;
            for (int o = 0; o < nr1; ++o) tot1 = o == 0 ? kw[o] : bmx_pm_logspace_add(tot1, kw[o]);
            sh_tot1 = tot1;
        }
        __syncthreads();
        // std::sort of the (projection, log-weight) pairs (:134): bitonic network over npad slots
        for (int k = 2; k <= npad; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npad; i += T) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const double p0 = kp[i], w0 = kw[i], p1 = kp[ixj], w1 = kw[ixj];
                        const bool up = (i & k) == 0;
                        if (up ? pair_less(p1, w1, p0, w0) : pair_less(p0, w0, p1, w1)) {
                            kp[i] = p1;
                            kw[i] = w1;
                            kp[ixj] = p0;
                            kw[ixj] = w0;
                        }
                    }
                }
                __syncthreads();
            }
        if (tid == 0) {  // :137-160
            double ref_quan = __builtin_nan("");
            if (nr1 > 0) {
                const double target = (sh_prob2 - sh_tot2) + sh_tot1;
                double cum = 0.0;
                ref_quan = kp[nr1 - 1];
                for (int o = 0; o < nr1; ++o) {
                    cum = o == 0 ? kw[o] : bmx_pm_logspace_add(cum, kw[o]);
                    if (cum >= target) {
                        ref_quan = kp[o];
                        break;
                    }
                }
            }
            out[cell] = (ref_quan - curproj) / l2;
        }
        __syncthreads();
    }
}

}  // namespace

// ws: n + U doubles (squared norms of every cell over the distance genes, densities of the MNN cells)
void smooth_gaussian_kernel_device(hipStream_t stream, const double* averaged, int g, int U, const int32_t* index,
                                   const double* mat, int gd, int n, double sigma2, double* out, double* ws) {
    if (n <= 0 || g <= 0) return;
    if (U <= 0) {  // no MNN cell: 0 / 0 everywhere, as the reference's final division gives
        hipLaunchKernelGGL(fill_nan, dim3((unsigned)cdiv((int64_t)n * g, 256)), dim3(256), 0, stream, out, (int64_t)n * g);
        BMX_LAUNCH_CHECK();
        return;
    }
    double* xn2 = ws;
    double* dens = ws + n;
    hipLaunchKernelGGL(row_norms2, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream, mat, (int64_t)n, gd, xn2);
    // densities: the log-sum-exp of every MNN cell over the MNN cells (:56-65)
    hipLaunchKernelGGL(sgk_flash<false>, dim3((unsigned)cdiv(U, 64), 1), dim3(256), 0, stream, mat, gd, xn2, index,
                       (int64_t)U, index, U, 1.0 / sigma2, nullptr, nullptr, 0, nullptr, dens);
    hipLaunchKernelGGL(sgk_flash<true>, dim3((unsigned)cdiv(n, 64), (unsigned)cdiv(g, SG_GT)), dim3(256), 0, stream, mat, gd,
                       xn2, nullptr, (int64_t)n, index, U, 1.0 / sigma2, dens, averaged, g, out, nullptr);
    BMX_LAUNCH_CHECK();
}

size_t adjust_shift_variance_scratch(int n2, int nr1, int nr2, int* blocks, int* npad, int* exact) {
    static const int force_fast = std::getenv("BMX_ASV_FAST") != nullptr;  // developer switch
    *exact = !force_fast && (int64_t)nr1 + nr2 <= 131072;
    int p = 1;
    while (p < std::max(nr1, 1)) p <<= 1;
    *npad = p;
    const size_t per_block = *exact ? 2 * (size_t)nr2 + 2 * (size_t)p : 2 * (size_t)std::max(nr1, 1);
    const size_t budget = (size_t)1 << 27;  // doubles: 1 GiB of scratch at most
    *blocks = (int)std::max<size_t>(1, std::min<size_t>({(size_t)std::max(n2, 1), (size_t)1024, budget / std::max<size_t>(per_block, 1)}));
    return per_block * (size_t)*blocks;
}

void adjust_shift_variance_device(hipStream_t stream, const double* data1, int g, int n1, const double* data2, int n2,
                                  const double* vect, double sigma2, const int32_t* restrict1, int nr1,
                                  const int32_t* restrict2, int nr2, double* out, double* ws_pairs, int vect_row_major) {
    // vect is an R matrix [n2 x g] (column-major) at the .Call boundary, row-major [n2][g] inside the engine
    const int64_t vs_cell = vect_row_major ? g : 1, vs_x = vect_row_major ? 1 : n2;
    if (n2 <= 0) return;
    int blocks = 1, npad = 1, exact = 1;
    (void)adjust_shift_variance_scratch(n2, nr1, nr2, &blocks, &npad, &exact);
    if (exact)
        hipLaunchKernelGGL(asv_exact_kernel, dim3(blocks), dim3(T), (size_t)2 * g * sizeof(double), stream, data1, g, data2,
                           n2, vect, vs_cell, vs_x, sigma2, restrict1, nr1, restrict2, nr2, npad, out, ws_pairs);
    else
        hipLaunchKernelGGL(asv_kernel, dim3(blocks), dim3(T), (size_t)2 * g * sizeof(double), stream, data1, g, n1, data2,
                           n2, vect, vs_cell, vs_x, sigma2, restrict1, nr1, restrict2, nr2, out, ws_pairs);
    BMX_LAUNCH_CHECK();
}

}  // namespace bmx
