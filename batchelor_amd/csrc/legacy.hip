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
// following the same order of operations with the same arithmetic gets (the tests' checker does).  Its sequential chains
// cost O(cells x (nr1 + nr2)) dependent steps (1.6 ns per pair): it is taken up to 4e7 pairs per call; beyond that
// asv_tile_kernel (16-cell tiles on the FP64 matrix cores + a sort-free histogram quantile, 0.015 ns per pair) takes over and
// agrees except on those ill-conditioned cells (tests/testthat/test-mnn-correct.R:141,396-399 acknowledge the effect
// upstream).
#include "bmx_ops.hpp"
#include "portable_math.hpp"

#include <algorithm>
#include <type_traits>
#include <mutex>
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
typedef float f4 __attribute__((ext_vector_type(4)));

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

// exp(x) for x <= 0 in the stream's epilogue (the own batch's online log-sum-exp: one per pair, two when a new maximum arrives).
// The library's exp costs the one wave of a SIMD ~1 000 cycles a call there -- a block of own-batch rows took 2.6 times a block
// of reference rows (round 6, by phase ticks with and without the division: EXPERIMENTS.md).  Table-driven instead: x = (32 m + j)
// ln 2 / 32 + r, |r| <= ln 2 / 64, exp(x) = 2^m 2^(j / 32) (1 + r + r^2 / 2 + ... + r^6 / 720); the 32 powers sit in the LDS (tab).
// Below 2 ulp (checked against extended precision over [-700, 0]); -inf and anything below -800 give 0.
__device__ __forceinline__ double asv_exp_neg(double x, const double* tab) {
    x = fmax(x, -800.0);
    const double nf = __builtin_rint(x * 46.16624130844683);     // 32 / ln 2
    const int n = (int)nf;
    double r = __builtin_fma(nf, -0x1.62e42fee00000p-6, x);      // ln 2 / 32, high part (n times it is exact)
    r = __builtin_fma(nf, -0x1.a39ef35793c76p-38, r);            // ... low part
    double p = 1.0 / 720.0;
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    const double t = tab[n & 31];
    return ldexp(__builtin_fma(t, p, t), n >> 5);
}

constexpr int SG_KC = 32;    // distance genes staged per step
// SG_GT: output genes per workgroup -- 64 where that covers all of them (the PC-space calls of the merge engine's world: half the
// weighted-sum MFMAs of a 128-gene tile would multiply zeros, and two workgroups fit a CU: one's exp() under the other's
// MFMAs), 128 for gene-space calls (every gene tile repeats the distance products)

// NKB > 0: the distance genes fit NKB steps of 32 and the cells' operands of all of them stay in registers for the whole
// sweep over the MNN cells (staged once); NKB = 0: any number of genes, the cells' tile staged again with every MNN tile
template <bool APPLY, int SG_GT, int NKB>
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
    __shared__ double e2tab[32];  // 2^(j / 32) for asv_exp_neg: one FP64 exp per (MNN cell, cell) pair is what this kernel spends
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 32) e2tab[tid] = exp2((double)tid * 0.03125);  // (visible after the first barrier below, long before its first use)
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
    double bx[NKB > 0 ? NKB * 8 : 1];
    if constexpr (NKB > 0) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb * SG_KC + seg + e;
                xs[lr * P + seg + e] = (xrow >= 0 && k < gd) ? X[xrow * gd + k] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < SG_KC / 4; ++kk) bx[kb * 8 + kk] = xs[cl * P + 4 * kk + (lane >> 4)];
            __syncthreads();
        }
    }
    // Round 6 (NKB > 0): the NEXT tile's MNN rows, norms, densities and averaged vectors are on their way into registers while
    // this tile multiplies.  Before, a tile was four dependent round trips to the L2 (row ids, the rows of each 32-gene step,
    // the averaged vectors) with two workgroups per CU to hide them: 35 us a tile against 3.4 us of matrix work.
    constexpr int AVN = APPLY ? 64 * SG_GT / 256 : 1;  // averaged values a thread stages per tile
    double pm[NKB > 0 ? NKB * 8 : 1], pav[NKB > 0 ? AVN : 1], pmi = 0.0, pdi = 0.0;
    auto fetch_rows = [&](int i0n, int kb) __attribute__((always_inline)) {
        const int64_t mrow = i0n + lr < U ? index[i0n + lr] : -1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kb * SG_KC + seg + e;
            pm[kb * 8 + e] = (mrow >= 0 && k < gd) ? X[mrow * gd + k] : 0.0;
        }
    };
    auto fetch_rest = [&](int i0n) __attribute__((always_inline)) {
        if (tid < 64) {
            const int64_t r = i0n + tid < U ? index[i0n + tid] : -1;
            pmi = r >= 0 ? xn2[r] : 0.0;
            pdi = (r >= 0 && dens) ? dens[i0n + tid] : 0.0;
        }
        if constexpr (APPLY) {
#pragma unroll
            for (int j = 0; j < AVN; ++j) {
                const int e = tid + j * 256, ii = e / SG_GT, gg = e - ii * SG_GT;
                pav[j] = (i0n + ii < U && g0 + gg < g) ? Av[(int64_t)(i0n + ii) * g + g0 + gg] : 0.0;
            }
        }
    };
    if constexpr (NKB > 0) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) fetch_rows(0, kb);
        fetch_rest(0);
    }
    for (int i0 = 0; i0 < U; i0 += 64) {
        // ---- scores: S_t[reg] = m_i . x_c, i = i0 + 16 t + 4 reg + (lane >> 4), c = cl
        d4 S[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) S[t] = d4{0.0, 0.0, 0.0, 0.0};
        const int64_t mrow = (NKB > 0) ? -1 : (i0 + lr < U ? index[i0 + lr] : -1);
        if constexpr (NKB > 0) {
            if (tid < 64) {
                mi2[tid] = pmi;
                di[tid] = pdi;
            }
            if constexpr (APPLY) {
#pragma unroll
                for (int j = 0; j < AVN; ++j) {
                    const int e = tid + j * 256, ii = e / SG_GT, gg = e - ii * SG_GT;
                    as_[ii * PA + gg] = pav[j];
                }
            }
        } else if (tid < 64) {
            const int64_t r = i0 + tid < U ? index[i0 + tid] : -1;
            mi2[tid] = r >= 0 ? xn2[r] : 0.0;
            di[tid] = (r >= 0 && dens) ? dens[i0 + tid] : 0.0;
        }
        if constexpr (NKB > 0) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ms[lr * P + seg + e] = pm[kb * 8 + e];
                __syncthreads();
                // (this step's registers are free: the next tile's rows of the same step are asked for; the last tile asks for
                // nothing -- rows beyond U read as zeros without a load)
                fetch_rows(i0 + 64, kb);
                if (kb == NKB - 1) fetch_rest(i0 + 64);
#pragma unroll
                for (int kk = 0; kk < SG_KC / 4; ++kk) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double a = ms[(16 * t + (lane & 15)) * P + 4 * kk + (lane >> 4)];
                        S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bx[kb * 8 + kk], S[t], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        } else {
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
        }
        if constexpr (APPLY && NKB == 0) {
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
        const double scale = asv_exp_neg(m_run - m_new, e2tab);  // exp(-inf) = 0 on the first tile
        double psum = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double pv = asv_exp_neg(S[t][reg] - m_new, e2tab);
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
                                                      double* __restrict__ out, double* __restrict__ scratch,
                                                      int cell_begin, int cell_end) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* grad = reinterpret_cast<double*>(smem_raw);  // [g]
    double* cur = grad + g;                              // [g]
    __shared__ double sh_l2, sh_proj, sh_prob2, sh_tot2, sh_tot1;
    const int tid = threadIdx.x;
    double* lw2 = scratch + (int64_t)blockIdx.x * (2 * (int64_t)nr2 + 2 * (int64_t)npad);
    double* add2 = lw2 + nr2;
    double* kp = add2 + nr2;  // projections
    double* kw = kp + npad;   // log-weights

    for (int cell = cell_begin + blockIdx.x; cell < cell_end; cell += gridDim.x) {
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

// ---------------------------------------------------------------------------------------------------
// adjust_shift_variance at scale (BASELINE.json configs[4]: every right cell of a merge against every restricted cell of
// both batches -- 5e11 pairs at the root of the 16-batch tree).  A workgroup owns a TILE OF 16 CELLS:
//   1. the restricted cells of both batches, gathered into one zero-padded matrix (asv_gather_stream), stream past it once:
//      every wave loads its 16 rows of a 64-row step straight into MFMA operand registers (LDS-staged beyond 128
//      dimensions); two FP64 MFMA chains (v_mfma_f64_16x16x4_f64) per 16 x 16 sub-tile give  D = x_c . x_o  and
//      P = g^_c . x_o, from which the projection on the cell's line (P) and the squared distance to it
//      |x_c|^2 + |x_o|^2 - 2 D - (g^_c . x_c - P)^2  follow per pair (src/adjust_shift_variance.cpp:9-27 in GEMM form);
//      projection and log-weight go to the tile's scratch, the per-cell maxima / projection range come out of the same pass;
//   2. cell by cell: the own-batch probability (log-sum-exp, :74-112) and the weighted quantile of the reference batch's
//      projections (:117-157) WITHOUT sorting them: linear histogram of the weights over the projection range (2 048
//      bins, integer fixed-point sums: order-independent, so runs are bit-identical), the bin where the cumulative weight
//      crosses the target is collected, sorted and walked; a bin that is still too full is subdivided again.
// Weights are taken relative to the cell's largest one and kept to 2^-40: a reference cell 28 sigma2 further from the line
// than the nearest contributes nothing, as in FP64 it would not either beyond 37.  Cells whose walk is decided on the last
// bits may pick the neighbouring quantile (asv_exact_kernel is the bit-exact form, taken up to 4e7 pairs).
// ---------------------------------------------------------------------------------------------------
constexpr int AT_C = 16;        // cells per tile
constexpr int AT_R = 64;        // streamed cells per step
constexpr int AT_KC = 32;       // dimensions staged per step
constexpr int AT_NB = 2048;     // histogram bins
constexpr int AT_CAP = 2048;    // collected entries of the crossing bin
constexpr int AT_U = 8;         // scratch elements per thread and batch in the per-cell passes (two batches in flight)

constexpr int AT_NP = 2 * AT_R; // the stream is padded to whole pairs of steps (zero rows)
constexpr int AT_SM = 4 * AT_C * 6;  // doubles of the block-reduction area: six values per (wave, cell) after the stream (>= T)

__host__ __device__ inline int asv_tile_gp(int g) { return (g + AT_KC - 1) / AT_KC * AT_KC + 2; }
// 8-dimension blocks of a streamed row; 0: the staged form.  Round 6: a row carries two more columns behind its g coordinates
// -- 1 and its squared norm -- so that the distance chain's MFMAs deliver -|x_c - x_o|^2 / sigma directly (asv_tile_kernel)
__host__ __device__ inline int asv_tile_nb8(int g) { return g + 2 <= 128 ? (g + 2 + 7) / 8 : 0; }
// row stride of the gathered stream: whole 8-dimension blocks (zero filled) for the register-streamed form
inline int asv_tile_gs(int g) { return asv_tile_nb8(g) > 0 ? asv_tile_nb8(g) * 8 : g; }
__host__ __device__ inline size_t asv_tile_npad(size_t N) {
    const size_t p = (N + AT_NP - 1) / AT_NP * AT_NP;
    return p > (size_t)AT_NP ? p : (size_t)AT_NP;
}
// addends a chain of the literal re-run may keep (its sort buffer and two 16 KB bin arrays must fit the LDS beside the tile)
// (a power of two: the lists are padded to one for the sorting networks)
// (BASELINE config 5 at full size, sigma 0.1: 7 % of the root merge's cells keep more than 32 768 addends in a chain, 33 cells of
// the whole tree more than 65 536, none more than 131 072 -- at 4 % of the step's time for the longer lists)
inline int asv_tile_lcap_default(int) { return 131072; }
// ... of which this many are sorted in the LDS (beside the tile); longer lists are sorted where they lie, in global memory
__host__ __device__ inline int asv_tile_lsort(int g) { return g <= 128 ? 4096 : 2048; }
// doubles of the kernel's multi-purpose LDS region (see there)
__host__ __device__ inline int asv_tile_ub_doubles(int g, int lcap) {
    const int nb8 = asv_tile_nb8(g);
    int u = 2 * AT_CAP;
    const int ls = lcap < asv_tile_lsort(g) ? lcap : asv_tile_lsort(g);
    u = u > 2 * ls ? u : 2 * ls;
    const int a = nb8 > 8 ? 2 * nb8 * 2 * 64 : 0;
    return u > a ? u : a;
}
// doubles of a workgroup's lists: per cell (log-weight, flag) of the own batch, log-weight of the reference in restrict order,
// (projection, log-weight) of the reference sorted; three index lists
__host__ __device__ inline int64_t asv_tile_list_doubles(int lcap) { return (int64_t)AT_C * lcap * 5 + (3 * (int64_t)lcap + 1) / 2 + 1; }
inline size_t asv_tile_lds_bytes(int g, int lcap) {
    const int nb8 = asv_tile_nb8(g);
    return ((size_t)2 * AT_C * asv_tile_gp(g) + (size_t)asv_tile_ub_doubles(g, lcap) +
            (nb8 == 0 ? (size_t)AT_R * (AT_KC + 2) : 0) + 8 * AT_C + 32 + AT_SM) * sizeof(double) +
           (size_t)2 * AT_NB * sizeof(unsigned long long);
}

// The weighted-quantile walk (src/adjust_shift_variance.cpp:137-157: the first entry at which the cumulative weight reaches
// the target) over up to 8 T integer weights in the LDS, by the whole block instead of one thread going entry by entry:
// every thread sums its eight consecutive entries, a scan over the threads gives each its starting weight, the thread whose
// stretch holds the first crossing reports it.  Integer sums: the result is what the sequential walk finds.
// target_of(total weight of v) -> target.  Returns the index (-1: never reached), *cum_before = base + weight before it.
template <class TargetOf>
__device__ __forceinline__ int asv_first_crossing(const unsigned long long* v, int n, unsigned long long base,
                                                  TargetOf target_of, unsigned long long* smu, int* sh_idx,
                                                  unsigned long long* sh_cum, double* target_out) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i0 = tid * 8;
    unsigned long long mine[8], loc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mine[k] = i0 + k < n ? v[i0 + k] : 0ull;
        loc += mine[k];
    }
    unsigned long long inc = loc;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();  // (smu may still be read as something else)
    if (lane == 63) smu[w] = inc;
    if (tid == 0) *sh_idx = 0x7fffffff;
    __syncthreads();
    unsigned long long woff = 0, total = 0;
    for (int ww = 0; ww < T / 64; ++ww) {
        const unsigned long long t = smu[ww];
        total += t;
        if (ww < w) woff += t;
    }
    const double target = target_of(total);
    unsigned long long cum = base + woff + (inc - loc), cbefore = 0;
    int cand = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool hit = cand == 0x7fffffff && i0 + k < n && (double)(cum + mine[k]) >= target;
        cbefore = hit ? cum : cbefore;
        cand = hit ? i0 + k : cand;
        cum += mine[k];
    }
    if (cand != 0x7fffffff) atomicMin(sh_idx, cand);
    __syncthreads();
    if (cand == *sh_idx && cand != 0x7fffffff) *sh_cum = cbefore;
    __syncthreads();
    *target_out = target;
    return *sh_idx == 0x7fffffff ? -1 : *sh_idx;
}

// One cell's scratch row, elements [jstart, jstart + n) of the stream, past the block: thread t visits jstart + t,
// + T, ... in batches of AT_U, the next batch on its way while the current one is consumed (one wave per SIMD: nothing
// else hides the round trip to the scratch).  Straight-line code: every load is issued (block numbers clamped to the
// row's last block) and `proc` gets the element number to tell the ones beyond n -- with the loads or their first use
// under per-element branches the compiler waited for ALL outstanding loads before the first use: no overlap at all.
// So `proc` should select, not branch, on what it was given.
// Layout (see asv_tile_kernel): element j of cell slot c sits at (j >> 6) * 1024 + ((c + (j >> 6) + rot) & 15) * 64 + (j & 63).
template <bool WITH_W, class Proc>
__device__ __forceinline__ void asv_row_scan(const double* __restrict__ P, const double* __restrict__ W, int64_t jstart,
                                             int n, int crot, int last_block, int tid, Proc proc) {
    const int64_t jt = jstart + tid;
    const int jlow = (int)(jt & 63), jb_t = (int)(jt >> 6);
    double pa[AT_U], wa[AT_U], pb[AT_U], wb[AT_U];
    auto fetch = [&](double (&p)[AT_U], double (&w)[AT_U], int o0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < AT_U; ++u) {
            const int o = o0 + u * T;  // (o - tid is a multiple of T = 4 blocks)
            int jb = jb_t + ((o - tid) >> 6);
            jb = jb < last_block ? jb : last_block;
            const int64_t off = (int64_t)jb * (AT_C * 64) + (((crot + jb) & (AT_C - 1)) << 6) + jlow;
            p[u] = P[off];
            w[u] = WITH_W ? W[off] : 0.0;
        }
    };
    auto consume = [&](const double (&p)[AT_U], const double (&w)[AT_U], int o0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < AT_U; ++u) proc(p[u], w[u], o0 + u * T);
    };
    fetch(pa, wa, tid);
    for (int o0 = tid; o0 < n; o0 += 2 * AT_U * T) {
        fetch(pb, wb, o0 + AT_U * T);
        consume(pa, wa, o0);
        fetch(pa, wa, o0 + 2 * AT_U * T);
        consume(pb, wb, o0 + AT_U * T);
    }
}

// ---------------------------------------------------------------------------------------------------
// The literal re-run of the cells whose quantile walk is decided by rounding (the "flagged" cells of asv_tile_kernel).
//
// The reference sums its weights by sequential R::logspace_add chains (src/adjust_shift_variance.cpp:96-109, :127-131,
// :147-151).  Where the bandwidth is small against the squared distances, a cell's own log-weight (0) dwarfs the rest of
// its batch, prob2 - totalprob2 is below an ulp of totalprob1, the target of the walk IS totalprob1 (or an ulp or two
// below it) and the walk ends where the cumulative chain over the SORTED reference cells first reaches what the chain in
// RESTRICT order ended at: a statement about the rounding of two summation orders, which only the chains themselves
// reproduce.  But in exactly that regime almost every addend of a chain is an exact no-op:
//
//   Lemma.  Let acc be a chain's running value, known to lie in [L, U] with L, U of one sign, and 2^e <= min(|L|, |U|).
//   An addend lw < L + (e - 54) ln 2 - 1/4 leaves it unchanged: logspace_add(acc, lw) = acc + log1p(exp(lw - acc)), the
//   second term is <= exp(lw - L) < 2^(e - 54) e^(-1/4), less than a quarter of the spacing of doubles at |acc| >= 2^e (half
//   of it just below a power of two, towards zero) -- the sum rounds to acc; the few-ulp errors of the portable exp / log1p
//   are inside the factor e^(-1/4).  An addend lw <= acc - 700 is a no-op outright: the portable exp returns 0 there.
//
// acc is bounded from what has gone by: it is >= the largest addend so far (M) and <= M + log(count); once the cell's own
// weight 1 has gone by it is log(1 + sum of the others) in [log1p(e^Mx), log1p(count e^Mx)], Mx the largest other addend.
// A chain restricted to the addends the lemma does not exclude therefore has, by induction over its steps, the SAME bits
// as the full chain -- and keeping more addends than necessary (bounds from an earlier point of the sequence, margins for
// the matrix-core form's rounding of lw) is harmless: every addend that is left out is individually a no-op.
// So, per flagged cell: (i) the kept addends of the three chains are picked from the tile's scratch (values in GEMM form,
// margins `mb` / `tolp` for their rounding) -- in restrict order for totalprob2 / prob2 (:96-109) and totalprob1 (:127-131),
// in projection order for the walk (:147-151), where "what has gone by" is taken per histogram bin, two bins back; (ii)
// for those few the projection and the distance to the line are recomputed in the reference's own order of operations
// (:9-27, :88-89, :120-123), sorted as std::sort sorts the pairs (:134); (iii) one lane per chain repeats the reference's
// sequential sums with the bit-reproducible exp / log1p.  Every such cell is bit-equal to what a CPU following the
// reference's order of operations gets -- whatever the size of the call.  A cell that keeps more than `lcap` addends in
// any chain (bandwidths of the order of the squared distances: nearly every pair carries weight) goes the histogram
// way, which there resolves the quantile to 2^-40 of the total weight.
// ---------------------------------------------------------------------------------------------------
constexpr double AT_LN2 = 0.6931471805599453;


// addends below the returned value cannot change a chain whose running value lies in [L, U] (the lemma above)
__device__ __forceinline__ double asv_noop_below(double L, double U) {
    double t = L - 701.0;  // (portable exp: 0 for arguments <= -700)
    if ((L > 0.0 && U > 0.0) || (L < 0.0 && U < 0.0)) {
        const double amin = fmin(fabs(L), fabs(U));
        t = fmax(t, L + (double)(ilogb(amin) - 54) * AT_LN2 - 0.25);
    }
    return t;
}

// Thresholds of a chain from what has gone by (addends below them are no-ops), for the two regimes of a chain:
//  * running value below zero (a reference chain; an own-batch chain until the cell's own weight 1 arrives): it is >= the
//    largest addend so far (lo) and never exceeds hi + log(count), hi >= every addend the chain can have met by then;
//  * the cell itself has gone by: the chain stands at log(1 + the others' weights) >= log1p(exp(largest other so far)), and
//    only grows -- away from zero, towards coarser spacing.
// mb >= the rounding of the GEMM-form log-weights the bounds are taken from.
__device__ __forceinline__ double asv_thr_neg(double lo, double hi, double cnt, double mb) {
    if (lo == -__builtin_inf()) return lo;  // nothing has gone by: the next addend starts the chain
    return asv_noop_below(lo - mb, fmax(lo, hi) + mb + log(cnt) + 0.01) - mb;
}
__device__ __forceinline__ double asv_thr_self(double others, double mb) {
    // (no other addend yet: the chain stands at 0 exactly, only exp's underflow makes a no-op)
    const double L = others == -__builtin_inf() ? 0.0 : log1p(exp(others - mb)) * (1.0 - 1e-9);
    return asv_noop_below(L, L) - mb;
}

// projection of `other` on the cell's line and its squared distance to it, in the reference's order of operations
// (:88-89 / :120-123 inner_product, :9-27 sq_distance_to_line); cur / grad: the cell and its unit gradient
__device__ __forceinline__ void asv_pair_literal(const double* cur, const double* grad, const double* __restrict__ other,
                                                 int g, double sigma2, double& proj, double& lw) {
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
}

// Bitonic sort of a[0, np2) (np2 a power of two) in GLOBAL memory by the whole block, through an LDS window of B elements
// (B a power of two): every stage whose partners lie inside an aligned window runs there (a window is loaded, taken through
// all such steps, stored), only the steps that reach across windows (j >= B) go to global memory one by one.  A list of
// 32 768 pairs: 4 window rounds + 6 global steps instead of 120 global passes.  less(x, y): x sorts in front of y.
template <class E, class Less>
__device__ __forceinline__ void asv_sort_global(E* a, int np2, E* win, int B, int tid, Less less) {
    auto pass_done = [&]() {  // what the other threads of the block wrote must be what the next step reads
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    };
    if (B > np2) B = np2;
    // the steps j = jtop, jtop / 2, ..., 1 of stage k inside every window (jtop < B)
    auto windows = [&](int k_lo, int k_hi, bool all_steps) {
        for (int w0 = 0; w0 < np2; w0 += B) {
            for (int i = tid; i < B; i += T) win[i] = a[w0 + i];
            __syncthreads();
            for (int k = k_lo; k <= k_hi; k <<= 1)
                for (int j = all_steps ? (k >> 1) : (B >> 1); j > 0; j >>= 1) {
                    for (int p = tid; p < B / 2; p += T) {
                        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), q = i | j;
                        const E x = win[i], y = win[q];
                        const bool up = ((w0 + i) & k) == 0;
                        if (up ? less(y, x) : less(x, y)) {
                            win[i] = y;
                            win[q] = x;
                        }
                    }
                    __syncthreads();
                }
            for (int i = tid; i < B; i += T) a[w0 + i] = win[i];
            __syncthreads();
        }
        pass_done();
    };
    pass_done();
    windows(2, B, true);  // every stage k <= B: all its steps stay inside a window
    for (int k = 2 * B; k <= np2; k <<= 1) {
        for (int j = k >> 1; j >= B; j >>= 1) {
#pragma unroll 4
            for (int p = tid; p < np2 / 2; p += T) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), q = i | j;
                const E x = a[i], y = a[q];
                const bool sw = ((i & k) == 0) ? less(y, x) : less(x, y);
                a[i] = sw ? y : x;
                a[q] = sw ? x : y;
            }
            pass_done();
        }
        windows(k, k, false);  // the rest of stage k: steps j < B
    }
}

// asv_row_scan in PIECES with a block-wide step in the middle of each: `pre` sees every element of a piece, then `hook(end)`
// runs -- it may synchronise the block (every thread makes every call, the loop bounds are uniform) and returns true to end
// the scan --, then `test` sees the piece's elements again (they are still in registers).  A piece is a row of T elements
// over the first two batches (the bounds of a chain tighten fastest at its start) and a batch of AT_U rows from then on.
template <class Load, class Pre, class Hook, class Test>
__device__ __forceinline__ void asv_row_scan_pieces(Load load, int64_t jstart, int n, int crot, int last_block, int tid,
                                                    Pre pre, Hook hook, Test test) {
    const int64_t jt = jstart + tid;
    const int jlow = (int)(jt & 63), jb_t = (int)(jt >> 6);
    double pa[AT_U], wa[AT_U], pb[AT_U], wb[AT_U];
    auto fetch = [&](double (&p)[AT_U], double (&w)[AT_U], int base) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < AT_U; ++u) {
            int jb = jb_t + ((base + u * T) >> 6);
            jb = jb < last_block ? jb : last_block;
            const int64_t off = (int64_t)jb * (AT_C * 64) + (((crot + jb) & (AT_C - 1)) << 6) + jlow;
            load(off, p[u], w[u]);
        }
    };
    bool stop = false;
    auto consume = [&](const double (&p)[AT_U], const double (&w)[AT_U], int base) __attribute__((always_inline)) {
        if (base < 2 * AT_U * T) {
#pragma unroll
            for (int u = 0; u < AT_U; ++u) {
                if (stop) break;
                pre(p[u], w[u], base + u * T + tid);
                stop = hook(base + (u + 1) * T);
                if (!stop) test(p[u], w[u], base + u * T + tid);
            }
        } else {
#pragma unroll
            for (int u = 0; u < AT_U; ++u) pre(p[u], w[u], base + u * T + tid);
            stop = hook(base + AT_U * T);
            if (!stop) {
#pragma unroll
                for (int u = 0; u < AT_U; ++u) test(p[u], w[u], base + u * T + tid);
            }
        }
    };
    fetch(pa, wa, 0);
    for (int base = 0; base < n && !stop; base += 2 * AT_U * T) {
        fetch(pb, wb, base + AT_U * T);
        consume(pa, wa, base);
        fetch(pa, wa, base + 2 * AT_U * T);
        if (base + AT_U * T < n && !stop) consume(pb, wb, base + AT_U * T);
    }
}

// The streamed cells of one call, in stream order (the own batch's restricted cells, then the reference's), as ONE
// contiguous matrix with their squared norms and -- for the own batch -- their cell ids: the tile kernel then reads plain
// consecutive rows (coalesced, prefetchable any distance ahead) instead of chasing restrict[] -> row -> norm per step.
__global__ __launch_bounds__(256) void asv_gather_stream(const double* __restrict__ data1, const double* __restrict__ data2,
                                                         int g, int gs, const int32_t* __restrict__ r1, int nr1,
                                                         const int32_t* __restrict__ r2, int nr2, int64_t npad,
                                                         double* __restrict__ S, double* __restrict__ snrm,
                                                         int32_t* __restrict__ sid) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= npad) return;
    const bool live = j < (int64_t)nr1 + nr2, own = j < nr2;
    const int rid = !live ? -1 : (own ? r2[j] : r1[j - nr2]);
    const double* src = live ? (own ? data2 : data1) + (int64_t)rid * g : nullptr;
    double sq = 0.0;
    const bool aug = gs >= g + 2;  // (the register-streamed form: column g holds 1, column g + 1 the squared norm)
    for (int k = lane; k < gs; k += 64) {  // (rows of gs >= g values and npad >= nr1 + nr2 rows: zeros beyond the data)
        const double v = live && k < g ? src[k] : 0.0;
        if (!(aug && k == g + 1)) S[j * gs + k] = (aug && k == g && live) ? 1.0 : v;
        sq += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) {
        snrm[j] = sq;
        sid[j] = own ? rid : -1;
        if (aug) S[j * gs + g + 1] = live ? sq : 0.0;
    }
}

// snrm[n] = max of snrm[0, n): non-negative doubles order like their bit patterns (one workgroup's maximum per atomic)
__global__ __launch_bounds__(256) void asv_max_norm(double* __restrict__ snrm, int64_t n, unsigned int* __restrict__ gbar,
                                                    unsigned int* __restrict__ tile_ctr) {
    __shared__ double sm[256];
    if (gbar && blockIdx.x == 0 && threadIdx.x == 0) *gbar = 0u;  // (the tile kernel's round barrier starts from zero)
    if (tile_ctr && blockIdx.x == 0 && threadIdx.x == 0) *tile_ctr = 0u;  // (... and so does its tile counter)
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmax(m, snrm[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned long long*>(snrm + n), (unsigned long long)__double_as_longlong(sm[0]));
}

// NB8 > 0 (= ceil(g / 8), g <= 128): the streamed cells go from global memory straight into the B-operand registers.  A row
// is taken in blocks of 8 dimensions, each two MFMA steps: step (q, t) multiplies dimensions 8 q + 2 (lane >> 4) + t, so the
// two values a lane needs of a block are 16 contiguous bytes of its row and a row costs ceil(g / 8) * 8 dimensions of
// matrix work (104 at 100 PCs; blocks of 32 dimensions, the first form, paid for 128).  Every wave runs its 16 streamed
// cells on its own -- no staging through the LDS, no barrier in the stream, the next step's rows in flight while the
// current ones multiply.  NB8 = 0: the staged form (any g <= 256).
// diagnostics (bmx_dev_get "asv_ticks_stream" / "_wait" / "_cells"): 100 MHz ticks the workgroups of the tiled form spent in the
// stream, at the round barrier and in the per-cell phase, added up over all workgroups since the last "asv_tally_reset"
__device__ unsigned long long g_asv_ticks[8];  // stream, barrier wait, per-cell phase, (spare); literal cells: selection + re-evaluation + sort, chains + walk, kept addends, tiles with chains

template <int NB8>
__global__ __launch_bounds__(T) void asv_tile_kernel(int g, const double* __restrict__ data2, int n2,
                                                     const double* __restrict__ vect, double sigma2, int nr1, int nr2,
                                                     const double* __restrict__ S, const double* __restrict__ snrm,
                                                     const int32_t* __restrict__ sid, double* __restrict__ out,
                                                     double* __restrict__ scratch, int cell_begin, int cell_end, int lcap,
                                                     unsigned long long* __restrict__ tally, unsigned int* __restrict__ gbar,
                                                     unsigned int* __restrict__ tile_ctr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int GP = asv_tile_gp(g);
    double* cx = reinterpret_cast<double*>(smem_raw);  // [16][GP] the tile's cells
    double* cg = cx + AT_C * GP;                        // [16][GP] their unit gradients (both live through the per-cell phase: the literal re-run reads them)
    // one region, three lives: the A operands in lane order during the stream (cxp / cgp, NB8 > 8), the collected bin of the
    // histogram walk (lp / lw_), the sort buffer of the literal re-run (2 lcap doubles)
    double* ub = cg + AT_C * GP;
    const int ub_doubles = asv_tile_ub_doubles(g, lcap);
    double* lp = ub;                                                             // [CAP] collected projections
    unsigned long long* lw_ = reinterpret_cast<unsigned long long*>(lp + AT_CAP);  // [CAP] and their weights
    double* rs = ub + ub_doubles;  // [64][KC + 2] a step of streamed cells (staged form only)
    double* sc_proj = rs + (NB8 == 0 ? AT_R * (AT_KC + 2) : 0);  // per cell: own projection, |x|^2, |vect|, maxima, projection range
    double* sc_n = sc_proj + AT_C;
    double* sc_l2 = sc_n + AT_C;
    double* sc_mx1 = sc_l2 + AT_C;
    double* sc_mx2 = sc_mx1 + AT_C;
    double* sc_lo = sc_mx2 + AT_C;
    double* sc_hi = sc_lo + AT_C;
    double* sc_tmp = sc_hi + AT_C;
    double* etab = sc_tmp + AT_C;                       // [32] 2^(j / 32) for asv_exp_neg
    double* sm = etab + 32;                             // [AT_SM] block reductions
    unsigned long long* hist = reinterpret_cast<unsigned long long*>(sm + AT_SM);  // [NB]
    unsigned long long* binmax = hist + AT_NB;  // [NB] per projection bin: the largest log-weight (bit pattern), then the bin's threshold
    double* cxp = ub;                                       // [2 NB8][64] the cells' coordinates as the lanes read them
    double* cgp = cxp + (NB8 > 8 ? NB8 * 2 * 64 : 0);       // [2 NB8][64] the unit gradients likewise (NB8 > 8)
    __shared__ int sh_cnt, sh_bin, sh_tile;
    __shared__ unsigned long long sh_before;
    __shared__ int sh_sel[4];        // literal re-run: kept addends (own batch, reference in restrict order, in projection order), abort flag
    __shared__ int sh_K[AT_C][3];    // per cell of the tile: the lengths of its three lists (-1: the cell went the histogram way)
    __shared__ double sh_chain[AT_C][3];  // per cell: totalprob2, prob2, totalprob1 as the chains leave them
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t N = (int64_t)nr1 + nr2;  // streamed cells: the own batch's restricted cells first, then the reference's
    // the tile's scratch: per block of 64 streamed cells, 16 rows (cells) of 64 values -- a step of the stream writes ONE
    // contiguous 8 KB piece of each array (the [cell][N] layout of the first version wrote 32 rows 8 MB apart per step and
    // spent its time in address translation), and a cell's row is 512-byte pieces 8 KB apart, read by whole waves
    const int64_t Npad = N <= AT_NP ? AT_NP : (N + AT_NP - 1) / AT_NP * AT_NP;
    const int64_t Nown = lcap > 0 ? (int64_t)asv_tile_npad((size_t)nr2) : 0;  // (the own batch's part of the stream, padded likewise)
    const int64_t per_block = 2 * AT_C * Npad + asv_tile_list_doubles(lcap) + AT_C * Nown / 2;
    double* SP = scratch + (int64_t)blockIdx.x * per_block;  // projections
    double* SW = SP + (int64_t)AT_C * Npad;                   // log-weights
    // the own batch's pairs, for the literal re-run only (it picks the kept addends of a flagged cell's own-batch chains from
    // them): ONE float per pair, the log-weight with its two lowest mantissa bits replaced by a code -- 0: above the cell's
    // projection (not counted in prob2, :90-92), 1: within rounding of it, 2: at or below it, 3: the cell itself
    float* SO = reinterpret_cast<float*>(SW + (int64_t)AT_C * Npad + asv_tile_list_doubles(lcap));
    // -- and a cell's slot inside a block's piece rotates with the block number: with a fixed slot the per-cell passes
    // walked the scratch at a stride of exactly 8 KB, i.e. through a handful of the memory channels
    const int rot0 = blockIdx.x;
    auto at = [rot0](int c, int64_t j) { return (((j >> 6) * AT_C + ((c + (int)(j >> 6) + rot0) & (AT_C - 1))) << 6) + (j & 63); };
    // (a rank of a multi-GPU run owns the cells [cell_begin, cell_end) only; n2 below is the end of that range)
    n2 = cell_end;
    const int ntiles = (cell_end - cell_begin + AT_C - 1) / AT_C;
    const double NEG = -__builtin_inf(), POS = __builtin_inf();
    unsigned long long tk_stream = 0, tk_wait = 0, tk_cells = 0, tk0 = 0, tk1 = 0;  // (thread 0's)
    unsigned long long tk_lit = 0, tk_chain = 0, tk_lit_n = 0, tk_chain_tiles = 0;   // (the literal re-run's share of tk_cells)

    // Tiles are handed out as workgroups come free (round 6: a counter on the device): a tile with flagged cells takes several
    // times a plain one -- with tiles dealt in a fixed stride the workgroups of a launch ended 11 % apart on config 5 at
    // sigma 1 (5.7 % without the re-run).  Every cell's value is independent of which workgroup computes it and when.  (With the
    // testing hook "asv_sync" the rounds need the fixed deal.)
    for (int tile = blockIdx.x;; ) {
        if (tile_ctr && !gbar) {
            __syncthreads();  // (sh_tile of the previous turn has been read by everybody)
            if (tid == 0) sh_tile = (int)atomicAdd(tile_ctr, 1u);
            __syncthreads();
            tile = sh_tile;
        }
        if (tile >= ntiles) break;
        const int c0 = cell_begin + tile * AT_C;
        // ---- the tile's cells: coordinates, unit gradient (:57-70), own projection
        for (int e = tid; e < AT_C * GP; e += T) {
            const int c = e / GP, x = e - c * GP;
            const bool in = c0 + c < n2 && x < g;
            cx[e] = in ? data2[(int64_t)(c0 + c) * g + x] : 0.0;
            cg[e] = in ? vect[(int64_t)(c0 + c) * g + x] : 0.0;
        }
        __syncthreads();
        if (tid < AT_C) {
            double l2 = 0.0, nn = 0.0;
            for (int x = 0; x < g; ++x) l2 += cg[tid * GP + x] * cg[tid * GP + x];
            l2 = sqrt(l2);
            if (l2 != 0.0)
                for (int x = 0; x < g; ++x) cg[tid * GP + x] /= l2;
            double p = 0.0;
            for (int x = 0; x < g; ++x) {
                p += cg[tid * GP + x] * cx[tid * GP + x];
                nn += cx[tid * GP + x] * cx[tid * GP + x];
            }
            sc_l2[tid] = l2;
            sc_proj[tid] = p;
            sc_n[tid] = nn;
        }
        __syncthreads();
        // ---- testing hook "asv_sync" (OFF by default): the workgroups start the stream of their round's tiles TOGETHER and
        // run in convoy, the first to ask for a row of S brings it into its XCD's L2 for the other 31.  That halves the kernel's
        // HBM reads (PMC: 3.36 -> 1.56 TB on a mid-size call) and leaves its time where it was -- the stream's loads are
        // prefetched a block ahead and hidden either way -- while the spread of the per-cell phases becomes idle time every
        // round (config 5: 9.94 -> 10.78 s per step; rounds 4 and 6, EXPERIMENTS.md).  One arrival per TILE on a device word
        // zeroed before the launch; the tiles of round r go on when all tiles of rounds <= r have arrived (or 50 ms have gone
        // by).  gbar is null unless the hook is on and the launch is no wider than the device.
        if (tid < 32) etab[tid] = exp2((double)tid * 0.03125);  // (for asv_exp_neg)
        __syncthreads();
        tk0 = __builtin_amdgcn_s_memrealtime();
        if (gbar) {
            if (tid == 0) {
                const unsigned int round = (unsigned int)((tile - (int)blockIdx.x) / (int)gridDim.x);
                const unsigned long long want_ = (unsigned long long)(round + 1u) * gridDim.x;
                const unsigned int want = want_ < (unsigned long long)ntiles ? (unsigned int)want_ : (unsigned int)ntiles;
                __hip_atomic_fetch_add(gbar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                // (the barrier is a matter of speed, never of correctness: a workgroup that has waited 50 ms -- another
                // process's kernel holds CUs this launch counted on -- goes on by itself)
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                while (__hip_atomic_load(gbar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want &&
                       __builtin_amdgcn_s_memrealtime() - t0 < 5000000ull)
                    __builtin_amdgcn_s_sleep(16);
            }
            __syncthreads();
        }
        tk1 = __builtin_amdgcn_s_memrealtime();
        tk_wait += tk1 - tk0;
        // ---- pass over the streamed cells: projections and log-weights of every (cell, streamed cell) pair
        // own batch: log-sum-exp of every cell's weights, all of them and those at or below its projection (:74-112), taken
        // ONLINE in the stream's epilogue -- running maximum om, sums relative to it -- so that the own batch's 40 % of the
        // (cell, streamed cell) pairs are not read back (round 3: 16 + 16 bytes per pair); they are written all the same when
        // the literal re-run is on (lit_on): a flagged cell picks the kept addends of its own-batch chains from them
#ifdef BMX_ASV_AB_NOLIT
        const bool lit_on = false;
#else
        const bool lit_on = lcap > 0;
#endif
#ifdef BMX_ASV_AB_NOSO  // timing build: the literal re-run without the own batch's packed pairs (its results are then wrong)
        const bool so_on = false;
#else
        const bool so_on = true;
#endif
        double om[4], oa[4], ob[4];
        double mx1[4], mx2[4], lo[4], hi[4], cp[4], cn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            om[i] = NEG;
            oa[i] = ob[i] = 0.0;
            mx1[i] = mx2[i] = NEG;
            lo[i] = POS;
            hi[i] = NEG;
            cp[i] = sc_proj[(lane >> 4) + 4 * i];
            cn[i] = sc_n[(lane >> 4) + 4 * i];
        }
        double tol_tile = 0.0;  // >= the rounding of g . x for any cell of the tile and any streamed cell (|g| = 1)
        if (lit_on) {
            double cmx = 0.0;
            for (int c = 0; c < AT_C; ++c) cmx = fmax(cmx, sc_n[c]);
            tol_tile = 1e-13 * (1.0 + cmx + snrm[Npad]);
        }
        // the partial log-sum-exps of the 16 lanes that share a cell: to their common maximum, then added; parked per (wave,
        // cell) in the block-reduction area (nobody else uses it during the stream) -- see the reduction behind the stream
        auto own_done = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                for (int o = 1; o < 16; o <<= 1) {
                    const double pm = __shfl_xor(om[i], o), pa = __shfl_xor(oa[i], o), pb = __shfl_xor(ob[i], o);
                    const double mm = fmax(om[i], pm);
                    const double f0 = om[i] == mm ? 1.0 : exp(om[i] - mm), f1 = pm == mm ? 1.0 : exp(pm - mm);
                    oa[i] = oa[i] * f0 + pa * f1;
                    ob[i] = ob[i] * f0 + pb * f1;
                    om[i] = mm;
                }
                if ((lane & 15) == 0) {
                    const int c = (lane >> 4) + 4 * i;
                    sm[(w * AT_C + c) * 6 + 3] = om[i];
                    sm[(w * AT_C + c) * 6 + 4] = oa[i];
                    sm[(w * AT_C + c) * 6 + 5] = ob[i];
                }
            }
        };
        if constexpr (NB8 > 0) {
            typedef double d2a __attribute__((ext_vector_type(2)));
            constexpr int GS = NB8 * 8;  // row stride of the stream (zero filled beyond g)
            constexpr int NST = 2 * NB8;  // MFMA steps per product
            const int kq = lane >> 4;
            // the A operands of MFMA step st = 2 q + t: lane l holds cell l & 15, dimension 8 q + 2 (l >> 4) + t.  Up to 64
            // dimensions they live in registers for the whole stream, beyond that in the LDS in lane order.
            // Round 6: the distance chain's A operand is the cell SCALED and AUGMENTED -- (2 / sigma) x_c, then -|x_c|^2 / sigma
            // against the row's column of ones and -1 / sigma against its squared norm -- so that the chain delivers
            // -|x_c - x_o|^2 / sigma and the epilogue is left with one multiply, one fused multiply-add and a clamp per pair
            // where it had an add, a multiply, two subtractions, a clamp and a division (FP64 vector work does not run under
            // FP64 MFMAs on this part: EXPERIMENTS.md round 6).  The values differ from the division's by rounding of the order
            // of 1e-14 (|x_c|^2 + |x_o|^2) / sigma, inside the margin `mb` every use of them carries.
            const double isig = 1.0 / sigma2;
            auto a_dist = [&](int cell, int k) -> double {
                return k < g ? cx[cell * GP + k] * (2.0 * isig) : (k == g ? -sc_n[cell] * isig : (k == g + 1 ? -isig : 0.0));
            };
            double axr[NB8 <= 8 ? NST : 1], agr[NB8 <= 8 ? NST : 1];
            if constexpr (NB8 <= 8) {
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int k = 8 * (st >> 1) + 2 * kq + (st & 1);
                    axr[st] = a_dist(lane & 15, k);
                    agr[st] = cg[(lane & 15) * GP + k];
                }
            } else {
                for (int e = tid; e < NST * 64; e += T) {
                    const int st = e >> 6, ln = e & 63;
                    const int k = 8 * (st >> 1) + 2 * (ln >> 4) + (st & 1);
                    cxp[e] = a_dist(ln & 15, k);
                    cgp[e] = cg[(ln & 15) * GP + k];
                }
                __syncthreads();
            }
            // this lane's streamed cell of step j0 is j0 + jl; rows, norms, ids and scratch are padded to whole pairs of
            // steps: no bounds checks (and no divergent branches) in the stream
            const int jl = 16 * w + (lane & 15);
            const double* srow = S + (int64_t)jl * GS + 2 * kq;
            double* spo = SP + jl;
            double* swo = SW + jl;
            // own-batch codes: [block of 64 streamed cells][kq][64][4 cells kq + 4 i] floats
            f4* so_ = reinterpret_cast<f4*>(SO) + kq * 64 + jl;
            auto load_rows = [&](double (&b)[NST], const double* src) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < NB8; ++q) {
                    const d2a v = *reinterpret_cast<const d2a*>(src + 8 * q);
                    b[2 * q] = v[0];
                    b[2 * q + 1] = v[1];
                }
            };
            // One 64-row block of the stream for this wave's 16 rows of it.  Round 6: on this part NOTHING runs under an FP64 MFMA
            // -- every vector instruction of a SIMD, whichever wave it comes from, adds its ~4 cycles to the 64 of each MFMA
            // (scripts/microbench/mfma_f64_valu.hip) -- so what a block costs is its 52 MFMAs plus the COUNT of everything else,
            // and rounds 3-5's single loop carried ~450 vector instructions a block (per-element selects between the two kinds
            // of rows, SGPR spills, accumulator moves).  Hence one loop per KIND of block (MODE), each with only its own running
            // state and its own half of the epilogue, and ONE row set per wave, each piece re-loaded right behind the two MFMAs
            // that consumed it (the same prefetch distance as a second buffer, 52 registers less):
            //   0 every row of the block belongs to the OWN batch: the online log-sum-exps (and the float codes of the re-run);
            //   1 a block that holds the end of the own batch (or rows of both kinds, or padding): everything, by selects;
            //   2 every row belongs to the REFERENCE batch: maxima, projection range, the scratch stores;
            //   3 reference rows and the stream's zero padding behind them.
            auto step = [&](auto mode_tag, double (&b)[NST], int64_t j0) __attribute__((always_inline)) {
                constexpr int MODE = decltype(mode_tag)::value;
                const int64_t jo = j0 + jl;
                const double* nsrc = srow + (j0 + AT_R < Npad ? (j0 + AT_R) : j0) * GS;  // (the last block asks for its own again)
                int rid = -1;
                if constexpr (MODE <= 1) rid = sid[jo];
                // two accumulator pairs (k-steps alternate): D and P chains are independent of each other as well
                d4 Dq[2], Pq[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) Dq[a] = Pq[a] = d4{0.0, 0.0, 0.0, 0.0};
                // (NB8 > 8: the A operands come from the LDS, asked for AHEAD k-steps before the MFMAs that take them -- read
                // right in front of its MFMAs, as the compiler places them, every pair of k-steps waits out the LDS's latency:
                // the bare chains ran at 0.79 of the matrix pipe's rate)
                constexpr int AHEAD = 4;
                double axq[NB8 > 8 ? AHEAD : 1], agq[NB8 > 8 ? AHEAD : 1];
                if constexpr (NB8 > 8) {
#pragma unroll
                    for (int t = 0; t < AHEAD; ++t) {
                        axq[t] = cxp[t * 64 + lane];
                        agq[t] = cgp[t * 64 + lane];
                    }
                }
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    double ax, ag;
                    if constexpr (NB8 <= 8) {
                        ax = axr[st];
                        ag = agr[st];
                    } else {
                        ax = axq[st % AHEAD];
                        ag = agq[st % AHEAD];
                        if (st + AHEAD < NST) {
                            axq[st % AHEAD] = cxp[(st + AHEAD) * 64 + lane];
                            agq[st % AHEAD] = cgp[(st + AHEAD) * 64 + lane];
                        }
                    }
                    Dq[st & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b[st], Dq[st & 1], 0, 0, 0);
                    Pq[st & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag, b[st], Pq[st & 1], 0, 0, 0);
                    if (st & 1) {  // both values of this 8-dimension block are consumed: the next block's piece takes their place
                        const d2a v = *reinterpret_cast<const d2a*>(nsrc + 8 * (st >> 1));
                        b[st - 1] = v[0];
                        b[st] = v[1];
                    }
                    if constexpr (NB8 > 8) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // (two MFMAs, then what belongs behind them)
                }
                const bool own = MODE == 0 ? true : (MODE >= 2 ? false : jo < nr2);
                const bool ref = MODE == 0 ? false : (MODE == 2 ? true : (!own && jo < N));
                const int blk = (int)(j0 >> 6);
                f4 oc = f4{0.f, 0.f, 0.f, 0.f};
                (void)tol_tile;
                double* sp_ = spo + (int64_t)blk * (AT_C * 64);
                double* sw_ = swo + (int64_t)blk * (AT_C * 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // this lane: streamed cell jo, tile cells (lane >> 4) + 4 i
                    double pr = Pq[0][i] + Pq[1][i];
                    const double s_ = cp[i] - pr;
                    // -(|x_c - x_o|^2 - s^2) / sigma: the chain's value plus s^2 / sigma, never above zero
                    double lw = __builtin_fma(s_, s_ * isig, Dq[0][i] + Dq[1][i]);
                    lw = lw < 0.0 ? lw : -0.0;  // (log-weights are <= -0, as -d2 / sigma was: the per-bin maxima order them by bit pattern)
                    bool self = false;
                    if constexpr (MODE <= 1) {
                        self = rid == c0 + kq + 4 * i;  // the cell itself: log-weight 0, always counted (:80-84)
                        lw = self ? 0.0 : lw;
                        pr = self ? NEG : pr;
                    }
                    if constexpr (MODE >= 1) {
                        // (selects, not branches: the padding rows of the stream belong to neither batch)
                        mx1[i] = fmax(mx1[i], ref ? lw : NEG);
                        lo[i] = fmin(lo[i], ref ? pr : POS);
                        hi[i] = fmax(hi[i], ref ? pr : NEG);
                    }
                    if constexpr (MODE <= 1) {
                        if (own) {  // (whole waves but for the one block where the own batch ends)
                            if (lw > om[i]) {  // a new maximum: rare once the cell itself (log-weight 0) has gone by
                                const double f = asv_exp_neg(om[i] - lw, etab);  // exp(-inf) = 0 the first time
                                oa[i] *= f;
                                ob[i] *= f;
                                om[i] = lw;
                            }
                            const double e = asv_exp_neg(lw - om[i], etab);
                            oa[i] += e;
                            ob[i] += !(pr > cp[i]) ? e : 0.0;
                        }
                    }
                    if (!own) {
                        const int slot = (kq + 4 * i + blk + rot0) & (AT_C - 1);  // == at(kq + 4 i, jo)
                        sp_[slot * 64] = pr;
                        sw_[slot * 64] = lw;
                    } else if (lit_on && so_on) {
                        // (s_ = the cell's projection minus this one's; tol_tile >= the rounding of the two, for every cell of the tile)
                        const unsigned code = self ? 3u : (s_ >= tol_tile ? 2u : (s_ >= -tol_tile ? 1u : 0u));
                        oc[i] = __uint_as_float((__float_as_uint((float)fmax(lw, -3.0e38)) & ~3u) | code);
                    }
                }
                // (this lane's four cells kq, kq + 4, kq + 8, kq + 12 of one streamed cell: one 16-byte store, 16 lanes a 256-byte run)
                if (own && lit_on && so_on) so_[(int64_t)blk * (AT_C * 16)] = oc;
            };
            double ba[NST];
            int64_t j0 = 0;
            load_rows(ba, srow);
            for (; j0 + AT_R <= nr2; j0 += AT_R) step(std::integral_constant<int, 0>{}, ba, j0);
            if (j0 < nr2) {
                step(std::integral_constant<int, 1>{}, ba, j0);
                j0 += AT_R;
            }
            own_done();  // (the own batch's sums leave the registers: the reference loop runs without them; see below)
            for (; j0 + AT_R <= N; j0 += AT_R) step(std::integral_constant<int, 2>{}, ba, j0);
            for (; j0 < Npad; j0 += AT_R) step(std::integral_constant<int, 3>{}, ba, j0);
        } else {
        // staging: 64 rows x 4 segments of 8 doubles per step of 32 dimensions; the step after the one being multiplied is
        // already on its way into registers (16-byte loads where the rows allow), the row pointers one streamed block ahead
        typedef double d2a __attribute__((ext_vector_type(2)));
        const int lr = tid >> 2, seg = (tid & 3) * 8;
        const bool vec = (g & 1) == 0;
        const int nkc = (g + AT_KC - 1) / AT_KC;
        auto row_ptr = [&](int64_t jr) -> const double* { return jr < N ? S + jr * g : nullptr; };
        double pf[8];
        auto fetch = [&](const double* src, int k0) __attribute__((always_inline)) {
            if (src && vec && k0 + seg + 8 <= g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const d2a v = *reinterpret_cast<const d2a*>(src + k0 + seg + 2 * e);
                    pf[2 * e] = v[0];
                    pf[2 * e + 1] = v[1];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + seg + e;
                    pf[e] = (src && k < g) ? src[k] : 0.0;
                }
            }
        };
        const double* src = row_ptr(lr);
        const double* src_next = row_ptr((int64_t)AT_R + lr);
        fetch(src, 0);
        for (int64_t j0 = 0; j0 < N; j0 += AT_R) {
            d4 D = d4{0.0, 0.0, 0.0, 0.0}, P = d4{0.0, 0.0, 0.0, 0.0};
            // this lane's streamed cell of the step: its norm and id are asked for now and used after the products
            const int64_t jo = j0 + 16 * w + (lane & 15);
            const double no = jo < N ? snrm[jo] : 0.0;
            const int rid = jo < N ? sid[jo] : -1;
            for (int kc = 0; kc < nkc; ++kc) {
                const int k0 = kc * AT_KC;
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[lr * (AT_KC + 2) + seg + e] = pf[e];
                __syncthreads();
                if (kc + 1 < nkc) {
                    fetch(src, k0 + AT_KC);
                } else {  // the next streamed block's first step; its successor's row pointer starts its own round trip
                    src = src_next;
                    fetch(src, 0);
                    src_next = row_ptr(j0 + 2 * AT_R + lr);
                    (void)src_next;
                }
#pragma unroll
                for (int kk = 0; kk < AT_KC / 4; ++kk) {
                    const double b = rs[(16 * w + (lane & 15)) * (AT_KC + 2) + 4 * kk + (lane >> 4)];
                    const double ax = cx[(lane & 15) * GP + k0 + 4 * kk + (lane >> 4)];
                    const double ag = cg[(lane & 15) * GP + k0 + 4 * kk + (lane >> 4)];
                    D = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b, D, 0, 0, 0);
                    P = __builtin_amdgcn_mfma_f64_16x16x4f64(ag, b, P, 0, 0, 0);
                }
                __syncthreads();
            }
            // this lane: streamed cell jo, tile cells (lane >> 4) + 4 i
            if (jo < N) {
                const bool own = jo < nr2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = (lane >> 4) + 4 * i;
                    double pr = P[i];
                    const double s_ = cp[i] - pr;
                    double d2 = (cn[i] + no) - 2.0 * D[i] - s_ * s_;
                    d2 = d2 > 0.0 ? d2 : 0.0;
                    double lw = -d2 / sigma2;
                    if (own && rid == c0 + c) {  // the cell itself: log-weight 0, always counted (:80-84)
                        lw = 0.0;
                        pr = NEG;
                    }
                    mx1[i] = fmax(mx1[i], own ? NEG : lw);
                    lo[i] = fmin(lo[i], own ? POS : pr);
                    hi[i] = fmax(hi[i], own ? NEG : pr);
                    if (own) {
                        if (lw > om[i]) {
                            const double f = exp(om[i] - lw);
                            oa[i] *= f;
                            ob[i] *= f;
                            om[i] = lw;
                        }
                        const double e = exp(lw - om[i]);
                        oa[i] += e;
                        ob[i] += !(pr > cp[i]) ? e : 0.0;
                    }
                    if (!own) {
                        SP[at(c, jo)] = pr;
                        SW[at(c, jo)] = lw;
                    } else if (lit_on) {
                        const unsigned code = rid == c0 + c ? 3u : (s_ >= tol_tile ? 2u : (s_ >= -tol_tile ? 1u : 0u));
                        SO[(((jo >> 6) * 4 + (c & 3)) * 64 + (jo & 63)) * 4 + (c >> 2)] =
                            __uint_as_float((__float_as_uint((float)fmax(lw, -3.0e38)) & ~3u) | code);
                    }
                }
            }
        }
        }
        // per-cell maxima and projection range: over the 16 lanes of a row group, then over the waves
        if constexpr (NB8 == 0) own_done();  // (the register-streamed form parked the own batch's sums when its last own block was through)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            for (int o = 1; o < 16; o <<= 1) {
                mx1[i] = fmax(mx1[i], __shfl_xor(mx1[i], o));
                lo[i] = fmin(lo[i], __shfl_xor(lo[i], o));
                hi[i] = fmax(hi[i], __shfl_xor(hi[i], o));
            }
        }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = (lane >> 4) + 4 * i;
                sm[(w * AT_C + c) * 6 + 0] = mx1[i];
                sm[(w * AT_C + c) * 6 + 1] = lo[i];
                sm[(w * AT_C + c) * 6 + 2] = hi[i];
            }
        }
        __syncthreads();
        if (tid < AT_C) {
            double a = NEG, l = POS, h = NEG, m2 = NEG;
            for (int ww = 0; ww < 4; ++ww) {
                a = fmax(a, sm[(ww * AT_C + tid) * 6 + 0]);
                l = fmin(l, sm[(ww * AT_C + tid) * 6 + 1]);
                h = fmax(h, sm[(ww * AT_C + tid) * 6 + 2]);
                m2 = fmax(m2, sm[(ww * AT_C + tid) * 6 + 3]);
            }
            double all = 0.0, below = 0.0;  // the four waves' partial sums to the common maximum, in wave order
            for (int ww = 0; ww < 4; ++ww) {
                const double pm = sm[(ww * AT_C + tid) * 6 + 3];
                const double f = pm == m2 ? 1.0 : exp(pm - m2);
                all += sm[(ww * AT_C + tid) * 6 + 4] * f;
                below += sm[(ww * AT_C + tid) * 6 + 5] * f;
            }
            sc_mx1[tid] = a;
            // prob2 (:74-112): log-sum of the cells at or below the projection minus the log-sum of all of them; 0 - the
            // latter when none is at or below (:76: prob2 then keeps its starting value)
            sc_mx2[tid] = nr2 > 0 ? (below > 0.0 ? m2 + log(below) : 0.0) - (m2 + log(all)) : 0.0;
            sc_lo[tid] = l;
            sc_hi[tid] = h;
        }
        __threadfence_block();
        __syncthreads();
        // (the scratch rows were written by this block and are read by it: same CU, through the L2)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        tk0 = __builtin_amdgcn_s_memrealtime();
        tk_stream += tk0 - tk1;
        // ---- cell by cell: own-batch probability, then the weighted quantile of the reference projections
        const double nmax_s = snrm[Npad];  // the largest squared norm of a streamed cell (asv_max_norm; bounds the rounding of the GEMM-form distances)
        const int last_block = (int)(Npad >> 6) - 1;
        const int gs_rt = NB8 > 0 ? NB8 * 8 : g;  // row stride of the gathered stream
        if (tid < AT_C) sh_K[tid][0] = -1;
        int n_lit = 0, n_back = 0;  // (thread 0's tallies: cells re-run literally, flagged cells that went the histogram way)
        for (int c = 0; c < AT_C && c0 + c < n2; ++c) {
            // (own batch: streamed cells [0, nr2); reference batch: [nr2, N))
            auto w1 = [&](int64_t o_) { return SW[at(c, nr2 + o_)]; };
            const double curproj = sc_proj[c], l2 = sc_l2[c];
            const double prob2 = sc_mx2[c];  // (taken online by the stream, see above)
            double ref_quan = __builtin_nan("");
            bool literal = false;
            int cell_mode = 0;  // 0: the histogram way (not flagged), 1: re-run literally, 2: flagged, beyond the re-run
            if (nr1 > 0) {
                const double mx = sc_mx1[c];
                const double FIX = 1099511627776.0;  // 2^40
                double blo = sc_lo[c], bhi = sc_hi[c];  // projections still in play: [blo, bhi]
                unsigned long long before = 0;          // weight of the projections below blo
                double target = -1.0;                   // in fixed-point units, known after the first histogram
                ref_quan = sc_hi[c];                    // default: the last one (:141)
                // Flagged: (i) the own batch's cumulative probability at the cell is 1 to within 1e-6, i.e. the walk over the
                // reference batch runs to where all but 1e-6 of the weight has gone by -- and is decided by single addends of
                // relative size <= 1e-6, far enough down for the rounding of the two summation orders to matter; (ii) it is below
                // e^-12 (a cell that is not in its own batch's restrict vector can sit far below all the weight): the walk ends
                // among addends the 2^-40 fixed-point weights of the histogram do not resolve.  (Cells in between cross on
                // addends the histogram way resolves.)  How long a re-run a flagged cell is worth: within 1e-9 of 1 (or below
                // e^-12) the walk IS a statement about rounding and the histogram way picks another quantile on most such cells:
                // up to lcap addends per chain; between 1e-9 and 1e-6 the histogram way is right on all but a few cells of a
                // small call (and on every sampled cell of config 5 at sigma 1), and a re-run of tens of thousands of addends for
                // each of the 2.5 % of config 5's cells that sit there doubled the step: up to 8 192.
                const bool flagged = lit_on && (-prob2 < 1e-6 || prob2 < -12.0);
                const int lcap_c = (-prob2 < 1e-9 || prob2 < -12.0) ? lcap : (lcap < 8192 ? lcap : 8192);
                const unsigned long long NEGBITS = 0xFFF0000000000000ull;  // -inf
                // ---- the first histogram of the walk; for a flagged cell also every bin's largest log-weight and the number
                // of reference cells within 38.5 of the largest one (all of those are kept addends)
                for (int b = tid; b < AT_NB; b += T) {
                    hist[b] = 0ull;
                    binmax[b] = NEGBITS;
                }
                if (tid == 0) sh_cnt = 0;
                __syncthreads();
                const double scale0 = bhi > blo ? (double)AT_NB / (bhi - blo) : 0.0;
                int cG = 0;
                if (flagged) {
                    asv_row_scan<true>(SP, SW, nr2, nr1, c + rot0, last_block, tid, [&](double pr, double lw, int o) {
                        const bool in = o < nr1 && pr >= blo && pr <= bhi;
                        int b = (int)((pr - blo) * scale0);
                        b = b < 0 ? 0 : (b > AT_NB - 1 ? AT_NB - 1 : b);
                        const double wv = exp(lw - mx) * FIX;
                        atomicAdd(&hist[in ? b : (tid & (AT_NB - 1))], in ? (unsigned long long)wv : 0ull);
                        // (log-weights are <= -0: as bit patterns the largest value is the smallest pattern)
                        atomicMin(&binmax[in ? b : (tid & (AT_NB - 1))], in ? (unsigned long long)__double_as_longlong(lw) : NEGBITS);
                        cG += in && lw >= mx - 38.5 ? 1 : 0;
                    });
                } else {
                    asv_row_scan<true>(SP, SW, nr2, nr1, c + rot0, last_block, tid, [&](double pr, double lw, int o) {
                        const bool in = o < nr1 && pr >= blo && pr <= bhi;
                        int b = (int)((pr - blo) * scale0);
                        b = b < 0 ? 0 : (b > AT_NB - 1 ? AT_NB - 1 : b);
                        const double wv = exp(lw - mx) * FIX;
                        // (an element out of play adds nothing to a bin of the thread's own: no branch, no pile-up on one bin)
                        atomicAdd(&hist[in ? b : (tid & (AT_NB - 1))], in ? (unsigned long long)wv : 0ull);
                    });
                }
                __syncthreads();
                if (flagged) {
                    int* smi = reinterpret_cast<int*>(sm);
                    smi[tid] = cG;
                    __syncthreads();
                    for (int o2 = T / 2; o2 > 0; o2 >>= 1) {
                        if (tid < o2) smi[tid] += smi[tid + o2];
                        __syncthreads();
                    }
                    const int cGtot = smi[0];
                    __syncthreads();
                    if (cGtot <= lcap_c) {
                        const unsigned long long tkl0 = __builtin_amdgcn_s_memrealtime();
                        // ================= the literal re-run (see the comment in front of asv_noop_below) =================
                        const double cn_c = sc_n[c];
                        const double mb = 1e-3 + 1e-13 * (cn_c + nmax_s) / sigma2;      // >= the rounding of a GEMM-form log-weight
                        const double tolp = 1e-13 * (sqrt(cn_c) + sqrt(nmax_s)) + 1e-300;  // >= the rounding of a projection
                        // the lists, per cell of the tile: own batch (log-weight, counted-in-prob2 flag) in restrict order, reference
                        // log-weights in restrict order, reference (projection, log-weight) sorted; three index lists, reused per cell
                        double* LO = SW + (int64_t)AT_C * Npad;        // [16][lcap][2]
                        double* LR = LO + (int64_t)AT_C * lcap * 2;    // [16][lcap]
                        double* LS = LR + (int64_t)AT_C * lcap;        // [16][lcap][2]
                        int32_t* GI = reinterpret_cast<int32_t*>(LS + (int64_t)AT_C * lcap * 2);  // [3][lcap]
                        int32_t* giO = GI;
                        int32_t* giR = GI + lcap;
                        int32_t* giS = GI + 2 * lcap;
                        // (b) per bin: the threshold below which a reference cell cannot change the SORTED chain -- from the
                        // largest log-weight of the bins at least two back (every cell of those sorts in front of every cell of
                        // this bin whatever the rounding of the projections, as long as a bin is much wider than that rounding)
                        double* binthr = reinterpret_cast<double*>(binmax);
                        {
                            const bool wide = (bhi - blo) > 16.0 * AT_NB * tolp;
                            double v[8], m8 = NEG;
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                v[k] = __longlong_as_double((long long)binmax[8 * tid + k]);
                                m8 = fmax(m8, v[k]);
                            }
                            double inc = m8;
                            for (int o2 = 1; o2 < 64; o2 <<= 1) {
                                const double t2 = __shfl_up(inc, o2);
                                if (lane >= o2) inc = fmax(inc, t2);
                            }
                            __syncthreads();
                            if (lane == 63) sm[w] = inc;
                            __syncthreads();
                            double run = __shfl_up(inc, 1);
                            if (lane == 0) run = NEG;
                            for (int ww = 0; ww < w; ++ww) run = fmax(run, sm[ww]);
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const double E = run;  // largest log-weight of the bins in front of bin 8 tid + k
                                run = fmax(run, v[k]);
                                if (8 * tid + k + 1 < AT_NB)
                                    binthr[8 * tid + k + 1] = wide ? asv_thr_neg(E, mx, (double)nr1, mb) : NEG;
                            }
                            if (tid == 0) {
                                binthr[0] = NEG;
                                sh_sel[0] = sh_sel[1] = sh_sel[2] = 0;
                            }
                            __syncthreads();
                        }
                        // (c) the reference batch: kept addends of totalprob1's chain (restrict order) and of the walk's.  A piece's
                        // elements are held against the largest log-weight of the pieces in front of it; the chain never exceeds
                        // mx + log(nr1), mx the largest log-weight of all (from the stream).
                        {
                            double Mr = NEG, thrR = NEG, mT = NEG;
                            int par = 0;
                            const double hi_keep = bhi - 2.0 * tolp;  // (the last projection of the sort is the walk's default, :141)
                            asv_row_scan_pieces(
                                [&](int64_t off, double& p_, double& w_) {
                                    p_ = SP[off];
                                    w_ = SW[off];
                                },
                                nr2, nr1, c + rot0, last_block, tid,
                                [&](double, double lw, int o) { mT = fmax(mT, o < nr1 ? lw : NEG); },
                                [&](int) -> bool {
                                    thrR = asv_thr_neg(Mr, mx, (double)nr1, mb);  // (from the pieces in front of this one)
                                    double r = mT;
                                    for (int o2 = 1; o2 < 64; o2 <<= 1) r = fmax(r, __shfl_xor(r, o2));
                                    double* smx = sm + par * 8;
                                    if (lane == 0) smx[w] = r;
                                    if (tid == 0) smx[4] = (sh_sel[1] > lcap_c || sh_sel[2] > lcap_c) ? 1.0 : 0.0;
                                    __syncthreads();
                                    Mr = fmax(Mr, fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3])));
                                    mT = NEG;
                                    par ^= 1;
                                    return smx[4] != 0.0;
                                },
                                [&](double pr, double lw, int o) {
                                    const bool in = o < nr1;
                                    int b = (int)((pr - blo) * scale0);
                                    b = b < 0 ? 0 : (b > AT_NB - 1 ? AT_NB - 1 : b);
                                    if (in && lw >= thrR) {
                                        const int pos = atomicAdd(&sh_sel[1], 1);
                                        if (pos < lcap) giR[pos] = o;
                                    }
                                    if (in && (lw >= binthr[b] || pr >= hi_keep)) {
                                        const int pos = atomicAdd(&sh_sel[2], 1);
                                        if (pos < lcap) giS[pos] = o;
                                    }
                                });
                            __syncthreads();
                        }
                        bool ok = sh_sel[1] <= lcap_c && sh_sel[2] <= lcap_c;
                        // (d) the own batch: kept addends of totalprob2's and prob2's chains (one list: an addend that is a
                        // no-op for one of them is one in the re-run as well).  The cell's own first place in restrict2 (spos: the
                        // first code-3 element, found by the scan as it goes) splits the chains into their two regimes: elements in
                        // front of it are held against the largest log-weight of the pieces in front of theirs, the chain bounded with
                        // their own piece's included; those behind it against the largest OTHER log-weight of the pieces in front (the
                        // cell's later occurrences, log-weight 0, are left out of that bound: lower bounds may always leave out).  The
                        // pairs come as float codes (see SO): log-weights to 1e-6 relative (they are <= 0: x (1 + 1e-6) bounds one
                        // from below, x (1 - 1e-6) from above), the side of the cell's projection as decided by the stream.
                        if (ok) {
                            const double DN = 1.0 + 1e-6, UP = 1.0 - 1e-6;
                            const int last_own = (int)(Nown >> 6) - 1;
                            int spos = 0x7fffffff, mspos = 0x7fffffff;
                            double Tx = NEG, Px = NEG;      // totalprob2 / prob2: the largest other log-weight of the pieces gone by
                            double mTx = NEG, mPx = NEG;    // the same of the current piece
                            double thrTb = NEG, thrPb = NEG, thrTa = NEG, thrPa = NEG;
                            int par = 0;
                            asv_row_scan_pieces(
                                [&](int64_t off, double& p_, double& w_) {
                                    // (the scanner's offset is (block * 16 + slot) * 64 + j: the block and j are what is needed)
                                    const int64_t blk_ = off >> 10;
                                    const unsigned u = __float_as_uint(SO[((blk_ * 4 + (c & 3)) * 64 + (off & 63)) * 4 + (c >> 2)]);
                                    p_ = (double)(u & 3u);
                                    w_ = (double)__uint_as_float(u & ~3u);
                                },
                                0, nr2, c + rot0, last_own, tid,
                                [&](double code, double lw, int o) {
                                    const bool in = o < nr2;
                                    const bool other = in && code != 3.0;
                                    const bool sure = other && code >= 2.0;    // counted in prob2 whatever the rounding (:90-92)
                                    mTx = fmax(mTx, other ? lw : NEG);
                                    mPx = fmax(mPx, sure ? lw : NEG);
                                    mspos = in && code == 3.0 && o < mspos ? o : mspos;
                                },
                                [&](int end) -> bool {
                                    double r1 = mTx, r3 = mPx;
                                    int rs_ = mspos;
                                    for (int o2 = 1; o2 < 64; o2 <<= 1) {
                                        r1 = fmax(r1, __shfl_xor(r1, o2));
                                        r3 = fmax(r3, __shfl_xor(r3, o2));
                                        const int t2 = __shfl_xor(rs_, o2);
                                        rs_ = t2 < rs_ ? t2 : rs_;
                                    }
                                    double* smx = sm + par * 32;
                                    if (lane == 0) {
                                        smx[w * 4 + 0] = r1;
                                        smx[w * 4 + 1] = r3;
                                        smx[w * 4 + 2] = (double)rs_;
                                    }
                                    if (tid == 0) smx[16] = sh_sel[0] > lcap_c ? 1.0 : 0.0;
                                    __syncthreads();
                                    double pTx = NEG, pPx = NEG, ps = 2147483647.0;
                                    for (int ww = 0; ww < 4; ++ww) {
                                        pTx = fmax(pTx, smx[ww * 4 + 0]);
                                        pPx = fmax(pPx, smx[ww * 4 + 1]);
                                        ps = fmin(ps, smx[ww * 4 + 2]);
                                    }
                                    if (spos == 0x7fffffff) spos = (int)ps;
                                    const double cnt = (double)(end < nr2 ? end : nr2);
                                    // in front of the cell: lower bound from the pieces in front, upper bound with this piece's (all of
                                    // its others: an upper bound may always take in more)
                                    const double hiT = fmax(Tx, pTx) * UP;
                                    thrTb = asv_thr_neg(Tx * DN, hiT, cnt, mb);
                                    thrPb = asv_thr_neg(Px * DN, hiT, cnt, mb);  // (prob2's chain never exceeds totalprob2's)
                                    // behind it: the chains stand at log(1 + others)
                                    thrTa = asv_thr_self(Tx * DN, mb);
                                    thrPa = asv_thr_self(Px * DN, mb);
                                    Tx = fmax(Tx, pTx);
                                    Px = fmax(Px, pPx);
                                    mTx = mPx = NEG;
                                    par ^= 1;
                                    return smx[16] != 0.0;
                                },
                                [&](double code, double lw, int o) {
                                    const bool behind = o > spos;
                                    const double tT = behind ? thrTa : thrTb, tP = behind ? thrPa : thrPb;
                                    const double up = lw * UP;  // (the log-weight is at most this)
                                    if (o < nr2 && (code == 3.0 || up >= tT || (code >= 1.0 && up >= tP))) {
                                        const int pos = atomicAdd(&sh_sel[0], 1);
                                        if (pos < lcap) giO[pos] = o;
                                    }
                                });
                            __syncthreads();
                            ok = sh_sel[0] <= lcap_c;
                        }
                        if (ok) {
                            // (e) the kept addends, recomputed in the reference's order of operations
                            const int KO = sh_sel[0], KR = sh_sel[1], KS = sh_sel[2];
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // (the index lists went to global memory)
                            int* ib = reinterpret_cast<int*>(ub);
                            const double* cur = cx + c * GP;
                            const double* grd = cg + c * GP;
                            const int lsort = lcap < asv_tile_lsort(g) ? lcap : asv_tile_lsort(g);
                            // an index list ascending: in the LDS (-> ib) up to 4 lsort entries, else where it lies
                            auto sorted_ints = [&](int32_t* gi, int cnt) -> const int32_t* {
                                int np2 = 1;
                                while (np2 < cnt) np2 <<= 1;
                                if (np2 > 4 * lsort) {
                                    for (int i = cnt + tid; i < np2; i += T) gi[i] = 0x7fffffff;
                                    asv_sort_global(gi, np2, ib, 4 * lsort, tid, [](int x, int y) { return x < y; });
                                    return gi;
                                }
                                for (int i = tid; i < np2; i += T) ib[i] = i < cnt ? gi[i] : 0x7fffffff;
                                __syncthreads();
                                for (int k = 2; k <= np2; k <<= 1)
                                    for (int j = k >> 1; j > 0; j >>= 1) {
                                        for (int i = tid; i < np2; i += T) {
                                            const int ixj = i ^ j;
                                            if (ixj > i) {
                                                const int x0 = ib[i], x1 = ib[ixj];
                                                if (((i & k) == 0) ? x0 > x1 : x0 < x1) {
                                                    ib[i] = x1;
                                                    ib[ixj] = x0;
                                                }
                                            }
                                        }
                                        __syncthreads();
                                    }
                                return ib;
                            };
                            {
                                const int32_t* ix = sorted_ints(giR, KR);  // totalprob1's chain runs in restrict order (:117-131)
                                for (int i = tid; i < KR; i += T) {
                                    double pr, lw;
                                    asv_pair_literal(cur, grd, S + ((int64_t)nr2 + ix[i]) * gs_rt, g, sigma2, pr, lw);
                                    LR[(int64_t)c * lcap + i] = lw;
                                }
                                __syncthreads();
                            }
                            {
                                const int32_t* ix = sorted_ints(giO, KO);  // totalprob2's and prob2's likewise (:78-109)
                                for (int i = tid; i < KO; i += T) {
                                    const int j = ix[i];
                                    double pr = 0.0, lw = 0.0;
                                    bool add = true;
                                    if (sid[j] != c0 + c) {  // :84
                                        asv_pair_literal(cur, grd, S + (int64_t)j * gs_rt, g, sigma2, pr, lw);
                                        add = !(pr > curproj);  // :90-92
                                    }
                                    LO[((int64_t)c * lcap + i) * 2] = lw;
                                    LO[((int64_t)c * lcap + i) * 2 + 1] = add ? 1.0 : 0.0;
                                }
                                __syncthreads();
                            }
                            // the walk's chain runs over the (projection, log-weight) pairs as std::sort orders them (:134)
                            int np2 = 1;
                            while (np2 < KS) np2 <<= 1;
                            if (np2 <= lsort) {
                                double* kp = ub;
                                double* kw = ub + lsort;
                                for (int i = tid; i < np2; i += T) {
                                    double pr = POS, lw = POS;
                                    if (i < KS) asv_pair_literal(cur, grd, S + ((int64_t)nr2 + giS[i]) * gs_rt, g, sigma2, pr, lw);
                                    kp[i] = pr;
                                    kw[i] = lw;
                                }
                                __syncthreads();
                                for (int k = 2; k <= np2; k <<= 1)
                                    for (int j = k >> 1; j > 0; j >>= 1) {
                                        for (int i = tid; i < np2; i += T) {
                                            const int ixj = i ^ j;
                                            if (ixj > i) {
                                                const double p0 = kp[i], w0 = kw[i], p1 = kp[ixj], w1_ = kw[ixj];
                                                const bool up = (i & k) == 0;
                                                if (up ? pair_less(p1, w1_, p0, w0) : pair_less(p0, w0, p1, w1_)) {
                                                    kp[i] = p1;
                                                    kw[i] = w1_;
                                                    kp[ixj] = p0;
                                                    kw[ixj] = w0;
                                                }
                                            }
                                        }
                                        __syncthreads();
                                    }
                                for (int i = tid; i < KS; i += T) {
                                    LS[((int64_t)c * lcap + i) * 2] = kp[i];
                                    LS[((int64_t)c * lcap + i) * 2 + 1] = kw[i];
                                }
                            } else {  // a long list: sorted where it lies, through an LDS window
                                typedef double d2a __attribute__((ext_vector_type(2)));
                                d2a* L2 = reinterpret_cast<d2a*>(LS + (int64_t)c * lcap * 2);
                                for (int i = tid; i < np2; i += T) {
                                    double pr = POS, lw = POS;
                                    if (i < KS) asv_pair_literal(cur, grd, S + ((int64_t)nr2 + giS[i]) * gs_rt, g, sigma2, pr, lw);
                                    L2[i] = d2a{pr, lw};
                                }
                                asv_sort_global(L2, np2, reinterpret_cast<d2a*>(ub), lsort, tid,
                                                [](const d2a& x, const d2a& y) { return pair_less(x[0], x[1], y[0], y[1]); });
                            }
                            if (tid == 0) {
                                sh_K[c][0] = KO;
                                sh_K[c][1] = KR;
                                sh_K[c][2] = KS;
                            }
                            literal = true;
                            ++n_lit;
                            tk_lit_n += (unsigned long long)KO + KR + KS;
                            __syncthreads();
                        }
                        tk_lit += __builtin_amdgcn_s_memrealtime() - tkl0;
                    }
                    if (!literal) ++n_back;
                    cell_mode = literal ? 1 : 2;
                }
                if (!literal)
                for (int round = 0; round < 40; ++round) {
                    if (round > 0) {  // (round 0's histogram is the one taken above)
                        for (int b = tid; b < AT_NB; b += T) hist[b] = 0ull;
                        if (tid == 0) sh_cnt = 0;
                        __syncthreads();
                    }
                    const double scale = bhi > blo ? (double)AT_NB / (bhi - blo) : 0.0;
                    if (round > 0) {
                        asv_row_scan<true>(SP, SW, nr2, nr1, c + rot0, last_block, tid, [&](double pr, double lw, int o) {
                            const bool in = o < nr1 && pr >= blo && pr <= bhi;
                            int b = (int)((pr - blo) * scale);
                            b = b < 0 ? 0 : (b > AT_NB - 1 ? AT_NB - 1 : b);
                            const double wv = exp(lw - mx) * FIX;
                            // (an element out of play adds nothing to a bin of the thread's own: no branch, no pile-up on one bin)
                            atomicAdd(&hist[in ? b : (tid & (AT_NB - 1))], in ? (unsigned long long)wv : 0ull);
                        });
                        __syncthreads();
                    }
                    const double ep2 = exp(prob2);
                    const int at = asv_first_crossing(
                        hist, AT_NB, before,
                        [&](unsigned long long tot) { return round == 0 ? ep2 * (double)tot : target; },  // the target (:137), fixed-point units
                        reinterpret_cast<unsigned long long*>(sm), &sh_bin, &sh_before, &target);
                    if (at < 0) {  // no prefix reaches the target: the last projection (:141)
                        ref_quan = sc_hi[c];
                        break;
                    }
                    before = sh_before;
                    // the projections that fall into bin `at`
                    asv_row_scan<false>(SP, SW, nr2, nr1, c + rot0, last_block, tid, [&](double pr, double, int o) {
                        int b = (int)((pr - blo) * scale);
                        b = b < 0 ? 0 : (b > AT_NB - 1 ? AT_NB - 1 : b);
                        if (!(o < nr1 && pr >= blo && pr <= bhi && b == at)) return;  // (rare: one bin of 2 048)
                        const int pos = atomicAdd(&sh_cnt, 1);
                        if (pos < AT_CAP) {
                            lp[pos] = pr;
                            lw_[pos] = (unsigned long long)(exp(w1(o) - mx) * FIX);
                        }
                    });
                    __syncthreads();
                    const int cnt = sh_cnt;
                    if (cnt <= AT_CAP || !(bhi > blo)) {
                        const int m = cnt < AT_CAP ? cnt : AT_CAP;
                        int npad = 1;
                        while (npad < m) npad <<= 1;
                        for (int i = m + tid; i < npad; i += T) {
                            lp[i] = POS;
                            lw_[i] = 0ull;
                        }
                        __syncthreads();
                        for (int k = 2; k <= npad; k <<= 1)
                            for (int j = k >> 1; j > 0; j >>= 1) {
                                for (int i = tid; i < npad; i += T) {
                                    const int ixj = i ^ j;
                                    if (ixj > i) {
                                        const double a = lp[i], b2 = lp[ixj];
                                        const unsigned long long wa = lw_[i], wb = lw_[ixj];
                                        const bool up = (i & k) == 0;
                                        // (projection, weight) ascending: equal projections in a fixed order
                                        const bool gt = a > b2 || (a == b2 && wa > wb);
                                        const bool lt = a < b2 || (a == b2 && wa < wb);
                                        if (up ? gt : lt) {
                                            lp[i] = b2;
                                            lp[ixj] = a;
                                            lw_[i] = wb;
                                            lw_[ixj] = wa;
                                        }
                                    }
                                }
                                __syncthreads();
                            }
                        double tg2;
                        const int hit = asv_first_crossing(
                            lw_, m, before, [&](unsigned long long) { return target; },
                            reinterpret_cast<unsigned long long*>(sm), &sh_bin, &sh_before, &tg2);
                        if (tid == 0) sc_tmp[1] = hit >= 0 ? lp[hit] : (m > 0 ? lp[m - 1] : bhi);
                        __syncthreads();
                        ref_quan = sc_tmp[1];
                        break;
                    }
                    // still too many in one bin: its range becomes the whole histogram
                    const double nlo = blo + (double)at / scale, nhi = blo + (double)(at + 1) / scale;
                    // (rounding of the bin edges: widen by an ulp-ish margin and keep inside the old range; entries that
                    // fall outside the bin but inside the widened range are binned again, which is harmless -- except that
                    // the weight below the range must then not contain them: recompute `before` as the weight below nlo)
                    const double margin = 4e-16 * fmax(fabs(blo), fabs(bhi));
                    blo = fmax(blo, nlo - margin);
                    bhi = fmin(bhi, nhi + margin);
                    ref_quan = bhi;  // (provisional: the quantile lies in [blo, bhi]; final unless another round refines it)
                    unsigned long long mine = 0;
                    asv_row_scan<true>(SP, SW, nr2, nr1, c + rot0, last_block, tid, [&](double pr, double lw, int o) {
                        const double wv = exp(lw - mx) * FIX;
                        mine += o < nr1 && pr < blo ? (unsigned long long)wv : 0ull;
                    });
                    // integer sums: any order gives the same total
                    __syncthreads();
                    unsigned long long* smu = reinterpret_cast<unsigned long long*>(sm);
                    smu[tid] = mine;
                    __syncthreads();
                    for (int o2 = T / 2; o2 > 0; o2 >>= 1) {
                        if (tid < o2) smu[tid] += smu[tid + o2];
                        __syncthreads();
                    }
                    before = smu[0];
                    __syncthreads();
                }
            }
            if (!literal && tid == 0) out[c0 + c] = (ref_quan - curproj) / l2;  // :160
            // (testing hook "asv_modes": which way every cell went, one byte per cell behind the tallies)
            if (tid == 0 && tally && (unsigned long long)(c0 + c) < tally[3])
                reinterpret_cast<unsigned char*>(tally + 4)[c0 + c] = (unsigned char)cell_mode;
            __syncthreads();
        }
        // ---- the literal re-run's chains: one lane per chain, the reference's sequential sums (:96-109, :127-131), then
        // one lane per cell for the walk (:137-157)
        if (lit_on) {
            const unsigned long long tkc0 = __builtin_amdgcn_s_memrealtime();
            const bool any_lit = n_lit > 0;  // (thread 0's count is the tile's)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // (the lists were written by other threads of the block)
            const double* LO = SW + (int64_t)AT_C * Npad;
            const double* LR = LO + (int64_t)AT_C * lcap * 2;
            const double* LS = LR + (int64_t)AT_C * lcap;
            // (one WAVE per kind of sum -- total, counted, reference, walk --, a lane per cell: lanes of one wave on different
            // kinds would take their loops one after the other)
            static_assert(T == 256, "the chains take the block's four waves");
            if (lane < AT_C) {
                const int c = lane, which = w;
                const int KO = sh_K[c][0];
                if (KO >= 0) {
                    double acc = 0.0;
                    bool first = true;
                    // (a chain is ONE lane's sequential sum: its addends are asked for eight at a time, ahead of the adds that use
                    // them -- one by one, a step of the chain was a global load's round trip plus the add: 0.95 us, 59 ms for a
                    // cell of config 5 at sigma 1 with 94 000 kept addends, the launch waiting for it)
                    if (which < 2) {
                        const double* L = LO + (int64_t)c * lcap * 2;
                        int i = 0;
                        for (; i + 8 <= KO; i += 8) {
                            double lw[8], fl[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                lw[u] = L[2 * (i + u)];
                                fl[u] = L[2 * (i + u) + 1];
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (which == 0 || fl[u] != 0.0) {
                                    acc = first ? lw[u] : bmx_pm_logspace_add(acc, lw[u]);
                                    first = false;
                                }
                        }
                        for (; i < KO; ++i) {
                            const double lw = L[2 * i];
                            if (which == 0 || L[2 * i + 1] != 0.0) {
                                acc = first ? lw : bmx_pm_logspace_add(acc, lw);
                                first = false;
                            }
                        }
                    } else if (which == 3) {
                        // the walk's running sums (:147-151) do not depend on where the walk stops: a fourth lane takes them
                        // WHILE the three chains run (the walk used to start when they were through: twice the sequential
                        // length) and leaves sum i in the place of log-weight i; they never decrease (a log-sum is at least its
                        // larger term), so the walk's stop is then a binary search for the target
                        double* L = const_cast<double*>(LS) + (int64_t)c * lcap * 2;
                        const int KS = sh_K[c][2];
                        double cum = 0.0;
                        int i = 0;
                        for (; i + 8 <= KS; i += 8) {
                            double lw[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) lw[u] = L[2 * (i + u) + 1];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                cum = i + u == 0 ? lw[u] : bmx_pm_logspace_add(cum, lw[u]);
                                lw[u] = cum;
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) L[2 * (i + u) + 1] = lw[u];
                        }
                        for (; i < KS; ++i) {
                            cum = i == 0 ? L[1] : bmx_pm_logspace_add(cum, L[2 * i + 1]);
                            L[2 * i + 1] = cum;
                        }
                    } else {
                        const double* L = LR + (int64_t)c * lcap;
                        const int KR = sh_K[c][1];
                        int i = 0;
                        for (; i + 8 <= KR; i += 8) {
                            double lw[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) lw[u] = L[i + u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                acc = first ? lw[u] : bmx_pm_logspace_add(acc, lw[u]);
                                first = false;
                            }
                        }
                        for (; i < KR; ++i) {
                            acc = first ? L[i] : bmx_pm_logspace_add(acc, L[i]);
                            first = false;
                        }
                    }
                    if (which < 3) sh_chain[c][which] = acc;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // (the walk's sums, written by one wave, read by another)
            __syncthreads();
            if (tid < AT_C && c0 + tid < n2 && sh_K[tid][0] >= 0) {
                const int c = tid, KS = sh_K[c][2];
                const double* L = LS + (int64_t)c * lcap * 2;
                const double tgt = (sh_chain[c][1] - sh_chain[c][0]) + sh_chain[c][2];  // :111, :138
                // the first place whose running sum reaches the target (:147-151; the sums stand where the log-weights stood and
                // never decrease), the last projection if none does (:141; a NaN target: none)
                int lo_ = 0, hi_ = KS;
                while (lo_ < hi_) {
                    const int mid = (lo_ + hi_) >> 1;
                    if (L[2 * mid + 1] >= tgt) hi_ = mid;
                    else lo_ = mid + 1;
                }
                const double rq = L[2 * (lo_ < KS ? lo_ : KS - 1)];
                out[c0 + c] = (rq - sc_proj[c]) / sc_l2[c];  // :160
            }
            if (any_lit) {
                tk_chain += __builtin_amdgcn_s_memrealtime() - tkc0;
                ++tk_chain_tiles;
            }
            if (tid == 0 && tally) {
                atomicAdd(&tally[0], (unsigned long long)n_lit);
                atomicAdd(&tally[1], (unsigned long long)n_back);
                atomicAdd(&tally[2], (unsigned long long)(n2 - c0 < AT_C ? n2 - c0 : AT_C));
            }
            __syncthreads();
        }
        tk_cells += __builtin_amdgcn_s_memrealtime() - tk0;
        if (!(tile_ctr && !gbar)) tile += gridDim.x;  // (the fixed deal)
    }
    if (tid == 0) {
        atomicAdd(&g_asv_ticks[0], tk_stream);
        atomicAdd(&g_asv_ticks[1], tk_wait);
        atomicAdd(&g_asv_ticks[2], tk_cells);
        if (tk_lit | tk_chain) {
            atomicAdd(&g_asv_ticks[4], tk_lit);
            atomicAdd(&g_asv_ticks[5], tk_chain);
            atomicAdd(&g_asv_ticks[6], tk_lit_n);
            atomicAdd(&g_asv_ticks[7], tk_chain_tiles);
        }
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
    const int nkb = gd <= 128 ? (gd + SG_KC - 1) / SG_KC : 0;
#define BMX_SGK(NKB)                                                                                                              \
    case NKB:                                                                                                                     \
        hipLaunchKernelGGL((sgk_flash<false, 64, NKB>), dim3((unsigned)cdiv(U, 64), 1), dim3(256), 0, stream, mat, gd, xn2, index,  \
                           (int64_t)U, index, U, 1.0 / sigma2, nullptr, nullptr, 0, nullptr, dens);                               \
        if (g <= 64)                                                                                                              \
            hipLaunchKernelGGL((sgk_flash<true, 64, NKB>), dim3((unsigned)cdiv(n, 64), 1), dim3(256), 0, stream, mat, gd, xn2,      \
                               nullptr, (int64_t)n, index, U, 1.0 / sigma2, dens, averaged, g, out, nullptr);                     \
        else                                                                                                                      \
            hipLaunchKernelGGL((sgk_flash<true, 128, NKB>), dim3((unsigned)cdiv(n, 64), (unsigned)cdiv(g, 128)), dim3(256), 0,     \
                               stream, mat, gd, xn2, nullptr, (int64_t)n, index, U, 1.0 / sigma2, dens, averaged, g, out, nullptr); \
        break
    switch (nkb) {
        BMX_SGK(1);
        BMX_SGK(2);
        BMX_SGK(3);
        BMX_SGK(4);
        default:
            BMX_SGK(0);
    }
#undef BMX_SGK
    BMX_LAUNCH_CHECK();
}

// Which form runs and what it needs: a PURE function of the sizes (and of the testing hook "asv_fast"), so that the
// caller's allocation and the launch agree whatever happens to the free memory in between.  exact = 1: asv_exact_kernel
// (bit-exact walk; up to 131 072 restricted cells and 4e7 pairs); exact = 0: the tiled FP64-MFMA form, `blocks` workgroups
// with 2 x 16 x (nr1 + nr2) doubles of scratch each, behind them (`extra_doubles`) the gathered stream, the squared norms
// of its cells and -- vect handed over column-major -- a row-major copy of vect.
// Counters of the tiled form, per device, for tests and bench.py (bmx_dev_get "asv_literal_cells" / "asv_fallback_cells" /
// "asv_tiled_cells"): cells re-run literally, flagged cells that went the histogram way (more than lcap kept addends),
// all cells the tiled form has handled.  Added up until "asv_tally_reset".
// Testing hook "asv_modes" = n: the tiled form also records which way each of the first n cells of a call went (one byte
// per cell behind the four tally words: 0 the histogram way, 1 re-run literally, 2 flagged but beyond the re-run); the LAST
// call's bytes stay (bmx_dev_get_bytes "asv_modes").
static unsigned long long* g_tally[64] = {};
static size_t g_tally_modes[64] = {};
static long long g_tally_cap_set[64] = {};  // what word 3 holds + 1 (0: unknown)
static std::mutex g_tally_mu;  // (several host threads may drive engines on one device: the buffer is made and regrown under it)
unsigned long long* asv_tally_device() {
    int dev = 0;
    BMX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_tally_mu);
    const size_t want = (size_t)std::max(0, dev_knobs().asv_modes);
    if (!g_tally[dev] || g_tally_modes[dev] < want) {
        unsigned long long keep[4] = {0, 0, 0, 0};
        if (g_tally[dev]) {
            BMX_HIP(hipDeviceSynchronize());
            BMX_HIP(hipMemcpy(keep, g_tally[dev], 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            (void)hipFree(g_tally[dev]);
            g_tally[dev] = nullptr;
        }
        BMX_HIP(hipMalloc(reinterpret_cast<void**>(&g_tally[dev]), 4 * sizeof(unsigned long long) + want + 8));
        BMX_HIP(hipMemset(g_tally[dev], 0, 4 * sizeof(unsigned long long) + want + 8));
        BMX_HIP(hipMemcpy(g_tally[dev], keep, 3 * sizeof(unsigned long long), hipMemcpyHostToDevice));
        g_tally_modes[dev] = want;
        g_tally_cap_set[dev] = 0;
    }
    if (g_tally_cap_set[dev] != (long long)want + 1) {  // (only when the hook changes: a blocking copy otherwise never happens)
        const unsigned long long cap = (unsigned long long)want;  // (0: no modes recorded)
        BMX_HIP(hipMemcpy(g_tally[dev] + 3, &cap, sizeof(cap), hipMemcpyHostToDevice));
        g_tally_cap_set[dev] = (long long)want + 1;
    }
    return g_tally[dev];
}
void asv_modes_read(unsigned char* dst, size_t n) {
    unsigned long long* t = asv_tally_device();
    int dev = 0;
    BMX_HIP(hipGetDevice(&dev));
    if (!t || dev < 0 || dev >= 64) {  // (a device beyond the table: nothing was recorded)
        for (size_t i = 0; i < n; ++i) dst[i] = 255;
        return;
    }
    BMX_HIP(hipDeviceSynchronize());
    const size_t have = std::min(n, g_tally_modes[dev]);
    if (have) BMX_HIP(hipMemcpy(dst, t + 4, have, hipMemcpyDeviceToHost));
    for (size_t i = have; i < n; ++i) dst[i] = 255;
}
void asv_tally_read(unsigned long long out[3], bool reset) {
    unsigned long long* t = asv_tally_device();
    out[0] = out[1] = out[2] = 0;
    if (!t) return;
    BMX_HIP(hipDeviceSynchronize());
    BMX_HIP(hipMemcpy(out, t, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) {
        BMX_HIP(hipMemset(t, 0, 3 * sizeof(unsigned long long)));
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        BMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_asv_ticks), z, sizeof(z)));
    }
}
void asv_ticks_read(unsigned long long out[8]) {
    unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    BMX_HIP(hipDeviceSynchronize());
    BMX_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_asv_ticks), sizeof(v)));
    for (int i = 0; i < 8; ++i) out[i] = v[i];
}

static int device_cu_count() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus[dev] = n > 0 ? n : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}

AsvPlan adjust_shift_variance_plan(int g, int n2, int nr1, int nr2, int vect_row_major) {
    AsvPlan pl;
    // the bit-exact form's sequential log-sum chains cost ~1.6 ns per (cell, restricted cell) pair, the tiled form 0.015:
    // exact up to 4e7 pairs (the reference's own test shapes and anything a test can check against the CPU), tiled beyond
    pl.exact = !dev_knobs().asv_fast && (int64_t)nr1 + nr2 <= 131072 && (double)std::max(n2, 1) * ((double)nr1 + nr2) <= 4e7;
    int p = 1;
    while (p < std::max(nr1, 1)) p <<= 1;
    pl.npad = p;
    if (pl.exact) {
        const size_t per_block = 2 * (size_t)nr2 + 2 * (size_t)p;
        const size_t budget = (size_t)1 << 27;  // doubles: 1 GiB of scratch at most
        pl.blocks = (int)std::max<size_t>(1, std::min<size_t>({(size_t)std::max(n2, 1), (size_t)1024, budget / std::max<size_t>(per_block, 1)}));
        pl.main_doubles = per_block * (size_t)pl.blocks;
        pl.extra_doubles = 16;
        return pl;
    }
    const size_t N = asv_tile_npad((size_t)nr1 + (size_t)nr2);  // (padded to whole pairs of steps of the stream)
    // the literal re-run of flagged cells keeps at most lcap addends per chain (testing hook "asv_cap": 0 = off, n = at most n)
    pl.lcap = asv_tile_lcap_default(g);
    {
        // a chain keeps addends of ONE batch's restricted cells: lists longer than the larger batch (as a power of two) are
        // never filled, and at 85 MB of list scratch per workgroup a call of a few thousand cells would ask for 22 GB
        int need = 1;
        while (need < std::max({nr1, nr2, 1})) need <<= 1;
        pl.lcap = std::min(pl.lcap, need);
    }
    if (dev_knobs().asv_cap >= 0) {  // (rounded down to a power of two: the lists are padded to one)
        int c2 = 0;
        for (int b = 1; b > 0 && b <= dev_knobs().asv_cap; b <<= 1) c2 = b;
        pl.lcap = std::min(pl.lcap, c2);
    }
    const size_t per_block = (size_t)2 * AT_C * N + (size_t)asv_tile_list_doubles(pl.lcap) +
                             (pl.lcap > 0 ? (size_t)AT_C * asv_tile_npad((size_t)nr2) / 2 : 0);  // (+ the own batch's float codes)
    const size_t budget = (size_t)12 << 30;  // doubles: 96 GiB of the 288 at most
    const size_t tiles = ((size_t)std::max(n2, 1) + AT_C - 1) / AT_C;
    pl.blocks = (int)std::max<size_t>(1, std::min<size_t>({tiles, (size_t)256, budget / per_block}));
    pl.main_doubles = per_block * (size_t)pl.blocks;
    // the nr1 + nr2 streamed cells: their rows, norms, ids; a row-major copy of vect if it came column-major
    pl.extra_doubles = N * asv_tile_gs(g) + N + 1 + (N + 1) / 2 + 2 + (vect_row_major ? 0 : (size_t)n2 * g) + 16;
    return pl;
}

void adjust_shift_variance_device(hipStream_t stream, const double* data1, int g, int n1, const double* data2, int n2,
                                  const double* vect, double sigma2, const int32_t* restrict1, int nr1,
                                  const int32_t* restrict2, int nr2, double* out, double* ws_pairs, const AsvPlan& pl,
                                  int vect_row_major, int cell_begin, int cell_end) {
    // vect is an R matrix [n2 x g] (column-major) at the .Call boundary, row-major [n2][g] inside the engine
    const int64_t vs_cell = vect_row_major ? g : 1, vs_x = vect_row_major ? 1 : n2;
    (void)n1;
    if (cell_end < 0) cell_end = n2;
    if (n2 <= 0 || cell_end <= cell_begin) return;
    const int blocks = pl.blocks;
    if (pl.exact) {
        hipLaunchKernelGGL(asv_exact_kernel, dim3(blocks), dim3(T), (size_t)2 * g * sizeof(double), stream, data1, g, data2,
                           n2, vect, vs_cell, vs_x, sigma2, restrict1, nr1, restrict2, nr2, pl.npad, out, ws_pairs, cell_begin,
                           cell_end);
    } else {
        if (g > 256) throw Error(BMX_ERR_ARG, "adjust_shift_variance: more than 256 dimensions at this size are not supported");
        double* extra = ws_pairs + pl.main_doubles;
        const int64_t N = (int64_t)asv_tile_npad((size_t)nr1 + (size_t)nr2);
        const int gs = asv_tile_gs(g);
        double* S = extra;                      // [N][gs] the streamed cells, zero rows up to whole pairs of steps
        double* snrm = S + N * gs;              // [N] + their maximum
        int32_t* sid = reinterpret_cast<int32_t*>(snrm + N + 1);  // [N]
        const double* vrm = vect;
        if (!vect_row_major) {
            double* t = snrm + N + 1 + (N + 1) / 2 + 2;
            transpose_cm_to_rm(stream, vect, n2, g, t);
            vrm = t;
        }
        BMX_HIP(hipMemsetAsync(snrm + N, 0, sizeof(double), stream));
        hipLaunchKernelGGL(asv_gather_stream, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, stream, data1, data2, g, gs, restrict1,
                           nr1, restrict2, nr2, N, S, snrm, sid);
        // the round barrier of the tile kernel needs every workgroup of the launch resident at once: one workgroup per CU is
        // always possible (its LDS and registers fit a CU by construction), so the launch must not be wider than the device
        // (testing hook "asv_sync" = 0: no barrier, round 5's free-running tiles)
        unsigned int* gbar = nullptr;
        if (dev_knobs().asv_sync != 0 && blocks <= device_cu_count())
            gbar = reinterpret_cast<unsigned int*>(extra + pl.extra_doubles - 2);
        unsigned int* tile_ctr = reinterpret_cast<unsigned int*>(extra + pl.extra_doubles - 4);
        hipLaunchKernelGGL(asv_max_norm, dim3(256), dim3(256), 0, stream, snrm, N, gbar, tile_ctr);
        const size_t lds = asv_tile_lds_bytes(g, pl.lcap);
        unsigned long long* tally = asv_tally_device();
#define BMX_ASV_TILE(NB8)                                                                                                    \
    case NB8:                                                                                                                \
        ensure_dynamic_lds(reinterpret_cast<const void*>(&asv_tile_kernel<NB8>), lds);                                       \
        hipLaunchKernelGGL(asv_tile_kernel<NB8>, dim3(blocks), dim3(T), lds, stream, g, data2, n2, vrm, sigma2, nr1, nr2,     \
                           (const double*)S, (const double*)snrm, (const int32_t*)sid, out, ws_pairs, cell_begin, cell_end,  \
                           pl.lcap, tally, gbar, tile_ctr);                                                                \
        break
        switch (asv_tile_nb8(g)) {
#ifdef BMX_ASV_AB_ONLY13
            BMX_ASV_TILE(13);
#else
            BMX_ASV_TILE(1);
            BMX_ASV_TILE(2);
            BMX_ASV_TILE(3);
            BMX_ASV_TILE(4);
            BMX_ASV_TILE(5);
            BMX_ASV_TILE(6);
            BMX_ASV_TILE(7);
            BMX_ASV_TILE(8);
            BMX_ASV_TILE(9);
            BMX_ASV_TILE(10);
            BMX_ASV_TILE(11);
            BMX_ASV_TILE(12);
            BMX_ASV_TILE(13);
            BMX_ASV_TILE(14);
            BMX_ASV_TILE(15);
            BMX_ASV_TILE(16);
#endif
            default:
                BMX_ASV_TILE(0);
        }
#undef BMX_ASV_TILE
    }
    BMX_LAUNCH_CHECK();
}

}  // namespace bmx
