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
// the sequential walk), with the bit-reproducible exp / log1p of portable_math.hpp: bit for bit the CPU oracle's
// result.  Its sequential chains cost O(cells x (nr1 + nr2)) dependent steps, fine up to ~1e5 restricted cells;
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

// density[i] = log sum_j exp(-|x_idx[i] - x_idx[j]|^2 / sigma2)     (smooth_gaussian_kernel.cpp:56-65)
__global__ __launch_bounds__(T) void sgk_density(const double* __restrict__ mat, int gd, const int32_t* __restrict__ index,
                                                 int U, double sigma2, double* __restrict__ density) {
    __shared__ double sm[T];
    const int i = blockIdx.x;
    const double* ci = mat + (int64_t)index[i] * gd;
    double mx = -__builtin_inf();
    for (int j = threadIdx.x; j < U; j += T) {
        const double* cj = mat + (int64_t)index[j] * gd;
        double s = 0.0;
        for (int x = 0; x < gd; ++x) {
            const double t = ci[x] - cj[x];
            s += t * t;
        }
        mx = fmax(mx, s / -sigma2);
    }
    mx = block_max(mx, sm);
    double acc = 0.0;
    for (int j = threadIdx.x; j < U; j += T) {
        const double* cj = mat + (int64_t)index[j] * gd;
        double s = 0.0;
        for (int x = 0; x < gd; ++x) {
            const double t = ci[x] - cj[x];
            s += t * t;
        }
        acc += exp(s / -sigma2 - mx);
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) density[i] = mx + log(acc);
}

// out[:, c] = sum_i averaged[:, i] * softmax_i(logw_i[c] - density_i)      (smooth_gaussian_kernel.cpp:70-115)
// grid (cells, gene blocks of T): every gene block re-derives the weights of its cell
constexpr int SGK_CHUNK = 1024;
__global__ __launch_bounds__(T) void sgk_apply(const double* __restrict__ averaged, int g, int U,
                                               const int32_t* __restrict__ index, const double* __restrict__ mat, int gd,
                                               int n, double sigma2, const double* __restrict__ density,
                                               double* __restrict__ out) {
    __shared__ double sm[T];
    __shared__ double lm[SGK_CHUNK];
    const int c = blockIdx.x;
    const int x = blockIdx.y * T + threadIdx.x;
    const double* cc = mat + (int64_t)c * gd;
    double M = -__builtin_inf(), Tsum = 0.0, acc = 0.0;
    for (int i0 = 0; i0 < U; i0 += SGK_CHUNK) {
        const int m = min(SGK_CHUNK, U - i0);
        double cmx = -__builtin_inf();
        for (int i = threadIdx.x; i < m; i += T) {
            const double* ci = mat + (int64_t)index[i0 + i] * gd;
            double s = 0.0;
            for (int t = 0; t < gd; ++t) {
                const double df = ci[t] - cc[t];
                s += df * df;
            }
            const double v = s / -sigma2 - density[i0 + i];
            lm[i] = v;
            cmx = fmax(cmx, v);
        }
        cmx = block_max(cmx, sm);
        const double newM = fmax(M, cmx);
        const double scale = M == -__builtin_inf() ? 0.0 : exp(M - newM);
        double part = 0.0;
        for (int i = threadIdx.x; i < m; i += T) {
            const double w = newM == -__builtin_inf() ? 0.0 : exp(lm[i] - newM);
            lm[i] = w;
            part += w;
        }
        part = block_sum(part, sm);  // also orders the lm[] writes before the reads below
        Tsum = Tsum * scale + part;
        acc *= scale;
        if (x < g)
            for (int i = 0; i < m; ++i) acc += averaged[(int64_t)(i0 + i) * g + x] * lm[i];
        M = newM;
        __syncthreads();
    }
    if (x < g) out[(int64_t)c * g + x] = acc / Tsum;
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

void smooth_gaussian_kernel_device(hipStream_t stream, const double* averaged, int g, int U, const int32_t* index,
                                   const double* mat, int gd, int n, double sigma2, double* out, double* ws_density) {
    if (n <= 0 || g <= 0) return;
    if (U > 0) {
        hipLaunchKernelGGL(sgk_density, dim3(U), dim3(T), 0, stream, mat, gd, index, U, sigma2, ws_density);
        BMX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sgk_apply, dim3(n, cdiv(g, T)), dim3(T), 0, stream, averaged, g, U, index, mat, gd, n, sigma2,
                       ws_density, out);
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
