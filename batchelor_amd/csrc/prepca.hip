// The two streaming steps directly upstream of the merge engine in fastMNN():
//   cosineNorm        (R/cosineNorm.R:63-82)      l2[c] = sqrt(sum_g x[g,c]^2), columns divided by pmax(1e-8, l2)
//   PCA projection    (R/multiBatchPCA.R:236-239) crossprod(x - centers, u)
// fused into ONE pass over the genes x cells matrix:   out[c, j] = (sum_g x[g,c] u[g,j]) / L_c - sum_g centers[g] u[g,j]
// with L_c = max(1e-8, l2[c]) (or 1 without cosine normalisation).  x is read exactly once; FP64 throughout.
#include "bmx_ops.hpp"

namespace bmx {
namespace {

constexpr int CB = 16;  // cells per workgroup
constexpr int GT = 64;  // genes per staged tile

__global__ __launch_bounds__(256) void colnorm_kernel(const double* __restrict__ x, int G, int n,
                                                      double* __restrict__ l2) {
    // one wave per cell (column): the column is contiguous
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= n) return;
    const double* col = x + (int64_t)c * G;
    double s = 0.0;
    for (int g = lane; g < G; g += 64) s += col[g] * col[g];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) l2[c] = sqrt(s);
}

__global__ void apply_cosnorm_kernel(const double* __restrict__ x, int G, int n, const double* __restrict__ l2,
                                     double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)G * n) return;
    const double L = l2[e / G];
    out[e] = x[e] / (L < 1e-8 ? 1e-8 : L);  // pmax(1e-8, l2)
}

__global__ __launch_bounds__(256) void cosnorm_project_kernel(const double* __restrict__ x, int G, int n,
                                                              const double* __restrict__ u, int d,
                                                              const double* __restrict__ cu, int cos_norm,
                                                              double* __restrict__ out, double* __restrict__ l2_out) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    double* xs = reinterpret_cast<double*>(smem_p);  // [CB][GT + 1]
    double* us = xs + CB * (GT + 1);                 // [GT][d]
    const int tid = threadIdx.x;
    const int c_local = tid >> 4, jg = tid & 15;
    const int c0 = blockIdx.x * CB;
    const int c = c0 + c_local;
    constexpr int JMAX = 16;  // d <= 256
    double acc[JMAX];
#pragma unroll
    for (int t = 0; t < JMAX; ++t) acc[t] = 0.0;
    double s2 = 0.0;
    for (int g0 = 0; g0 < G; g0 += GT) {
        const int gn = min(GT, G - g0);
        for (int e = tid; e < CB * GT; e += 256) {
            const int cc = e / GT, gg = e - cc * GT;
            xs[cc * (GT + 1) + gg] = (c0 + cc < n && gg < gn) ? x[(int64_t)(c0 + cc) * G + g0 + gg] : 0.0;
        }
        for (int e = tid; e < GT * d; e += 256) {
            const int j = e / GT, gg = e - j * GT;
            us[gg * d + j] = gg < gn ? u[(int64_t)j * G + g0 + gg] : 0.0;
        }
        __syncthreads();
        const double* xr = xs + c_local * (GT + 1);
        for (int gg = 0; gg < GT; ++gg) {
            const double xv = xr[gg];
            if (jg == 0) s2 += xv * xv;
#pragma unroll
            for (int t = 0; t < JMAX; ++t) {
                const int j = jg + 16 * t;
                if (j < d) acc[t] += xv * us[gg * d + j];
            }
        }
        __syncthreads();
    }
    s2 = __shfl(s2, (tid & 63) & ~15);  // the jg == 0 lane of this cell's 16-lane group
    if (c >= n) return;
    const double l2 = sqrt(s2);
    const double L = cos_norm ? (l2 < 1e-8 ? 1e-8 : l2) : 1.0;
#pragma unroll
    for (int t = 0; t < JMAX; ++t) {
        const int j = jg + 16 * t;
        if (j < d) out[(int64_t)j * n + c] = acc[t] / L - cu[j];
    }
    if (jg == 0 && l2_out) l2_out[c] = l2;
}

__global__ void center_dot_kernel(const double* __restrict__ centers, const double* __restrict__ u, int G, int d,
                                  double* __restrict__ cu) {
    // cu[j] = sum_g centers[g] u[g, j]; one wave per j
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= d) return;
    double s = 0.0;
    for (int g = lane; g < G; g += 64) s += centers[g] * u[(int64_t)j * G + g];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) cu[j] = s;
}

}  // namespace

void cosine_l2_device(hipStream_t stream, const double* x, int G, int n, double* l2) {
    if (n <= 0) return;
    hipLaunchKernelGGL(colnorm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, stream, x, G, n, l2);
    BMX_LAUNCH_CHECK();
}

void apply_cosine_norm_device(hipStream_t stream, const double* x, int G, int n, const double* l2, double* out) {
    const int64_t total = (int64_t)G * n;
    if (total <= 0) return;
    hipLaunchKernelGGL(apply_cosnorm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, G, n, l2, out);
    BMX_LAUNCH_CHECK();
}

void cosnorm_project_device(hipStream_t stream, const double* x, int G, int n, const double* u, int d,
                            const double* centers, int cos_norm, double* out, double* l2_out, double* cu_scratch) {
    if (n <= 0) return;
    if (d > 256) throw Error(BMX_ERR_ARG, "more than 256 dimensions are not supported");
    hipLaunchKernelGGL(center_dot_kernel, dim3(cdiv(d, 4)), dim3(256), 0, stream, centers, u, G, d, cu_scratch);
    BMX_LAUNCH_CHECK();
    const size_t lds = ((size_t)CB * (GT + 1) + (size_t)GT * d) * sizeof(double);  // 139 KiB at d = 256
    ensure_dynamic_lds(reinterpret_cast<const void*>(&cosnorm_project_kernel), lds);
    hipLaunchKernelGGL(cosnorm_project_kernel, dim3(cdiv(n, CB)), dim3(256), lds, stream, x, G, n, u, d, cu_scratch,
                       cos_norm, out, l2_out);
    BMX_LAUNCH_CHECK();
}

}  // namespace bmx
