// HBM-bound correction primitives of the fastMNN merge step (R/fastMNN.R:567-658, R/utils_tricube.R:1-27).
// One pass over [cells x d] FP64 rows each; coalesced along d; deterministic reductions (no float atomics), so a
// run is bitwise reproducible and 1/2/4/8-GPU runs agree.
#include "bmx_ops.hpp"

namespace bmx {
namespace {

constexpr int RED_ROWS = 512;  // rows per block in the column reductions

__global__ __launch_bounds__(256) void col_reduce_partial(const double* __restrict__ X,
                                                          const int32_t* __restrict__ rows, int r0, int r1, int d,
                                                          int mode, const double* __restrict__ centre,
                                                          double* __restrict__ partial) {
    __shared__ double sm[4][64];
    const int c0 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int b0 = r0 + blockIdx.x * RED_ROWS;
    const int b1 = min(r1, b0 + RED_ROWS);
    for (int cb = 0; cb < d; cb += 64) {
        const int c = cb + c0;
        double s = 0.0;
        if (c < d) {
            const double m = mode == 2 ? centre[c] : 0.0;
            for (int r = b0 + rl; r < b1; r += 4) {
                const int64_t row = rows ? rows[r] : r;
                const double x = X[row * d + c];
                if (mode == 0)
                    s += x;
                else if (mode == 1)
                    s += x * x;
                else
                    s += (x - m) * (x - m);
            }
        }
        sm[rl][c0] = s;
        __syncthreads();
        if (rl == 0 && c < d) partial[(int64_t)blockIdx.x * d + c] = (sm[0][c0] + sm[1][c0]) + (sm[2][c0] + sm[3][c0]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void col_reduce_final(const double* __restrict__ partial, int nblocks, int d,
                                                        double scale, double* __restrict__ out) {
    // one workgroup; four thread groups share each column's partials, combined in a fixed order (deterministic)
    __shared__ double sm[4][64];
    const int c0 = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int cb = 0; cb < d; cb += 64) {
        const int c = cb + c0;
        double s = 0.0;
        if (c < d)
            for (int b = g; b < nblocks; b += 4) s += partial[(int64_t)b * d + c];
        sm[g][c0] = s;
        __syncthreads();
        if (g == 0 && c < d) out[c] = ((sm[0][c0] + sm[1][c0]) + (sm[2][c0] + sm[3][c0])) * scale;
        __syncthreads();
    }
}

// ---- per-segment variants: all original batches of a node in one launch (.compute_perbatch_var) -----------------
struct SegDesc {
    int start[16];
    int n[16];
    int nseg;
};

__global__ __launch_bounds__(256) void seg_reduce_partial(const double* __restrict__ X, int d, SegDesc sd, int mode,
                                                          const double* __restrict__ centres, int maxnb,
                                                          double* __restrict__ partial) {
    __shared__ double sm[4][64];
    const int seg = blockIdx.y;
    const int c0 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int b0 = sd.start[seg] + blockIdx.x * RED_ROWS;
    const int b1 = min(sd.start[seg] + sd.n[seg], b0 + RED_ROWS);
    if (blockIdx.x * RED_ROWS >= sd.n[seg]) return;
    for (int cb = 0; cb < d; cb += 64) {
        const int c = cb + c0;
        double s = 0.0;
        if (c < d) {
            const double m = mode == 2 ? centres[(int64_t)seg * d + c] : 0.0;
            for (int r = b0 + rl; r < b1; r += 4) {
                const double x = X[(int64_t)r * d + c];
                s += mode == 0 ? x : (x - m) * (x - m);
            }
        }
        sm[rl][c0] = s;
        __syncthreads();
        if (rl == 0 && c < d)
            partial[((int64_t)seg * maxnb + blockIdx.x) * d + c] = (sm[0][c0] + sm[1][c0]) + (sm[2][c0] + sm[3][c0]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void seg_reduce_final(const double* __restrict__ partial, int maxnb, int d, SegDesc sd,
                                                        int mode, double* __restrict__ out) {
    __shared__ double sm[4][64];
    const int seg = blockIdx.x;
    const int nb = (sd.n[seg] + RED_ROWS - 1) / RED_ROWS;
    const double scale = mode == 0 ? 1.0 / (double)sd.n[seg] : 1.0 / (double)(sd.n[seg] - 1);
    const int c0 = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int cb = 0; cb < d; cb += 64) {
        const int c = cb + c0;
        double s = 0.0;
        if (c < d)
            for (int b = g; b < nb; b += 4) s += partial[((int64_t)seg * maxnb + b) * d + c];
        sm[g][c0] = s;
        __syncthreads();
        if (g == 0 && c < d) out[(int64_t)seg * d + c] = ((sm[0][c0] + sm[1][c0]) + (sm[2][c0] + sm[3][c0])) * scale;
        __syncthreads();
    }
}

__global__ void seg_sum_kernel(const double* __restrict__ vars, int d, int nseg, double* __restrict__ out,
                               int out_stride) {
    const int seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= nseg) return;
    double s = 0.0;
    for (int c = 0; c < d; ++c) s += vars[(int64_t)seg * d + c];
    out[(int64_t)seg * out_stride] = s;
}

__global__ void sum_vector_kernel(const double* __restrict__ in, int d, double scale, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int c = 0; c < d; ++c) s += in[c];
    *out = s * scale;
}

// one wave per MNN-involved right cell, lanes over dimensions
__global__ __launch_bounds__(256) void average_correction_kernel(
    const double* __restrict__ L, const int32_t* __restrict__ lrows, const double* __restrict__ R,
    const int32_t* __restrict__ rrows, int d, const int32_t* __restrict__ second_u, int U,
    const int32_t* __restrict__ partR, const int32_t* __restrict__ cntR, int k1, double* __restrict__ averaged) {
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (u >= U) return;
    const int r = second_u[u];
    const int m = cntR[r];
    const double* rv = R + (int64_t)(rrows ? rrows[r] : r) * d;
    const int32_t* part = partR + (int64_t)r * k1;
    for (int c = lane; c < d; c += 64) {
        const double rc = rv[c];
        double s = 0.0;
        for (int p = 0; p < m; ++p) {
            const int l = part[p];
            s += L[(int64_t)(lrows ? lrows[l] : l) * d + c] - rc;
        }
        averaged[(int64_t)u * d + c] = s / (double)m;
    }
}

// projections onto the unit batch vector; one wave per row
__global__ __launch_bounds__(256) void project_rows(const double* __restrict__ X, int n, int d,
                                                    const double* __restrict__ vec, double* __restrict__ loc) {
    __shared__ double vhat[256];
    __shared__ double nrm;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int c = 0; c < d; ++c) s += vec[c] * vec[c];
        nrm = sqrt(s);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) vhat[c] = vec[c] / nrm;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        double s = 0.0;
        for (int c = lane; c < d; c += 64) s += X[(int64_t)i * d + c] * vhat[c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) loc[i] = s;
    }
}

// deterministic mean of loc over a row list: block partials then one thread
__global__ __launch_bounds__(256) void mean_partial(const double* __restrict__ loc, const int32_t* __restrict__ rows,
                                                    int n, double* __restrict__ partial) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += loc[rows ? rows[i] : i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ void mean_final(const double* __restrict__ partial, int nb, double inv_n, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += partial[b];
    *out = s * inv_n;
}

__global__ __launch_bounds__(256) void shift_rows(double* __restrict__ X, int n, int d, const double* __restrict__ vec,
                                                  const double* __restrict__ loc, const double* __restrict__ central) {
    __shared__ double vhat[256];
    __shared__ double nrm;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int c = 0; c < d; ++c) s += vec[c] * vec[c];
        nrm = sqrt(s);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) vhat[c] = vec[c] / nrm;
    __syncthreads();
    const double cen = *central;
    const int64_t total = (int64_t)n * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e / d), c = (int)(e - (int64_t)i * d);
        X[e] = X[e] + (cen - loc[i]) * vhat[c];  // mat + outer(central.loc - batch.loc, batch.vec)
    }
}

// one wave per row: tricube weights from the k ascending distances, then the weighted sum of correction vectors
__global__ __launch_bounds__(256) void tricube_apply_kernel(double* __restrict__ X, int n, int d,
                                                            const double* __restrict__ averaged,
                                                            const int32_t* __restrict__ idx,
                                                            const double* __restrict__ dist, int k, double ndist) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const double* di = dist + (int64_t)i * k;
    const int32_t* ii = idx + (int64_t)i * k;
    const int middle = (k + 1) / 2;  // ceiling(k / 2), 1-based
    double bw = di[middle - 1] * ndist;
    bw = bw < 1e-8 ? 1e-8 : bw;  // pmax(1e-8, bandwidth)
    double total = 0.0;
    for (int j = 0; j < k; ++j) {
        double rel = di[j] / bw;
        rel = rel > 1.0 ? 1.0 : rel;
        const double t = 1.0 - rel * rel * rel;
        total += t * t * t;
    }
    for (int c = lane; c < d; c += 64) {
        double acc = 0.0;
        for (int j = 0; j < k; ++j) {
            double rel = di[j] / bw;
            rel = rel > 1.0 ? 1.0 : rel;
            const double t = 1.0 - rel * rel * rel;
            const double w = (t * t * t) / total;
            acc = acc + averaged[(int64_t)ii[j] * d + c] * w;
        }
        X[(int64_t)i * d + c] = X[(int64_t)i * d + c] + acc;
    }
}

__global__ void transpose_kernel(const double* __restrict__ in, int rows_in, int cols_in, double* __restrict__ out,
                                 int ld_out, int out_col_off) {
    // in: [cols_in][rows_in] viewed as column-major (rows_in x cols_in) i.e. in[c * rows_in + r];
    // out[r * ld_out + out_col_off + c]  (row-major with leading dimension ld_out)
    __shared__ double tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int cc = threadIdx.y; cc < 32; cc += 8) {
        const int r = r0 + threadIdx.x, c = c0 + cc;
        if (r < rows_in && c < cols_in) tile[cc][threadIdx.x] = in[(int64_t)c * rows_in + r];
    }
    __syncthreads();
    for (int rr = threadIdx.y; rr < 32; rr += 8) {
        const int r = r0 + rr, c = c0 + threadIdx.x;
        if (r < rows_in && c < cols_in) out[(int64_t)r * ld_out + out_col_off + c] = tile[threadIdx.x][rr];
    }
}

}  // namespace

void col_reduce(hipStream_t stream, ReduceWorkspace& ws, const double* X, const int32_t* rows, int r0, int r1, int d,
                int mode, const double* centre, double scale, double* out) {
    const int n = r1 - r0;
    const int nb = std::max(1, cdiv(n, RED_ROWS));
    double* partial = ws.partial.reserve((size_t)nb * d);
    hipLaunchKernelGGL(col_reduce_partial, dim3(nb), dim3(256), 0, stream, X, rows, r0, r1, d, mode, centre, partial);
    BMX_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_reduce_final, dim3(1), dim3(256), 0, stream, partial, nb, d, scale, out);
    BMX_LAUNCH_CHECK();
}

void segment_variances(hipStream_t stream, ReduceWorkspace& ws, const double* X, int d, const int* starts,
                       const int* ns, int nseg, double* out, int out_stride) {
    for (int s0 = 0; s0 < nseg; s0 += 16) {
        SegDesc sd;
        sd.nseg = std::min(16, nseg - s0);
        int maxn = 1;
        for (int i = 0; i < sd.nseg; ++i) {
            sd.start[i] = starts[s0 + i];
            sd.n[i] = ns[s0 + i];
            maxn = std::max(maxn, ns[s0 + i]);
        }
        const int maxnb = cdiv(maxn, RED_ROWS);
        double* partial = ws.partial.reserve((size_t)sd.nseg * maxnb * d + (size_t)2 * 16 * d);
        double* means = partial + (size_t)sd.nseg * maxnb * d;
        double* vars = means + (size_t)16 * d;
        hipLaunchKernelGGL(seg_reduce_partial, dim3(maxnb, sd.nseg), dim3(256), 0, stream, X, d, sd, 0, nullptr, maxnb,
                           partial);
        hipLaunchKernelGGL(seg_reduce_final, dim3(sd.nseg), dim3(256), 0, stream, partial, maxnb, d, sd, 0, means);
        hipLaunchKernelGGL(seg_reduce_partial, dim3(maxnb, sd.nseg), dim3(256), 0, stream, X, d, sd, 2, means, maxnb,
                           partial);
        hipLaunchKernelGGL(seg_reduce_final, dim3(sd.nseg), dim3(256), 0, stream, partial, maxnb, d, sd, 2, vars);
        hipLaunchKernelGGL(seg_sum_kernel, dim3(1), dim3(64), 0, stream, vars, d, sd.nseg, out + (size_t)s0 * out_stride,
                           out_stride);
        BMX_LAUNCH_CHECK();
    }
}

void sum_vector(hipStream_t stream, const double* in, int d, double scale, double* out) {
    hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(64), 0, stream, in, d, scale, out);
    BMX_LAUNCH_CHECK();
}

void average_correction(hipStream_t stream, const double* L, const int32_t* lrows, const double* R,
                        const int32_t* rrows, int d, const int32_t* second_u, int U, const int32_t* partR,
                        const int32_t* cntR, int k1, double* averaged) {
    if (U <= 0) return;
    hipLaunchKernelGGL(average_correction_kernel, dim3(cdiv(U, 4)), dim3(256), 0, stream, L, lrows, R, rrows, d,
                       second_u, U, partR, cntR, k1, averaged);
    BMX_LAUNCH_CHECK();
}

void center_along_batch_vector(hipStream_t stream, ReduceWorkspace& ws, double* X, int n, int d, const double* vec,
                               const int32_t* restrict_rows, int n_restrict, double* loc, double* scratch3) {
    if (n <= 0) return;
    if (d > 256) throw Error(BMX_ERR_ARG, "more than 256 dimensions are not supported");
    const int gp = std::min(cdiv(n, 4), 4096);
    hipLaunchKernelGGL(project_rows, dim3(gp), dim3(256), 0, stream, X, n, d, vec, loc);
    BMX_LAUNCH_CHECK();
    const int m = restrict_rows ? n_restrict : n;
    const int nb = std::min(std::max(1, cdiv(m, 1024)), 1024);
    double* partial = ws.partial.reserve(nb);
    hipLaunchKernelGGL(mean_partial, dim3(nb), dim3(256), 0, stream, loc, restrict_rows, m, partial);
    BMX_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_final, dim3(1), dim3(64), 0, stream, partial, nb, 1.0 / (double)m, scratch3);
    BMX_LAUNCH_CHECK();
    const int gs = (int)std::min<int64_t>(((int64_t)n * d + 255) / 256, 8192);
    hipLaunchKernelGGL(shift_rows, dim3(gs), dim3(256), 0, stream, X, n, d, vec, loc, scratch3);
    BMX_LAUNCH_CHECK();
}

void tricube_apply(hipStream_t stream, double* X, int n, int d, const double* averaged, const int32_t* idx,
                   const double* dist, int k, double ndist) {
    if (n <= 0 || k <= 0) return;  // k == 0: the weighted correction is all zeros (R/utils_tricube.R:22-23)
    hipLaunchKernelGGL(tricube_apply_kernel, dim3(cdiv(n, 4)), dim3(256), 0, stream, X, n, d, averaged, idx, dist, k,
                       ndist);
    BMX_LAUNCH_CHECK();
}

void transpose_cm_to_rm(hipStream_t stream, const double* cm, int n, int d, double* rm) {
    if (n <= 0) return;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(n, 32), cdiv(d, 32)), dim3(32, 8), 0, stream, cm, n, d, rm, d, 0);
    BMX_LAUNCH_CHECK();
}

void transpose_rm_to_cm(hipStream_t stream, const double* rm, int n, int d, double* cm, int ld_cm, int row_off) {
    // row-major [n][d] is column-major (d x n): "rows_in" = d, "cols_in" = n; out[(c=dim) * ld_cm + row_off + (r=cell)]
    if (n <= 0) return;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(d, 32), cdiv(n, 32)), dim3(32, 8), 0, stream, rm, d, n, cm, ld_cm,
                       row_off);
    BMX_LAUNCH_CHECK();
}

}  // namespace bmx
