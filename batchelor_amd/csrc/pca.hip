// multiBatchPCA on the device (R/multiBatchPCA.R:211-322): the step directly upstream of the merge engine in fastMNN()
// (R/fastMNN.R:353-354), "often the most time-consuming step" (R/reducedMNN.R:25).
//
// The reference takes an SVD of the scaled genes x cells matrix  A = [ C_1 sqrt(w_1/n_1) | C_2 sqrt(w_2/n_2) | ... ],
// C_b = x_b - mu 1^T with mu the weighted mean of the batch means, and projects the UNSCALED centred batches on the top
// d left singular vectors u.  At 20 000 genes x 800 000 cells A is 128 GB; here it is never formed, and neither is the
// genes x genes Gram matrix: the batches stay in HBM as uploaded (genes x cells column-major = one contiguous gene
// vector per cell), and the top subspace of  M = A A^T = sum_b (w_b/n_b) C_b C_b^T  is found by blocked subspace
// iteration with L = 64 vectors (d <= 50 wanted + oversampling):
//     Z_b = C_b^T Q            (cells x 64)   -- "NT" product, K = genes
//     Y  += (w_b/n_b) C_b Z_b  (genes x 64)   -- "TN" product, K = cells
//     Q   = orth(Y)            (Cholesky QR, twice)
// and a Rayleigh-Ritz step on the last pair (Q, Y = M Q) gives the rotation and the singular values.  Cosine
// normalisation (R/cosineNorm.R, fastMNN's cos.norm=TRUE) is a per-cell factor folded into the two products, the
// centring a rank-one correction -- the normalised, centred data is never written.
// Both products run on the FP64 matrix cores (v_mfma_f64_16x16x4_f64): 64 x 64 output tiles per workgroup, operands
// brought in by whole-row 16-byte loads one K step ahead (registers) and staged in the LDS with pitches that keep the
// 32-lane fragment reads conflict-free.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "bmx_ops.hpp"

namespace bmx {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));  // two doubles at any 8-byte boundary
constexpr int PL = 64;   // subspace width (MFMA tile multiple)
constexpr int KC = 32;   // K elements staged per step

// ---------------------------------------------------------------------------------------------------
// NT:  Z[r][j] = rs[r] * sum_k X[r][k] * B[j][k]  -  off[j]        X [n][K] row-major, B [64][K] row-major, Z [n][64]
// (rs = per-row factor, e.g. 1 / max(1e-8, l2) of the cosine normalisation; off = mu . B_j; either may be null)
// A operand of v_mfma_f64_16x16x4_f64: lane l holds A[row = l & 15][k = l >> 4]; B operand: B[k = l >> 4][col = l & 15];
// C/D: 4 doubles per lane, col = l & 15, row = (l >> 4) + 4 * reg.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_nt64(const double* __restrict__ X, int64_t n, int K, int64_t ldx,
                                                 const double* __restrict__ B, int64_t ldb,
                                                 const double* __restrict__ rs, const double* __restrict__ off,
                                                 double* __restrict__ Z) {
    constexpr int P = KC + 2;  // pitch 34 doubles: lanes (row 0..15, k 0..1) hit 32 different 8-byte bank pairs
    __shared__ __attribute__((aligned(16))) double xs[64 * P];
    __shared__ __attribute__((aligned(16))) double bs[64 * P];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    // a K step of both operands is 64 rows x 256 bytes: sixteen lanes take one row piece in 16-byte loads (whole cache
    // lines per row), and the step after the one being multiplied is already on its way into registers
    const int lrow = tid >> 4, lk = (tid & 15) * 2;
    const double* xp[4];
    const double* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + lrow + 16 * i;
        xp[i] = X + (r < n ? r : n - 1) * ldx + lk;  // rows past the end: any valid row, never stored
        bp[i] = B + (int64_t)(lrow + 16 * i) * ldb + lk;
    }
    d2u px[4], pb[4];
    auto fetch = [&](const int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            px[i] = *reinterpret_cast<const d2u*>(xp[i] + k0);
            pb[i] = *reinterpret_cast<const d2u*>(bp[i] + k0);
        }
    };
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {
            const double a = xs[(16 * w + (lane & 15)) * P + 4 * kk + (lane >> 4)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double b = bs[(16 * t + (lane & 15)) * P + 4 * kk + (lane >> 4)];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
    };
    const int nfull = K / KC;
    if (nfull > 0) fetch(0);
    for (int st = 0; st < nfull; ++st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<d2u*>(&xs[(lrow + 16 * i) * P + lk]) = px[i];
            *reinterpret_cast<d2u*>(&bs[(lrow + 16 * i) * P + lk]) = pb[i];
        }
        __syncthreads();
        if (st + 1 < nfull) fetch((st + 1) * KC);
        multiply();
        __syncthreads();
    }
    if (K % KC) {  // the ragged last step, element by element
        const int k0 = nfull * KC, lr = tid >> 2, seg = (tid & 3) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + seg + e;
            xs[lr * P + seg + e] = (r0 + lr < n && k < K) ? X[(r0 + lr) * ldx + k] : 0.0;
            bs[lr * P + seg + e] = k < K ? B[(int64_t)lr * ldb + k] : 0.0;
        }
        __syncthreads();
        multiply();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = 16 * t + (lane & 15);
        const double o = off ? off[j] : 0.0;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t r = r0 + 16 * w + (lane >> 4) + 4 * reg;
            if (r < n) Z[r * 64 + j] = (rs ? rs[r] : 1.0) * acc[t][reg] - o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// TN:  Ypart[split][g][j] = sum_{r in split} X[r][g] * (rs[r] * Z[r][j])       X [n][G] row-major, Z [n][64]
// grid (ceil(G / 64), nsplit); the partial results are summed in a fixed order by reduce_parts.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_tn64(const double* __restrict__ X, int64_t n, int G, int64_t ldx,
                                                 const double* __restrict__ Z, const double* __restrict__ rs,
                                                 int64_t rows_per_split, double* __restrict__ Ypart) {
    constexpr int P = 64 + 16;  // pitch 80 doubles: lanes (col 0..15, k 0..1) hit 32 different bank pairs
    __shared__ __attribute__((aligned(16))) double xs[KC * P];
    __shared__ __attribute__((aligned(16))) double zs[KC * P];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g0 = blockIdx.x * 64;
    const int64_t rbeg = (int64_t)blockIdx.y * rows_per_split, rend = min(n, rbeg + rows_per_split);
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {
            const double a = xs[(4 * kk + (lane >> 4)) * P + 16 * w + (lane & 15)];  // A[row = gene][k = cell]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double b = zs[(4 * kk + (lane >> 4)) * P + 16 * t + (lane & 15)];  // B[k = cell][col = j]
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
    };
    if (g0 + 64 <= G) {
        // a step is 32 cells x 64 genes (and x 64 subspace columns): thirty-two lanes take one cell's 512 bytes in 16-byte
        // loads, the step after the one being multiplied already on its way into registers
        const int lr = tid >> 5, lg = (tid & 31) * 2;
        d2u px[4], pz[4];
        auto fetch = [&](const int64_t r0) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t r = r0 + lr + 8 * i;
                if (r < rend) {
                    const double f = rs ? rs[r] : 1.0;
                    px[i] = *reinterpret_cast<const d2u*>(X + r * ldx + g0 + lg);
                    const d2u z = *reinterpret_cast<const d2u*>(Z + r * 64 + lg);
                    pz[i] = d2u{f * z[0], f * z[1]};
                } else {
                    px[i] = d2u{0.0, 0.0};
                    pz[i] = d2u{0.0, 0.0};
                }
            }
        };
        if (rbeg < rend) fetch(rbeg);
        for (int64_t r0 = rbeg; r0 < rend; r0 += KC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<d2u*>(&xs[(lr + 8 * i) * P + lg]) = px[i];
                *reinterpret_cast<d2u*>(&zs[(lr + 8 * i) * P + lg]) = pz[i];
            }
            __syncthreads();
            if (r0 + KC < rend) fetch(r0 + KC);
            multiply();
            __syncthreads();
        }
    } else {  // the ragged last gene tile, element by element
        const int lr = tid >> 3, seg = (tid & 7) * 8;  // 32 rows x 8 segments of 8 doubles
        for (int64_t r0 = rbeg; r0 < rend; r0 += KC) {
            const int64_t r = r0 + lr;
            const double f = (r < rend && rs) ? rs[r] : 1.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int g = g0 + seg + e;
                xs[lr * P + seg + e] = (r < rend && g < G) ? X[r * ldx + g] : 0.0;
                zs[lr * P + seg + e] = r < rend ? f * Z[r * 64 + seg + e] : 0.0;
            }
            __syncthreads();
            multiply();
            __syncthreads();
        }
    }
    double* out = Ypart + (int64_t)blockIdx.y * G * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = 16 * t + (lane & 15);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int g = g0 + 16 * w + (lane >> 4) + 4 * reg;
            if (g < G) out[(int64_t)g * 64 + j] = acc[t][reg];
        }
    }
}

__global__ void reduce_parts(const double* __restrict__ part, int nsplit, int64_t len, double alpha, double beta,
                             double* __restrict__ Y) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= len) return;
    double s = 0.0;
    for (int p = 0; p < nsplit; ++p) s += part[(int64_t)p * len + e];
    Y[e] = (beta == 0.0 ? 0.0 : beta * Y[e]) + alpha * s;
}

// column sums of Z [n][64] with the per-row factor: two stages, deterministic
__global__ __launch_bounds__(256) void colsum64_partial(const double* __restrict__ Z, const double* __restrict__ rs,
                                                        int64_t n, int64_t rows_per_block, double* __restrict__ part) {
    __shared__ double sm[4][64];
    const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    double s = 0.0;
    for (int64_t r = r0 + q; r < r1; r += 4) s += (rs ? rs[r] : 1.0) * Z[r * 64 + j];
    sm[q][j] = s;
    __syncthreads();
    if (q == 0) part[(int64_t)blockIdx.x * 64 + j] = (sm[0][j] + sm[1][j]) + (sm[2][j] + sm[3][j]);
}

// per-cell 1 / max(1e-8, l2) (R/cosineNorm.R:63-82); one wave per cell
__global__ __launch_bounds__(256) void inv_l2_kernel(const double* __restrict__ X, int64_t n, int G, double* __restrict__ inv) {
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= n) return;
    const double* col = X + c * G;
    double s = 0.0;
    for (int g = lane; g < G; g += 64) s += col[g] * col[g];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const double l2 = sqrt(s);
        inv[c] = 1.0 / (l2 < 1e-8 ? 1e-8 : l2);
    }
}

// gene sums over the cells of a chunk: part[chunk][g] = sum_c rs[c] X[c][g]
__global__ __launch_bounds__(256) void genesum_partial(const double* __restrict__ X, const double* __restrict__ rs, int64_t n,
                                                       int G, int64_t rows_per_chunk, double* __restrict__ part) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    double s = 0.0;
    for (int64_t r = r0; r < r1; ++r) s += (rs ? rs[r] : 1.0) * X[r * G + g];
    part[(int64_t)blockIdx.y * G + g] = s;
}

__global__ void axpy_kernel(double* __restrict__ y, const double* __restrict__ x, double a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}

// Y[g][j] -= coef * mu[g] * zsum[j]
__global__ void rank1_sub(double* __restrict__ Y, const double* __restrict__ mu, const double* __restrict__ zsum, double coef,
                          int G) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)G * 64) return;
    Y[e] -= coef * mu[e >> 6] * zsum[e & 63];
}

__global__ void transpose64(const double* __restrict__ in, int64_t rows, double* __restrict__ out) {
    // in [rows][64] -> out [64][rows]
    __shared__ double tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int rr = e >> 6, j = e & 63;
        tile[rr][j] = r0 + rr < rows ? in[(r0 + rr) * 64 + j] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int j = e >> 6, rr = e & 63;
        if (r0 + rr < rows) out[(int64_t)j * rows + r0 + rr] = tile[rr][j];
    }
}

// ---- small dense helpers on the host (64 x 64) --------------------------------------------------------
// upper-triangular R with R^T R = S (S symmetric positive definite, row-major); returns false if not
bool cholesky_upper(std::vector<double>& S, int n) {
    for (int i = 0; i < n; ++i) {
        for (int j = i; j < n; ++j) {
            double s = S[(size_t)i * n + j];
            for (int k = 0; k < i; ++k) s -= S[(size_t)k * n + i] * S[(size_t)k * n + j];
            if (i == j) {
                if (!(s > 0.0)) return false;
                S[(size_t)i * n + i] = std::sqrt(s);
            } else {
                S[(size_t)i * n + j] = s / S[(size_t)i * n + i];
            }
        }
        for (int j = 0; j < i; ++j) S[(size_t)i * n + j] = 0.0;
    }
    return true;
}
// inverse of an upper-triangular matrix (row-major), in place
void invert_upper(std::vector<double>& R, int n) {
    std::vector<double> inv((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        inv[(size_t)j * n + j] = 1.0 / R[(size_t)j * n + j];
        for (int i = j - 1; i >= 0; --i) {
            double s = 0.0;
            for (int k = i + 1; k <= j; ++k) s += R[(size_t)i * n + k] * inv[(size_t)k * n + j];
            inv[(size_t)i * n + j] = -s / R[(size_t)i * n + i];
        }
    }
    R = inv;
}
// cyclic Jacobi eigen-decomposition of a symmetric matrix: A -> eigenvalues on the diagonal, V columns = eigenvectors
void jacobi_eigen(std::vector<double>& A, std::vector<double>& V, int n) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double offd = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) (i == j ? diag : offd) += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        if (offd <= 1e-30 * diag) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
struct PcaBatch {
    DevBuf<double> x;    // [n][G]  (= genes x cells column-major)
    DevBuf<double> inv;  // [n] 1 / max(1e-8, l2), empty without cosine normalisation
    int64_t n = 0;
    double weight = 1.0;
    bool cos_norm = false;
};

class Pca {
  public:
    Pca(int device, int G) : device_(device), G_(G) {
        BMX_HIP(hipSetDevice(device_));
        BMX_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    }
    ~Pca() {
        (void)hipSetDevice(device_);
        if (stream_) {
            (void)hipStreamSynchronize(stream_);
            (void)hipStreamDestroy(stream_);
        }
        DevBlockCache::current() = &cache_;
    }
    DevBlockCache* cache() { return &cache_; }

    void add_batch(const double* x, int64_t n, double weight, bool cos_norm) {
        CacheScope scope(&cache_);
        BMX_HIP(hipSetDevice(device_));
        if (n < 1) throw Error(BMX_ERR_ARG, "every batch needs at least one cell");
        batches_.emplace_back(new PcaBatch());
        PcaBatch& b = *batches_.back();
        b.n = n;
        b.weight = weight;
        b.cos_norm = cos_norm;
        double* p = b.x.reserve((size_t)n * G_);
        // blocked upload: at most 1 GiB per copy (pageable host memory)
        const int64_t per = std::max<int64_t>(1, ((int64_t)1 << 27) / G_);
        for (int64_t c0 = 0; c0 < n; c0 += per) {
            const int64_t m = std::min(per, n - c0);
            BMX_HIP(hipMemcpyAsync(p + c0 * G_, x + c0 * G_, (size_t)m * G_ * sizeof(double), hipMemcpyHostToDevice,
                                   stream_));
        }
        if (cos_norm) {
            double* inv = b.inv.reserve((size_t)n);
            hipLaunchKernelGGL(inv_l2_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream_, p, n, G_, inv);
            BMX_LAUNCH_CHECK();
        }
        BMX_HIP(hipStreamSynchronize(stream_));  // the caller's buffer is free again
        fitted_ = false;
    }

    // multiBatchPCA: centres [G], rotation [G x d] column-major, sdev [d] (singular values of the scaled matrix)
    void fit(int d, int iters, double* centers, double* rotation, double* sdev) {
        CacheScope scope(&cache_);
        BMX_HIP(hipSetDevice(device_));
        if (batches_.empty()) throw Error(BMX_ERR_ARG, "at least one batch must be specified");
        if (d < 1 || d > PL - 8) throw Error(BMX_ERR_ARG, "the device PCA takes 1 <= d <= 56");
        if (d > G_) throw Error(BMX_ERR_ARG, "d exceeds the number of genes");
        d_ = d;
        const int G = G_;
        // ---- grand centre: weighted mean of the batch means (R/multiBatchPCA.R:268-281)
        double* mu = mu_.reserve((size_t)G);
        BMX_HIP(hipMemsetAsync(mu, 0, (size_t)G * sizeof(double), stream_));
        double wsum = 0.0;
        for (auto& bp : batches_) wsum += bp->weight;
        for (auto& bp : batches_) {
            PcaBatch& b = *bp;
            const int nchunk = (int)std::min<int64_t>(512, std::max<int64_t>(1, b.n / 256));
            const int64_t per = (b.n + nchunk - 1) / nchunk;
            double* part = part_.reserve((size_t)nchunk * G + (size_t)G);
            double* mean = part + (size_t)nchunk * G;
            hipLaunchKernelGGL(genesum_partial, dim3(cdiv(G, 256), nchunk), dim3(256), 0, stream_, b.x.p,
                               b.cos_norm ? b.inv.p : nullptr, b.n, G, per, part);
            hipLaunchKernelGGL(reduce_parts, dim3((unsigned)cdiv(G, 256)), dim3(256), 0, stream_, part, nchunk, (int64_t)G,
                               1.0 / (double)b.n, 0.0, mean);
            hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)cdiv(G, 256)), dim3(256), 0, stream_, mu, mean, b.weight / wsum,
                               (int64_t)G);
            BMX_LAUNCH_CHECK();
        }
        // ---- starting block: a fixed pseudo-random G x 64 matrix, orthonormalised
        double* Q = q_.reserve((size_t)G * PL);   // [G][64]
        double* Y = y_.reserve((size_t)G * PL);   // [G][64]
        double* Qt = qt_.reserve((size_t)G * PL); // [64][G]
        {
            std::vector<double> h((size_t)G * PL);
            unsigned long long st = 0x9E3779B97F4A7C15ull;
            for (auto& v : h) {  // splitmix64 -> uniform in (-1, 1): any full-rank start will do
                st += 0x9E3779B97F4A7C15ull;
                unsigned long long z = st;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z ^= z >> 31;
                v = (double)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
            }
            BMX_HIP(hipMemcpyAsync(Y, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
            BMX_HIP(hipStreamSynchronize(stream_));
        }
        orthonormalise(Y, Q);
        for (int it = 0; it < iters; ++it) {
            apply_operator(Q, Qt, Y);  // Y = M Q
            if (it + 1 < iters) orthonormalise(Y, Q);
        }
        // ---- Rayleigh-Ritz on (Q, Y = M Q): T = Q^T Y, eigenvectors V, rotation = Q V
        double* T = small_.reserve((size_t)PL * PL * 3);
        product_tn(Q, Y, (int64_t)G, T);
        std::vector<double> hT((size_t)PL * PL), V;
        BMX_HIP(hipMemcpyAsync(hT.data(), T, hT.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
        BMX_HIP(hipStreamSynchronize(stream_));
        for (int i = 0; i < PL; ++i)
            for (int j = i + 1; j < PL; ++j) {
                const double v = 0.5 * (hT[(size_t)i * PL + j] + hT[(size_t)j * PL + i]);
                hT[(size_t)i * PL + j] = hT[(size_t)j * PL + i] = v;
            }
        jacobi_eigen(hT, V, PL);
        std::vector<int> order(PL);
        for (int i = 0; i < PL; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return hT[(size_t)a * PL + a] > hT[(size_t)b * PL + b]; });
        // Bm [j][g'] = V[g'][order[j]]: rotation column j = Q V[:, order[j]]
        std::vector<double> Bm((size_t)PL * PL, 0.0);
        for (int j = 0; j < PL; ++j)
            for (int gq = 0; gq < PL; ++gq) Bm[(size_t)j * PL + gq] = V[(size_t)gq * PL + order[j]];
        double* dB = T + (size_t)PL * PL;
        BMX_HIP(hipMemcpyAsync(dB, Bm.data(), Bm.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
        double* U = u_.reserve((size_t)G * PL);  // [G][64], columns sorted by eigenvalue
        hipLaunchKernelGGL(gemm_nt64, dim3((unsigned)cdiv(G, 64)), dim3(256), 0, stream_, Q, (int64_t)G, PL, (int64_t)PL, dB,
                           (int64_t)PL, nullptr, nullptr, U);
        BMX_LAUNCH_CHECK();
        double* Ut = ut_.reserve((size_t)G * PL);  // [64][G]: the B operand of the projection
        hipLaunchKernelGGL(transpose64, dim3((unsigned)cdiv(G, 64)), dim3(256), 0, stream_, U, (int64_t)G, Ut);
        BMX_LAUNCH_CHECK();
        // mu . u_j for the projection's centring
        double* muU = T + (size_t)2 * PL * PL;
        hipLaunchKernelGGL(gemm_nt64, dim3(1), dim3(256), 0, stream_, mu, (int64_t)1, G, (int64_t)G, Ut, (int64_t)G, nullptr,
                           nullptr, muU);  // one row: Z[0][j] = mu . Ut[j]
        BMX_LAUNCH_CHECK();
        if (centers) BMX_HIP(hipMemcpyAsync(centers, mu, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, stream_));
        if (rotation)  // the first d rows of Ut are the d columns of the rotation, column-major
            BMX_HIP(hipMemcpyAsync(rotation, Ut, (size_t)G * d * sizeof(double), hipMemcpyDeviceToHost, stream_));
        BMX_HIP(hipStreamSynchronize(stream_));
        if (sdev)
            for (int j = 0; j < d; ++j) sdev[j] = std::sqrt(std::max(0.0, hT[(size_t)order[j] * PL + order[j]]));
        fitted_ = true;
    }

    // crossprod(x_b - centers, rotation): [n_b x d] column-major
    void project(int b, double* out) {
        CacheScope scope(&cache_);
        BMX_HIP(hipSetDevice(device_));
        if (!fitted_) throw Error(BMX_ERR_ARG, "bmx_pca_fit has not been run");
        if (b < 0 || b >= (int)batches_.size()) throw Error(BMX_ERR_ARG, "batch index out of range");
        PcaBatch& B = *batches_[b];
        double* Z = z_.reserve((size_t)B.n * PL);
        double* muU = small_.p + (size_t)2 * PL * PL;
        hipLaunchKernelGGL(gemm_nt64, dim3((unsigned)cdiv(B.n, 64)), dim3(256), 0, stream_, B.x.p, B.n, G_, (int64_t)G_,
                           ut_.p, (int64_t)G_, B.cos_norm ? B.inv.p : nullptr, muU, Z);
        BMX_LAUNCH_CHECK();
        double* Zt = zt_.reserve((size_t)B.n * PL);
        hipLaunchKernelGGL(transpose64, dim3((unsigned)cdiv(B.n, 64)), dim3(256), 0, stream_, Z, B.n, Zt);
        BMX_LAUNCH_CHECK();
        BMX_HIP(hipMemcpyAsync(out, Zt, (size_t)B.n * d_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
        BMX_HIP(hipStreamSynchronize(stream_));
    }
    int nbatches() const { return (int)batches_.size(); }
    int64_t ncells(int b) const { return batches_[b]->n; }

  private:
    // out [64][64] row-major = A^T B for A, B [rows][64]
    void product_tn(const double* A, const double* Bm, int64_t rows, double* out) {
        const int nsplit = (int)std::min<int64_t>(256, std::max<int64_t>(1, rows / 512));
        const int64_t per = round_up((rows + nsplit - 1) / nsplit, KC);
        double* part = part_.reserve((size_t)nsplit * PL * PL);
        hipLaunchKernelGGL(gemm_tn64, dim3(1, nsplit), dim3(256), 0, stream_, A, rows, PL, (int64_t)PL, Bm, nullptr, per,
                           part);
        hipLaunchKernelGGL(reduce_parts, dim3((unsigned)cdiv(PL * PL, 256)), dim3(256), 0, stream_, part, nsplit,
                           (int64_t)PL * PL, 1.0, 0.0, out);
        BMX_LAUNCH_CHECK();
    }
    // Q = Y R^-1 with R^T R = Y^T Y, twice (Cholesky QR 2: orthonormal to rounding for any reasonable Y)
    void orthonormalise(double* Y, double* Q) {
        const int G = G_;
        double* S = small_.reserve((size_t)PL * PL * 3);
        double* src = Y;
        double* dst = Q;
        for (int pass = 0; pass < 2; ++pass) {
            product_tn(src, src, (int64_t)G, S);
            std::vector<double> h((size_t)PL * PL);
            BMX_HIP(hipMemcpyAsync(h.data(), S, h.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
            BMX_HIP(hipStreamSynchronize(stream_));
            if (!cholesky_upper(h, PL)) throw Error(BMX_ERR_ARG, "PCA: the data has rank below the subspace width");
            invert_upper(h, PL);  // Rinv (upper); Q[g][j] = sum_i Y[g][i] Rinv[i][j] -> NT with Bm[j][i] = Rinv[i][j]
            std::vector<double> bt((size_t)PL * PL);
            for (int i = 0; i < PL; ++i)
                for (int j = 0; j < PL; ++j) bt[(size_t)j * PL + i] = h[(size_t)i * PL + j];
            double* dB = S + (size_t)PL * PL;
            BMX_HIP(hipMemcpyAsync(dB, bt.data(), bt.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
            hipLaunchKernelGGL(gemm_nt64, dim3((unsigned)cdiv(G, 64)), dim3(256), 0, stream_, src, (int64_t)G, PL, (int64_t)PL,
                               dB, (int64_t)PL, nullptr, nullptr, dst);
            BMX_LAUNCH_CHECK();
            BMX_HIP(hipStreamSynchronize(stream_));  // bt goes out of scope
            std::swap(src, dst);
        }
        // two passes: Y -> Q -> Y; the result is back in Y's storage, bring it to Q
        BMX_HIP(hipMemcpyAsync(Q, Y, (size_t)G * PL * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    }
    // Y = M Q = sum_b (w_b / n_b) C_b C_b^T Q
    void apply_operator(const double* Q, double* Qt, double* Y) {
        const int G = G_;
        hipLaunchKernelGGL(transpose64, dim3((unsigned)cdiv(G, 64)), dim3(256), 0, stream_, Q, (int64_t)G, Qt);
        BMX_LAUNCH_CHECK();
        double* muQ = small_.reserve((size_t)PL * PL * 3) + (size_t)2 * PL * PL;
        hipLaunchKernelGGL(gemm_nt64, dim3(1), dim3(256), 0, stream_, mu_.p, (int64_t)1, G, (int64_t)G, Qt, (int64_t)G, nullptr,
                           nullptr, muQ);
        BMX_LAUNCH_CHECK();
        BMX_HIP(hipMemsetAsync(Y, 0, (size_t)G * PL * sizeof(double), stream_));
        for (auto& bp : batches_) {
            PcaBatch& b = *bp;
            const double* rs = b.cos_norm ? b.inv.p : nullptr;
            double* Z = z_.reserve((size_t)b.n * PL);
            // Z = C_b^T Q = diag(rs) X Q - 1 (mu^T Q)
            hipLaunchKernelGGL(gemm_nt64, dim3((unsigned)cdiv(b.n, 64)), dim3(256), 0, stream_, b.x.p, b.n, G, (int64_t)G, Qt,
                               (int64_t)G, rs, muQ, Z);
            BMX_LAUNCH_CHECK();
            // Y += coef (X^T diag(rs) Z - mu (1^T Z))
            const double coef = b.weight / (double)b.n;
            const int gtiles = cdiv(G, 64);
            int nsplit = (int)std::min<int64_t>(std::max<int64_t>(1, (int64_t)1024 / gtiles), std::max<int64_t>(1, b.n / 2048));
            nsplit = std::max(1, nsplit);
            const int64_t per = round_up((b.n + nsplit - 1) / nsplit, KC);
            nsplit = (int)((b.n + per - 1) / per);
            double* part = part_.reserve((size_t)nsplit * G * PL + (size_t)4096 * PL + PL);
            hipLaunchKernelGGL(gemm_tn64, dim3(gtiles, nsplit), dim3(256), 0, stream_, b.x.p, b.n, G, (int64_t)G, Z, rs, per,
                               part);
            hipLaunchKernelGGL(reduce_parts, dim3((unsigned)cdiv((int64_t)G * PL, 256)), dim3(256), 0, stream_, part, nsplit,
                               (int64_t)G * PL, coef, 1.0, Y);
            BMX_LAUNCH_CHECK();
            // column sums of Z (no row factor: the centring term is mu 1^T Z)
            const int nb = (int)std::min<int64_t>(4096, std::max<int64_t>(1, b.n / 256));
            const int64_t rpb = (b.n + nb - 1) / nb;
            double* zpart = part + (size_t)nsplit * G * PL;
            double* zsum = zpart + (size_t)nb * PL;
            hipLaunchKernelGGL(colsum64_partial, dim3(nb), dim3(256), 0, stream_, Z, nullptr, b.n, rpb, zpart);
            hipLaunchKernelGGL(reduce_parts, dim3(1), dim3(64), 0, stream_, zpart, nb, (int64_t)PL, 1.0, 0.0, zsum);
            hipLaunchKernelGGL(rank1_sub, dim3((unsigned)cdiv((int64_t)G * PL, 256)), dim3(256), 0, stream_, Y, mu_.p, zsum, coef,
                               G);
            BMX_LAUNCH_CHECK();
        }
    }

    DevBlockCache cache_;
    int device_ = 0, G_ = 0, d_ = 0;
    hipStream_t stream_ = nullptr;
    std::vector<std::unique_ptr<PcaBatch>> batches_;
    DevBuf<double> mu_, q_, y_, qt_, u_, ut_, z_, zt_, part_, small_;
    bool fitted_ = false;
};

Pca* pca_create(int device, int G) { return new Pca(device, G); }
void pca_destroy(Pca* p) { delete p; }
void pca_add_batch(Pca* p, const double* x, int64_t n, double weight, int cos_norm) { p->add_batch(x, n, weight, cos_norm != 0); }
void pca_fit(Pca* p, int d, int iters, double* centers, double* rotation, double* sdev) {
    p->fit(d, iters, centers, rotation, sdev);
}
void pca_project(Pca* p, int b, double* out) { p->project(b, out); }

}  // namespace bmx
