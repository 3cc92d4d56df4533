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
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "bmx_ops.hpp"
#include "host_xfer.hpp"

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
                                                 double* __restrict__ Z, int64_t ldz) {
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
            if (r < n) Z[r * ldz + j] = (rs ? rs[r] : 1.0) * acc[t][reg] - o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// TN:  Ypart[split][g][j] = sum_{r in split} X[r][g] * (rs[r] * Z[r][j])       X [n][G] row-major, Z [n][64]
// grid (ceil(G / 64), nsplit); the partial results are summed in a fixed order by reduce_parts.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_tn64(const double* __restrict__ X, int64_t n, int G, int64_t ldx,
                                                 const double* __restrict__ Z, int64_t ldz, const double* __restrict__ rs,
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
                    const d2u z = *reinterpret_cast<const d2u*>(Z + r * ldz + lg);
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
                zs[lr * P + seg + e] = r < rend ? f * Z[r * ldz + seg + e] : 0.0;
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

// Y[(e / w) * ldy + e % w] = beta * Y[..] + alpha * sum_p part[p][e]   (parts are dense rows of w columns)
__global__ void reduce_parts(const double* __restrict__ part, int nsplit, int64_t len, double alpha, double beta,
                             double* __restrict__ Y, int w, int64_t ldy) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= len) return;
    double s = 0.0;
    for (int p = 0; p < nsplit; ++p) s += part[(int64_t)p * len + e];
    const int64_t o = (e / w) * ldy + e % w;
    Y[o] = (beta == 0.0 ? 0.0 : beta * Y[o]) + alpha * s;
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
__global__ void rank1_sub(double* __restrict__ Y, int64_t ldy, const double* __restrict__ mu, const double* __restrict__ zsum,
                          double coef, int G) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)G * 64) return;
    Y[(e >> 6) * ldy + (e & 63)] -= coef * mu[e >> 6] * zsum[e & 63];
}

__global__ void transpose64(const double* __restrict__ in, int64_t ld, int64_t rows, double* __restrict__ out) {
    // 64 columns of in [rows][ld] -> out [64][rows]
    __shared__ double tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int rr = e >> 6, j = e & 63;
        tile[rr][j] = r0 + rr < rows ? in[(r0 + rr) * ld + j] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int j = e >> 6, rr = e & 63;
        if (r0 + rr < rows) out[(int64_t)j * rows + r0 + rr] = tile[rr][j];
    }
}

// out = a X + b Y + c Z, element by element (Z may be null); the three-term recurrence of the polynomial filter
__global__ void lincomb3(double* __restrict__ out, double a, const double* __restrict__ X, double b,
                         const double* __restrict__ Y, double c, const double* __restrict__ Z, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = a * X[i] + b * Y[i];
    if (Z) v += c * Z[i];
    out[i] = v;
}

// part[block][j] = sum over the block's rows g of (Yr[g][j] - theta[j] Xr[g][j])^2: the squared residual norms of the
// Ritz pairs (Xr, theta) of the operator whose image of Xr is Yr.  256 rows per block, deterministic.
__global__ __launch_bounds__(256) void resid_partial(const double* __restrict__ Yr, const double* __restrict__ Xr,
                                                     const double* __restrict__ theta, int64_t G, int L,
                                                     double* __restrict__ part) {
    __shared__ double sm[4][128];
    const int j0 = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * 256, r1 = min(G, r0 + 256);
    for (int jb = 0; jb < L; jb += 64) {
        const int j = jb + j0;
        const double th = theta[j];
        double s = 0.0;
        for (int64_t r = r0 + q; r < r1; r += 4) {
            const double t = Yr[r * L + j] - th * Xr[r * L + j];
            s += t * t;
        }
        sm[q][j] = s;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < L; j += 256) part[(int64_t)blockIdx.x * L + j] = (sm[0][j] + sm[1][j]) + (sm[2][j] + sm[3][j]);
}

// ---- small dense helpers on the host (L x L, L = 64 or 128) --------------------------------------------------------
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
    int64_t n = 0;       // cells announced by begin_batch
    int64_t filled = 0;  // cells received so far
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

    // a batch of n cells whose columns arrive in one or more blocks (add_block), in order
    void begin_batch(int64_t n, double weight, bool cos_norm) {
        CacheScope scope(&cache_);
        BMX_HIP(hipSetDevice(device_));
        if (n < 1) throw Error(BMX_ERR_ARG, "every batch needs at least one cell");
        if (!batches_.empty() && batches_.back()->filled != batches_.back()->n)
            throw Error(BMX_ERR_ARG, "the previous batch has not received all its cells");
        batches_.emplace_back(new PcaBatch());
        PcaBatch& b = *batches_.back();
        b.n = n;
        b.weight = weight;
        b.cos_norm = cos_norm;
        b.x.reserve((size_t)n * G_);
        if (cos_norm) b.inv.reserve((size_t)n);
        fitted_ = false;
    }
    // the next m cells (columns) of the batch begun last: x_block is G x m column-major host memory (pageable is fine:
    // it goes through the pinned staging ring, which has taken the block by the time this returns)
    void add_block(const double* x_block, int64_t m) {
        CacheScope scope(&cache_);
        BMX_HIP(hipSetDevice(device_));
        if (batches_.empty()) throw Error(BMX_ERR_ARG, "bmx_pca_begin_batch has not been called");
        PcaBatch& b = *batches_.back();
        if (m < 1 || b.filled + m > b.n) throw Error(BMX_ERR_ARG, "the block does not fit into the batch announced");
        double* p = b.x.p + b.filled * G_;
        upload_pageable(p, x_block, (size_t)m * G_ * sizeof(double), stream_);
        if (b.cos_norm) {
            hipLaunchKernelGGL(inv_l2_kernel, dim3((unsigned)cdiv(m, 4)), dim3(256), 0, stream_, p, m, G_, b.inv.p + b.filled);
            BMX_LAUNCH_CHECK();
        }
        b.filled += m;
    }
    void add_batch(const double* x, int64_t n, double weight, bool cos_norm) {
        begin_batch(n, weight, cos_norm);
        add_block(x, n);
    }

    // multiBatchPCA: centres [G], rotation [G x d] column-major, sdev [d] (singular values of the scaled matrix).
    // Chebyshev-filtered subspace iteration on a block of L = 64 (d <= 56) or 128 (d <= 120) vectors until the Ritz
    // residuals max_j |M x_j - theta_j x_j| / theta_1 of the d wanted pairs are <= tol (tol > 0), at most max_applies
    // applications of M; tol <= 0: exactly max_applies plain subspace steps (the fixed-count form).
    void fit(int d, double tol, int max_applies, double* centers, double* rotation, double* sdev, int* applies_used,
             double* resid_out) {
        CacheScope scope(&cache_);
        BMX_HIP(hipSetDevice(device_));
        if (batches_.empty()) throw Error(BMX_ERR_ARG, "at least one batch must be specified");
        if (batches_.back()->filled != batches_.back()->n)
            throw Error(BMX_ERR_ARG, "the last batch has not received all its cells");
        if (d < 1 || d > 2 * PL - 8) throw Error(BMX_ERR_ARG, "the device PCA takes 1 <= d <= 120");
        if (d > G_) throw Error(BMX_ERR_ARG, "d exceeds the number of genes");
        if (max_applies < 1) throw Error(BMX_ERR_ARG, "the PCA needs at least one iteration");
        const int L = d <= PL - 8 ? PL : 2 * PL;
        int64_t ncells = 0;
        for (auto& bp : batches_) ncells += bp->n;
        if (G_ < L || ncells <= L)
            throw Error(BMX_ERR_ARG, "PCA: the data has rank below the subspace width (fewer genes or cells than the block)");
        L_ = L;
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
                               1.0 / (double)b.n, 0.0, mean, G, (int64_t)G);
            hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)cdiv(G, 256)), dim3(256), 0, stream_, mu, mean, b.weight / wsum,
                               (int64_t)G);
            BMX_LAUNCH_CHECK();
        }
        // ---- starting block: a fixed pseudo-random G x L matrix, orthonormalised
        const size_t GL = (size_t)G * L;
        double* Q = q_.reserve(GL);
        double* Y = y_.reserve(GL);
        double* Xr = xr_.reserve(GL);  // Ritz vectors Q V
        double* Yr = yr_.reserve(GL);  // their images Y V
        double* W = w_.reserve(GL);    // filter scratch
        qt_.reserve((size_t)G * PL);
        small_.reserve((size_t)L * L * 3 + 4 * (size_t)L);
        {
            std::vector<double> h(GL);
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
        std::vector<double> theta(L, 0.0), V;
        std::vector<int> order(L);
        int applies = 0;
        double resid = std::numeric_limits<double>::infinity();
        const bool fixed = !(tol > 0.0);
        for (;;) {
            apply_operator(Q, Y);  // Y = M Q
            ++applies;
            const bool last_fixed = fixed && applies >= max_applies;
            if (fixed && !last_fixed) {  // plain subspace iteration, no convergence test
                orthonormalise(Y, Q);
                continue;
            }
            // ---- Rayleigh-Ritz on (Q, Y = M Q): T = Q^T Y = V diag(theta) V^T; Ritz vectors Xr = Q V, images Yr = Y V
            rayleigh_ritz(Q, Y, theta, V, order);
            rotate(Q, V, order, Xr);
            rotate(Y, V, order, Yr);
            resid = residual(Yr, Xr, theta, d);
            if (last_fixed || resid <= tol || applies >= max_applies) break;
            // ---- next block: p(M) Xr with p the Chebyshev polynomial that is bounded on [0, theta_L] (everything the
            // block does not want) and grows above it; the degree is capped so that the largest wanted direction
            // outgrows the smallest by at most ~1e5 (Cholesky QR squares the block's condition number)
            const double lam = theta[0], cut = theta[L - 1];
            int deg = 1;
            if (cut > 0.0 && lam > cut * (1.0 + 1e-12)) {
                const double x = 2.0 * lam / cut - 1.0;  // (lam - c) / e with c = e = cut / 2
                deg = (int)std::floor(std::log(1e5) / std::acosh(x));
                deg = std::max(1, std::min({deg, 12, max_applies - applies + 1}));
            }
            if (deg <= 1) {
                orthonormalise(Yr, Q);  // plain step from the Ritz basis (same subspace as Y)
                continue;
            }
            // scaled three-term recurrence (p(lam) = 1):
            //   X0 = Xr, X1 = (s1 / e)(M - c) X0, X_{i+1} = 2 (s_{i+1} / e)(M - c) X_i - s_i s_{i+1} X_{i-1}
            const double c = 0.5 * cut, e = 0.5 * cut;
            const double sg1 = e / (lam - c);
            double sg = sg1;
            const int64_t nel = (int64_t)GL;
            const unsigned nblk = (unsigned)cdiv(nel, 256);
            double* X0 = Xr;
            double* X1 = W;
            hipLaunchKernelGGL(lincomb3, dim3(nblk), dim3(256), 0, stream_, X1, sg1 / e, (const double*)Yr, -c * sg1 / e,
                               (const double*)Xr, 0.0, (const double*)nullptr, nel);
            BMX_LAUNCH_CHECK();
            double* spare = Yr;  // Yr is free once X1 exists
            for (int i = 2; i <= deg; ++i) {
                const double sg2 = 1.0 / (2.0 / sg1 - sg);
                apply_operator(X1, Y);  // Y = M X1
                ++applies;
                hipLaunchKernelGGL(lincomb3, dim3(nblk), dim3(256), 0, stream_, spare, 2.0 * sg2 / e, (const double*)Y,
                                   -2.0 * c * sg2 / e, (const double*)X1, -sg * sg2, (const double*)X0, nel);
                BMX_LAUNCH_CHECK();
                double* t = X0;
                X0 = X1;
                X1 = spare;
                spare = t;
                sg = sg2;
            }
            orthonormalise(X1, Q);
        }
        // ---- results: rotation = the first d Ritz vectors
        double* Ut = ut_.reserve(GL);  // [L][G]: the B operand of the projection
        for (int h = 0; h < L / PL; ++h) {
            hipLaunchKernelGGL(transpose64, dim3((unsigned)cdiv(G, 64)), dim3(256), 0, stream_, (const double*)(Xr + h * PL),
                               (int64_t)L, (int64_t)G, Ut + (size_t)h * PL * G);
            BMX_LAUNCH_CHECK();
        }
        // mu . u_j for the projection's centring
        double* muU = small_.p + (size_t)3 * L * L;
        for (int h = 0; h < L / PL; ++h) {
            hipLaunchKernelGGL(gemm_nt64, dim3(1), dim3(256), 0, stream_, (const double*)mu, (int64_t)1, G, (int64_t)G,
                               (const double*)(Ut + (size_t)h * PL * G), (int64_t)G, (const double*)nullptr,
                               (const double*)nullptr, muU + h * PL, (int64_t)PL);  // one row: Z[0][j] = mu . Ut[j]
            BMX_LAUNCH_CHECK();
        }
        if (centers) BMX_HIP(hipMemcpyAsync(centers, mu, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, stream_));
        if (rotation)  // the first d rows of Ut are the d columns of the rotation, column-major
            BMX_HIP(hipMemcpyAsync(rotation, Ut, (size_t)G * d * sizeof(double), hipMemcpyDeviceToHost, stream_));
        BMX_HIP(hipStreamSynchronize(stream_));
        if (sdev)
            for (int j = 0; j < d; ++j) sdev[j] = std::sqrt(std::max(0.0, theta[j]));
        if (applies_used) *applies_used = applies;
        if (resid_out) *resid_out = resid;
        fitted_ = true;
        if (!fixed && !(resid <= tol)) {
            char msg[256];
            std::snprintf(msg, sizeof(msg),
                          "PCA: the subspace iteration did not reach the tolerance within %d applications of the operator "
                          "(relative residual %.3g, tolerance %.3g)", applies, resid, tol);
            throw Error(BMX_ERR_ARG, msg);
        }
    }

    // crossprod(x_b - centers, rotation): [n_b x d] column-major
    void project(int b, double* out) {
        CacheScope scope(&cache_);
        BMX_HIP(hipSetDevice(device_));
        if (!fitted_) throw Error(BMX_ERR_ARG, "bmx_pca_fit has not been run");
        if (b < 0 || b >= (int)batches_.size()) throw Error(BMX_ERR_ARG, "batch index out of range");
        PcaBatch& B = *batches_[b];
        double* Z = z_.reserve((size_t)B.n * PL);
        double* Zt = zt_.reserve((size_t)B.n * PL);
        double* muU = small_.p + (size_t)3 * L_ * L_;
        for (int h = 0; h * PL < d_; ++h) {
            hipLaunchKernelGGL(gemm_nt64, dim3((unsigned)cdiv(B.n, 64)), dim3(256), 0, stream_, (const double*)B.x.p, B.n, G_,
                               (int64_t)G_, (const double*)(ut_.p + (size_t)h * PL * G_), (int64_t)G_,
                               (const double*)(B.cos_norm ? B.inv.p : nullptr), (const double*)(muU + h * PL), Z, (int64_t)PL);
            BMX_LAUNCH_CHECK();
            hipLaunchKernelGGL(transpose64, dim3((unsigned)cdiv(B.n, 64)), dim3(256), 0, stream_, (const double*)Z, (int64_t)PL,
                               B.n, Zt);
            BMX_LAUNCH_CHECK();
            const int cols = std::min(PL, d_ - h * PL);
            BMX_HIP(hipMemcpyAsync(out + (size_t)h * PL * B.n, Zt, (size_t)B.n * cols * sizeof(double), hipMemcpyDeviceToHost,
                                   stream_));
            BMX_HIP(hipStreamSynchronize(stream_));  // Z / Zt are reused by the next half
        }
    }
    int nbatches() const { return (int)batches_.size(); }
    int64_t ncells(int b) const { return batches_[b]->n; }

  private:
    // out [L][L] row-major = A^T B for A, B [rows][L]
    void product_tn(const double* A, const double* Bm, int64_t rows, double* out) {
        const int L = L_;
        const int nsplit = (int)std::min<int64_t>(256, std::max<int64_t>(1, rows / 512));
        const int64_t per = round_up((rows + nsplit - 1) / nsplit, KC);
        double* part = part_.reserve((size_t)nsplit * L * PL);
        for (int h = 0; h < L / PL; ++h) {  // 64 columns of B at a time
            hipLaunchKernelGGL(gemm_tn64, dim3(L / PL, nsplit), dim3(256), 0, stream_, A, rows, L, (int64_t)L, Bm + h * PL,
                               (int64_t)L, (const double*)nullptr, per, part);
            hipLaunchKernelGGL(reduce_parts, dim3((unsigned)cdiv(L * PL, 256)), dim3(256), 0, stream_, (const double*)part,
                               nsplit, (int64_t)L * PL, 1.0, 0.0, out + h * PL, PL, (int64_t)L);
            BMX_LAUNCH_CHECK();
        }
    }
    // dst [G][L] = src [G][L] * Bt^T for a host matrix Bt [L][L] row-major (dst[g][j] = sum_i src[g][i] Bt[j][i])
    void times_small(const double* src, const std::vector<double>& Bt, double* dst) {
        const int L = L_;
        double* dB = small_.p + (size_t)L * L;
        BMX_HIP(hipMemcpyAsync(dB, Bt.data(), Bt.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
        for (int h = 0; h < L / PL; ++h) {
            hipLaunchKernelGGL(gemm_nt64, dim3((unsigned)cdiv(G_, 64)), dim3(256), 0, stream_, src, (int64_t)G_, L, (int64_t)L,
                               (const double*)(dB + (size_t)h * PL * L), (int64_t)L, (const double*)nullptr,
                               (const double*)nullptr, dst + h * PL, (int64_t)L);
            BMX_LAUNCH_CHECK();
        }
        BMX_HIP(hipStreamSynchronize(stream_));  // Bt may go out of scope
    }
    // Q = Y R^-1 with R^T R = Y^T Y, twice (Cholesky QR 2: orthonormal to rounding for any reasonable Y).  Y is
    // overwritten (it holds the first pass's result).
    void orthonormalise(double* Y, double* Q) {
        const int L = L_;
        double* S = small_.p;
        double* src = Y;
        double* dst = Q;
        for (int pass = 0; pass < 2; ++pass) {
            product_tn(src, src, (int64_t)G_, S);
            std::vector<double> h((size_t)L * L);
            BMX_HIP(hipMemcpyAsync(h.data(), S, h.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
            BMX_HIP(hipStreamSynchronize(stream_));
            if (!cholesky_upper(h, L)) throw Error(BMX_ERR_ARG, "PCA: the data has rank below the subspace width");
            invert_upper(h, L);  // Rinv (upper); Q[g][j] = sum_i Y[g][i] Rinv[i][j] -> Bt[j][i] = Rinv[i][j]
            std::vector<double> bt((size_t)L * L);
            for (int i = 0; i < L; ++i)
                for (int j = 0; j < L; ++j) bt[(size_t)j * L + i] = h[(size_t)i * L + j];
            times_small(src, bt, dst);
            std::swap(src, dst);
        }
        // two passes: Y -> Q -> Y; the result is back in Y's storage, bring it to Q
        BMX_HIP(hipMemcpyAsync(Q, Y, (size_t)G_ * L * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    }
    void rayleigh_ritz(const double* Q, const double* Y, std::vector<double>& theta, std::vector<double>& V,
                       std::vector<int>& order) {
        const int L = L_;
        double* T = small_.p;
        product_tn(Q, Y, (int64_t)G_, T);
        std::vector<double> hT((size_t)L * L);
        BMX_HIP(hipMemcpyAsync(hT.data(), T, hT.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
        BMX_HIP(hipStreamSynchronize(stream_));
        for (int i = 0; i < L; ++i)
            for (int j = i + 1; j < L; ++j) {
                const double v = 0.5 * (hT[(size_t)i * L + j] + hT[(size_t)j * L + i]);
                hT[(size_t)i * L + j] = hT[(size_t)j * L + i] = v;
            }
        jacobi_eigen(hT, V, L);
        for (int i = 0; i < L; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return hT[(size_t)a * L + a] > hT[(size_t)b * L + b]; });
        for (int j = 0; j < L; ++j) theta[j] = hT[(size_t)order[j] * L + order[j]];
    }
    // dst = src V with the columns of V taken in `order`
    void rotate(const double* src, const std::vector<double>& V, const std::vector<int>& order, double* dst) {
        const int L = L_;
        std::vector<double> Bt((size_t)L * L);
        for (int j = 0; j < L; ++j)
            for (int i = 0; i < L; ++i) Bt[(size_t)j * L + i] = V[(size_t)i * L + order[j]];
        times_small(src, Bt, dst);
    }
    // max_j<d |Yr_j - theta_j Xr_j| / theta_0
    double residual(const double* Yr, const double* Xr, const std::vector<double>& theta, int d) {
        const int L = L_;
        double* dth = small_.p + (size_t)3 * L * L + L;
        double* dres = dth + L;
        BMX_HIP(hipMemcpyAsync(dth, theta.data(), (size_t)L * sizeof(double), hipMemcpyHostToDevice, stream_));
        const int nb = cdiv(G_, 256);
        double* part = part_.reserve((size_t)nb * L);
        hipLaunchKernelGGL(resid_partial, dim3(nb), dim3(256), 0, stream_, Yr, Xr, (const double*)dth, (int64_t)G_, L, part);
        hipLaunchKernelGGL(reduce_parts, dim3((unsigned)cdiv(L, 256)), dim3(256), 0, stream_, (const double*)part, nb, (int64_t)L,
                           1.0, 0.0, dres, L, (int64_t)L);
        BMX_LAUNCH_CHECK();
        std::vector<double> h(L);
        BMX_HIP(hipMemcpyAsync(h.data(), dres, (size_t)L * sizeof(double), hipMemcpyDeviceToHost, stream_));
        BMX_HIP(hipStreamSynchronize(stream_));
        double worst = 0.0;
        for (int j = 0; j < d; ++j) worst = std::max(worst, std::sqrt(std::max(0.0, h[j])));
        return theta[0] > 0.0 ? worst / theta[0] : 0.0;
    }
    // Y = M Q = sum_b (w_b / n_b) C_b C_b^T Q for a block of L vectors, 64 at a time
    void apply_operator(const double* Q, double* Y) {
        const int G = G_, L = L_;
        double* Qt = qt_.p;
        BMX_HIP(hipMemsetAsync(Y, 0, (size_t)G * L * sizeof(double), stream_));
        for (int h = 0; h < L / PL; ++h) {
            hipLaunchKernelGGL(transpose64, dim3((unsigned)cdiv(G, 64)), dim3(256), 0, stream_, Q + h * PL, (int64_t)L, (int64_t)G,
                               Qt);
            BMX_LAUNCH_CHECK();
            double* muQ = small_.p + (size_t)2 * L * L;
            hipLaunchKernelGGL(gemm_nt64, dim3(1), dim3(256), 0, stream_, (const double*)mu_.p, (int64_t)1, G, (int64_t)G,
                               (const double*)Qt, (int64_t)G, (const double*)nullptr, (const double*)nullptr, muQ, (int64_t)PL);
            BMX_LAUNCH_CHECK();
            double* Yh = Y + h * PL;
            for (auto& bp : batches_) {
                PcaBatch& b = *bp;
                const double* rs = b.cos_norm ? b.inv.p : nullptr;
                double* Z = z_.reserve((size_t)b.n * PL);
                // Z = C_b^T Q = diag(rs) X Q - 1 (mu^T Q)
                hipLaunchKernelGGL(gemm_nt64, dim3((unsigned)cdiv(b.n, 64)), dim3(256), 0, stream_, (const double*)b.x.p, b.n, G,
                                   (int64_t)G, (const double*)Qt, (int64_t)G, rs, (const double*)muQ, Z, (int64_t)PL);
                BMX_LAUNCH_CHECK();
                // Y += coef (X^T diag(rs) Z - mu (1^T Z))
                const double coef = b.weight / (double)b.n;
                const int gtiles = cdiv(G, 64);
                int nsplit = (int)std::min<int64_t>(std::max<int64_t>(1, (int64_t)1024 / gtiles), std::max<int64_t>(1, b.n / 2048));
                nsplit = std::max(1, nsplit);
                const int64_t per = round_up((b.n + nsplit - 1) / nsplit, KC);
                nsplit = (int)((b.n + per - 1) / per);
                double* part = part_.reserve((size_t)nsplit * G * PL + (size_t)4096 * PL + PL);
                hipLaunchKernelGGL(gemm_tn64, dim3(gtiles, nsplit), dim3(256), 0, stream_, (const double*)b.x.p, b.n, G, (int64_t)G,
                                   (const double*)Z, (int64_t)PL, rs, per, part);
                hipLaunchKernelGGL(reduce_parts, dim3((unsigned)cdiv((int64_t)G * PL, 256)), dim3(256), 0, stream_,
                                   (const double*)part, nsplit, (int64_t)G * PL, coef, 1.0, Yh, PL, (int64_t)L);
                BMX_LAUNCH_CHECK();
                // column sums of Z (no row factor: the centring term is mu 1^T Z)
                const int nb = (int)std::min<int64_t>(4096, std::max<int64_t>(1, b.n / 256));
                const int64_t rpb = (b.n + nb - 1) / nb;
                double* zpart = part + (size_t)nsplit * G * PL;
                double* zsum = zpart + (size_t)nb * PL;
                hipLaunchKernelGGL(colsum64_partial, dim3(nb), dim3(256), 0, stream_, (const double*)Z, (const double*)nullptr,
                                   b.n, rpb, zpart);
                hipLaunchKernelGGL(reduce_parts, dim3(1), dim3(64), 0, stream_, (const double*)zpart, nb, (int64_t)PL, 1.0, 0.0,
                                   zsum, PL, (int64_t)PL);
                hipLaunchKernelGGL(rank1_sub, dim3((unsigned)cdiv((int64_t)G * PL, 256)), dim3(256), 0, stream_, Yh, (int64_t)L,
                                   (const double*)mu_.p, (const double*)zsum, coef, G);
                BMX_LAUNCH_CHECK();
            }
        }
    }

    DevBlockCache cache_;
    int device_ = 0, G_ = 0, d_ = 0, L_ = PL;
    hipStream_t stream_ = nullptr;
    std::vector<std::unique_ptr<PcaBatch>> batches_;
    DevBuf<double> mu_, q_, y_, xr_, yr_, w_, qt_, ut_, z_, zt_, part_, small_;
    bool fitted_ = false;
};

Pca* pca_create(int device, int G) { return new Pca(device, G); }
void pca_destroy(Pca* p) { delete p; }
void pca_add_batch(Pca* p, const double* x, int64_t n, double weight, int cos_norm) { p->add_batch(x, n, weight, cos_norm != 0); }
void pca_begin_batch(Pca* p, int64_t n, double weight, int cos_norm) { p->begin_batch(n, weight, cos_norm != 0); }
void pca_add_block(Pca* p, const double* x_block, int64_t m) { p->add_block(x_block, m); }
void pca_fit(Pca* p, int d, double tol, int max_applies, double* centers, double* rotation, double* sdev, int* applies_used,
             double* resid) {
    p->fit(d, tol, max_applies, centers, rotation, sdev, applies_used, resid);
}
void pca_project(Pca* p, int b, double* out) { p->project(b, out); }

}  // namespace bmx
