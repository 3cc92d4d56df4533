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

// column sums and column sums of squares of X [n][d] in one pass (overall.batch and the mean squares that
// .get_batch_magnitude wants, R/fastMNN.R:481,588); same two-stage shape as col_reduce
__global__ __launch_bounds__(256) void col_reduce2_partial(const double* __restrict__ X, int n, int d,
                                                           double* __restrict__ partial) {
    __shared__ double sm[4][2][64];
    const int c0 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int b0 = blockIdx.x * RED_ROWS;
    const int b1 = min(n, b0 + RED_ROWS);
    for (int cb = 0; cb < d; cb += 64) {
        const int c = cb + c0;
        double s = 0.0, q = 0.0;
        if (c < d)
            for (int r = b0 + rl; r < b1; r += 4) {
                const double x = X[(int64_t)r * d + c];
                s += x;
                q += x * x;
            }
        sm[rl][0][c0] = s;
        sm[rl][1][c0] = q;
        __syncthreads();
        if (rl == 0 && c < d) {
            partial[(int64_t)blockIdx.x * 2 * d + c] = (sm[0][0][c0] + sm[1][0][c0]) + (sm[2][0][c0] + sm[3][0][c0]);
            partial[(int64_t)blockIdx.x * 2 * d + d + c] = (sm[0][1][c0] + sm[1][1][c0]) + (sm[2][1][c0] + sm[3][1][c0]);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void col_reduce2_final(const double* __restrict__ partial, int nblocks, int d, double scale,
                                                         double* __restrict__ out_sum, double* __restrict__ out_sq) {
    __shared__ double sm[4][64];
    const int c0 = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int which = 0; which < 2; ++which)
        for (int cb = 0; cb < d; cb += 64) {
            const int c = cb + c0;
            double s = 0.0;
            if (c < d)
                for (int b = g; b < nblocks; b += 4) s += partial[(int64_t)b * 2 * d + which * d + c];
            sm[g][c0] = s;
            __syncthreads();
            if (g == 0 && c < d) (which ? out_sq : out_sum)[c] = ((sm[0][c0] + sm[1][c0]) + (sm[2][c0] + sm[3][c0])) * scale;
            __syncthreads();
        }
}

// ---- segments: the original batches of a node (row ranges), at most 16 per launch ------------------------------
struct SegDesc {
    int start[16];
    int n[16];
    int nseg;
};
// per segment: the matrix its rows live in (left and right node of a merge go through ONE launch), the column mean the
// centring of its node works with, and the shift of its one-pass variance (any vector near the segment's mean: its old
// mean where it has statistics, else its first row -- read in place when the pass does not rewrite the rows)
struct SegPtrs {
    double* X[16];
    const double* mu[16];
    const double* pivot[16];
};
struct StatSlots {
    int slot[16];
};

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
    const int32_t* __restrict__ partR, const int32_t* __restrict__ cntR, int k1, double* __restrict__ averaged,
    const int32_t* __restrict__ dup_next) {
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (u >= U) return;
    const int r = second_u[u];
    const double* rv = R + (int64_t)(rrows ? rrows[r] : r) * d;
    for (int c = lane; c < d; c += 64) {
        const double rc = rv[c];
        double s = 0.0;
        int m = 0;
        // (dup_next: the right cell is named by several positions of the restrict list -- its pairs are those of all of them)
        for (int q = r; q >= 0; q = dup_next ? dup_next[q] : -1) {
            const int mq = cntR[q];
            const int32_t* part = partR + (int64_t)q * k1;
            for (int p = 0; p < mq; ++p) {
                const int l = part[p];
                s += L[(int64_t)(lrows ? lrows[l] : l) * d + c] - rc;
            }
            m += mq;
        }
        averaged[(int64_t)u * d + c] = s / (double)m;
    }
}

// The same for an even d <= 64 and k1 <= 32: half a wave per MNN-involved right cell, 16-byte pieces, the partner rows of
// four partners in flight at a time; the sum runs over the partners in ascending order, as above.  The workgroups stride
// over the cells; with `partial` each leaves the column sums and sums of squares of what it wrote ([grid][2][d], combined
// in block order by average_final: overall.batch and the mean squares .get_batch_magnitude wants come out of this very
// pass), and with `srows` the row of every MNN-involved cell in its node (the reference list of the tricube search).
__global__ __launch_bounds__(256) void average_correction_half(
    const double* __restrict__ L, const int32_t* __restrict__ lrows, const double* __restrict__ R,
    const int32_t* __restrict__ rrows, int d, const int32_t* __restrict__ second_u, int U,
    const int32_t* __restrict__ partR, const int32_t* __restrict__ cntR, int k1, double* __restrict__ averaged,
    double* __restrict__ partial, int32_t* __restrict__ srows, int bid0, int nblocks) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ double red[8][2][64];
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int np = d >> 1;
    const bool act = hl < np;
    d2 cs = d2{0.0, 0.0}, cq = d2{0.0, 0.0};
    // (logical workgroup bid of nblocks: a rank of a multi-GPU run launches its share of them, the cells a workgroup takes
    // and the order it adds them in do not depend on how many ranks there are)
    const int bid = bid0 + (int)blockIdx.x;
    for (int u0 = bid * 8; u0 < U; u0 += nblocks * 8) {
        const int u = u0 + (threadIdx.x >> 6) * 2 + half;
        const bool live = u < U;
        const int r = second_u[live ? u : 0];
        const int m = live ? cntR[r] : 0;
        // lane j of the half holds partner j's row
        int64_t prow = 0;
        if (hl < m) {
            const int l = partR[(int64_t)r * k1 + hl];
            prow = lrows ? lrows[l] : l;
        }
        const int64_t rrow = rrows ? rrows[r] : r;
        const d2 rc = act ? reinterpret_cast<const d2*>(R + rrow * d)[hl] : d2{0.0, 0.0};
        d2 s = d2{0.0, 0.0};
        const int mm = max(m, __shfl_xor(m, 32));  // both halves run the same trip count (shuffles stay convergent)
        for (int p0 = 0; p0 < mm; p0 += 4) {
            d2 x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t row = __shfl(prow, (half << 5) + ((p0 + t) & 31));
                x[t] = (act && p0 + t < m) ? reinterpret_cast<const d2*>(L + row * d)[hl] : rc;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (p0 + t < m) {
                    s[0] += x[t][0] - rc[0];
                    s[1] += x[t][1] - rc[1];
                }
        }
        if (live && act) {
            d2 o;
            o[0] = s[0] / (double)m;
            o[1] = s[1] / (double)m;
            reinterpret_cast<d2*>(averaged + (int64_t)u * d)[hl] = o;
            cs[0] += o[0];
            cs[1] += o[1];
            cq[0] += o[0] * o[0];
            cq[1] += o[1] * o[1];
        }
        if (srows && live && hl == 0) srows[u] = (int32_t)rrow;
    }
    if (partial) {
        const int h8 = (threadIdx.x >> 6) * 2 + half;
        if (act) {
            red[h8][0][2 * hl] = cs[0];
            red[h8][0][2 * hl + 1] = cs[1];
            red[h8][1][2 * hl] = cq[0];
            red[h8][1][2 * hl + 1] = cq[1];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * d; e += 256) {
            const int which = e / d, c = e - which * d;
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q][which][c];
            partial[(int64_t)bid * 2 * d + e] = t;
        }
    }
}

// overall.batch = colMeans(averaged) and colMeans(averaged^2) from average_correction_half's partials, then
// .get_batch_magnitude (R/fastMNN.R:582-595) -- one workgroup
__global__ __launch_bounds__(1024) void average_final(const double* __restrict__ partial, int nblocks, int d, double scale,
                                                      double* __restrict__ out_sum, double* __restrict__ out_sq,
                                                      double* __restrict__ magnitude) {
    // 2 d <= 128 columns x 8 thread groups that share each column's partials, combined in a fixed order (deterministic)
    __shared__ double sm[8][128];
    const int e = threadIdx.x & 127, g = threadIdx.x >> 7;
    double s = 0.0;
    if (e < 2 * d) {
        const double* p = partial + e;
#pragma unroll 8
        for (int b = g; b < nblocks; b += 8) s += p[(int64_t)b * 2 * d];
    }
    sm[g][e] = s;
    __syncthreads();
    if (g == 0 && e < 2 * d) {
        const double v = (((sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e])) + ((sm[4][e] + sm[5][e]) + (sm[6][e] + sm[7][e]))) * scale;
        (e < d ? out_sum : out_sq)[e < d ? e : e - d] = v;
        sm[0][e] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0 && magnitude) {
        double l2sq = 0.0, ave = 0.0;
        for (int c = 0; c < d; ++c) {
            ave += sm[0][d + c];
            l2sq += sm[0][c] * sm[0][c];
        }
        *magnitude = ave == 0.0 ? 0.0 : sqrt(l2sq / ave);
    }
}

// one wave per row: tricube weights from the k ascending distances, then the weighted sum of correction vectors
template <bool IN_PLACE>
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
        X[(int64_t)i * d + c] = IN_PLACE ? X[(int64_t)i * d + c] + acc : acc;
    }
}

// The same for k <= 32 and an even d (rows of whole 16-byte pieces): half a wave per row.  Lane j of the half holds
// neighbour j (one coalesced read of the row's k distances and indices), the weights are formed once per row -- the
// normalising sum taken in neighbour order, like rowSums -- and each lane accumulates two columns from 16-byte loads of
// the gathered correction vectors, in neighbour order (the reference's loop over k, R/utils_tricube.R:18-20).
template <bool IN_PLACE>
__global__ __launch_bounds__(256) void tricube_apply_half(double* __restrict__ X, int n, int d,
                                                          const double* __restrict__ averaged,
                                                          const int32_t* __restrict__ idx, const double* __restrict__ dist,
                                                          int k, double ndist) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int i = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    const bool live = i < n;
    const int64_t ib = live ? i : 0;  // (an idle half follows row 0 and stores nothing: the shuffles stay convergent)
    const double dj = hl < k ? dist[ib * k + hl] : 0.0;
    const int ij = hl < k ? idx[ib * k + hl] : 0;
    const int middle = (k + 1) / 2;  // ceiling(k / 2), 1-based
    double bw = __shfl(dj, (half << 5) + middle - 1) * ndist;
    bw = bw < 1e-8 ? 1e-8 : bw;  // pmax(1e-8, bandwidth)
    double rel = dj / bw;
    rel = rel > 1.0 ? 1.0 : rel;
    const double t = 1.0 - rel * rel * rel;
    const double wu = t * t * t;
    double total = 0.0;
    for (int j = 0; j < k; ++j) total += __shfl(wu, (half << 5) + j);
    const double w = wu / total;
    const int np = d >> 1;  // 16-byte pieces per row
    for (int p0 = 0; p0 < np; p0 += 32) {
        const int p = p0 + hl;
        d2 acc = d2{0.0, 0.0};
        for (int j = 0; j < k; ++j) {
            const int src = __shfl(ij, (half << 5) + j);
            const double wj = __shfl(w, (half << 5) + j);
            if (p < np) {
                const d2 a = reinterpret_cast<const d2*>(averaged + (int64_t)src * d)[p];
                acc[0] = acc[0] + a[0] * wj;
                acc[1] = acc[1] + a[1] * wj;
            }
        }
        if (live && p < np) {
            d2* row = reinterpret_cast<d2*>(X + ib * d);
            if (IN_PLACE) {
                const d2 x = row[p];
                acc[0] = x[0] + acc[0];
                acc[1] = x[1] + acc[1];
            }
            row[p] = acc;
        }
    }
}

__global__ void add_scaled_rows_kernel(double* __restrict__ X, int64_t total, int d, const double* __restrict__ corr,
                                       const double* __restrict__ scaling) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    double s = scaling[e / d];
    s = s < 1.0 ? 1.0 : s;  // pmax(scaling, 1): a NaN stays a NaN
    X[e] = X[e] + s * corr[e];
}

// ---------------------------------------------------------------------------------------------------
// Fused row passes.  Every step of the merge that touches all rows of a node is one of
//   * .orthogonalize_other (R/fastMNN.R:642-647): centre along E batch vectors in turn;
//   * .center_along_batch_vector (R/fastMNN.R:626-640) for the new batch vector;
//   * .compute_perbatch_var (R/fastMNN.R:651-658) before and after.
// Centring along v:  x <- x + (mean_restrict(x.v^) - x.v^) v^  =  x - ((x - mu).v^) v^  with mu the column mean over
// the restrict rows -- and mu does not change under the step (the mean of the removed components is zero), so the E
// steps of an orthogonalisation are row-local given ONE mu: a single pass applies them all, where the reference's
// formulation makes 3 E passes (project, mean, shift).  The same pass can accumulate, per original batch (segment),
// shifted column sums and sums of squares of what it writes: mean and sample variance without another pass.
// One wave per row (lanes over the columns, <= 4 per lane), 256 rows per workgroup inside one segment, deterministic
// two-stage reduction of the statistics.
// ---------------------------------------------------------------------------------------------------
constexpr int PASS_ROWS = 512;
constexpr int PASS_EMAX = 8;

struct VecIds {
    int n;
    int id[PASS_EMAX];
};

template <bool APPLY, bool STATS>
__global__ __launch_bounds__(256) void rows_pass(int d, SegDesc sd, SegPtrs sp, const double* __restrict__ vec_pool, VecIds ids,
                                                 int maxnb, double* __restrict__ partial) {
    __shared__ double vhat[APPLY ? PASS_EMAX : 1][256];
    __shared__ double red[STATS ? 4 : 1][2][256];
    const int seg = blockIdx.y;
    double* __restrict__ X = sp.X[seg];
    const double* __restrict__ mu = sp.mu[seg];
    const double* __restrict__ pivots = sp.pivot[seg];
    const int b0 = sd.start[seg] + blockIdx.x * PASS_ROWS;
    const int b1 = min(sd.start[seg] + sd.n[seg], b0 + PASS_ROWS);
    if (blockIdx.x * PASS_ROWS >= sd.n[seg]) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if constexpr (APPLY) {
        // unit vectors of this launch's batch vectors: wave e normalises vector e, e + 4
        for (int e = w; e < ids.n; e += 4) {
            const double* v = vec_pool + (int64_t)ids.id[e] * d;
            double s = 0.0;
            for (int c = lane; c < d; c += 64) s += v[c] * v[c];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const double nrm = sqrt(s);
            for (int c = lane; c < d; c += 64) vhat[e][c] = v[c] / nrm;
        }
        __syncthreads();
    }
    double m_[4], pv[4], s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        m_[i] = (APPLY && c < d) ? mu[c] : 0.0;
        pv[i] = (STATS && c < d) ? pivots[c] : 0.0;
    }
    // two rows per wave and trip: their loads are in flight together
    for (int r = b0 + w; r < b1; r += 8) {
        double x[2][4];
        const bool has2 = r + 4 < b1;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const double* row = X + (int64_t)(rr == 0 || has2 ? r + 4 * rr : r) * d;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + 64 * i;
                x[rr][i] = c < d ? row[c] : 0.0;
            }
        }
        if constexpr (APPLY) {
            for (int e = 0; e < ids.n; ++e) {
                double p0 = 0.0, p1 = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = lane + 64 * i;
                    if (c < d) {
                        p0 += (x[0][i] - m_[i]) * vhat[e][c];
                        p1 += (x[1][i] - m_[i]) * vhat[e][c];
                    }
                }
                for (int o = 32; o > 0; o >>= 1) {
                    p0 += __shfl_xor(p0, o);
                    p1 += __shfl_xor(p1, o);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = lane + 64 * i;
                    if (c < d) {
                        x[0][i] = x[0][i] - p0 * vhat[e][c];
                        x[1][i] = x[1][i] - p1 * vhat[e][c];
                    }
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (rr == 1 && !has2) break;
                double* row = X + (int64_t)(r + 4 * rr) * d;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = lane + 64 * i;
                    if (c < d) row[c] = x[rr][i];
                }
            }
        }
        if constexpr (STATS) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (rr == 1 && !has2) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double t = x[rr][i] - pv[i];
                    s1[i] += t;
                    s2[i] += t * t;
                }
            }
        }
    }
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[w][0][lane + 64 * i] = s1[i];
            red[w][1][lane + 64 * i] = s2[i];
        }
        __syncthreads();
        double* out = partial + ((int64_t)seg * maxnb + blockIdx.x) * 2 * d;
        for (int e = threadIdx.x; e < 2 * d; e += 256) {
            const int which = e / d, c = e - which * d;
            out[e] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
        }
    }
}

// The same pass for an even d <= 128 (rows of whole 16-byte pieces): LPR lanes per row -- 32 (d <= 64: two rows per wave
// instruction) or 64 -- each lane holding one 16-byte piece (two columns), four wave-loads of rows in flight per trip.
// Partial statistics come out in the layout rows_stats_final reads.
template <bool APPLY, bool STATS, int LPR>
__global__ __launch_bounds__(256) void rows_pass_v2(int d, SegDesc sd, SegPtrs sp, const double* __restrict__ vec_pool,
                                                    VecIds ids, int maxnb, double* __restrict__ partial) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int RPW = 64 / LPR;   // rows per wave-load
    constexpr int UNR = 4;          // wave-loads in flight
    __shared__ double vhat[APPLY ? PASS_EMAX : 1][128];
    __shared__ double red[STATS ? 4 * RPW : 1][2][128];
    const int seg = blockIdx.y;
    if (blockIdx.x * PASS_ROWS >= sd.n[seg]) return;
    double* __restrict__ X = sp.X[seg];
    const double* __restrict__ mu = sp.mu[seg];
    const double* __restrict__ pivots = sp.pivot[seg];
    const int b0 = sd.start[seg] + blockIdx.x * PASS_ROWS;
    const int b1 = min(sd.start[seg] + sd.n[seg], b0 + PASS_ROWS);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int hl = lane & (LPR - 1), sub = lane / LPR;  // piece of the row, row of the wave-load
    const int np = d >> 1;
    const bool act = hl < np;
    if constexpr (APPLY) {
        for (int e = w; e < ids.n; e += 4) {  // unit vectors of this launch's batch vectors: wave e normalises vector e, e + 4
            const double* v = vec_pool + (int64_t)ids.id[e] * d;
            double sq = 0.0;
            for (int c = lane; c < d; c += 64) sq += v[c] * v[c];
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            const double nrm = sqrt(sq);
            for (int c = lane; c < d; c += 64) vhat[e][c] = v[c] / nrm;
        }
        __syncthreads();
    }
    d2 m_ = d2{0.0, 0.0}, pv = d2{0.0, 0.0}, s1 = d2{0.0, 0.0}, s2 = d2{0.0, 0.0};
    if (act) {
        if (APPLY) m_ = reinterpret_cast<const d2*>(mu)[hl];
        if (STATS) pv = reinterpret_cast<const d2*>(pivots)[hl];
    }
    for (int r0 = b0 + (w * UNR) * RPW; r0 < b1; r0 += 4 * UNR * RPW) {
        d2 x[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = r0 + u * RPW + sub;
            ok[u] = act && r < b1;
            x[u] = ok[u] ? reinterpret_cast<const d2*>(X + (int64_t)r * d)[hl] : d2{0.0, 0.0};
        }
        if constexpr (APPLY) {
            for (int e = 0; e < ids.n; ++e) {
                const d2 vh = act ? *reinterpret_cast<const d2*>(&vhat[e][2 * hl]) : d2{0.0, 0.0};
                double pr[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) pr[u] = (x[u][0] - m_[0]) * vh[0] + (x[u][1] - m_[1]) * vh[1];
#pragma unroll
                for (int o = LPR / 2; o > 0; o >>= 1) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) pr[u] += __shfl_xor(pr[u], o);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    x[u][0] = x[u][0] - pr[u] * vh[0];
                    x[u][1] = x[u][1] - pr[u] * vh[1];
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (ok[u]) reinterpret_cast<d2*>(X + (int64_t)(r0 + u * RPW + sub) * d)[hl] = x[u];
        }
        if constexpr (STATS) {
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (ok[u]) {
                    const double t0 = x[u][0] - pv[0], t1 = x[u][1] - pv[1];
                    s1[0] += t0;
                    s1[1] += t1;
                    s2[0] += t0 * t0;
                    s2[1] += t1 * t1;
                }
        }
    }
    if constexpr (STATS) {
        if (act) {
            red[w * RPW + sub][0][2 * hl] = s1[0];
            red[w * RPW + sub][0][2 * hl + 1] = s1[1];
            red[w * RPW + sub][1][2 * hl] = s2[0];
            red[w * RPW + sub][1][2 * hl + 1] = s2[1];
        }
        __syncthreads();
        double* out = partial + ((int64_t)seg * maxnb + blockIdx.x) * 2 * d;
        for (int e = threadIdx.x; e < 2 * d; e += 256) {
            const int which = e / d, c = e - which * d;
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 4 * RPW; ++q) t += red[q][which][c];
            out[e] = t;
        }
    }
}

// per segment: column means and the sum over columns of the sample variance.  Four thread groups share each column's
// partials and are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void rows_stats_final(const double* __restrict__ partial, int maxnb, int d, SegDesc sd,
                                                        SegPtrs sp, StatSlots slots, double* __restrict__ means_pool,
                                                        double* __restrict__ scal) {
    __shared__ double sa[4][64], sb[4][64];
    __shared__ double sacc[64];
    const int seg = blockIdx.x;
    const int n = sd.n[seg];
    const double* __restrict__ pivots = sp.pivot[seg];
    const int nb = (n + PASS_ROWS - 1) / PASS_ROWS;
    const int c0 = threadIdx.x & 63, g = threadIdx.x >> 6;
    double acc = 0.0;
    for (int cb = 0; cb < d; cb += 64) {
        const int c = cb + c0;
        double a = 0.0, b = 0.0;
        if (c < d) {
            const double* p = partial + (int64_t)seg * maxnb * 2 * d + c;
#pragma unroll 4
            for (int blk = g; blk < nb; blk += 4) {
                a += p[(int64_t)blk * 2 * d];
                b += p[(int64_t)blk * 2 * d + d];
            }
        }
        sa[g][c0] = a;
        sb[g][c0] = b;
        __syncthreads();
        if (g == 0 && c < d) {
            a = (sa[0][c0] + sa[1][c0]) + (sa[2][c0] + sa[3][c0]);
            b = (sb[0][c0] + sb[1][c0]) + (sb[2][c0] + sb[3][c0]);
            means_pool[(int64_t)slots.slot[seg] * d + c] = pivots[c] + a / (double)n;
            acc += (b - a * a / (double)n) / (double)(n - 1);  // one cell: 0 / 0 = NaN, as colVars gives NA
        }
        __syncthreads();
    }
    if (g == 0) sacc[c0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 64; ++i) t += sacc[i];
        scal[slots.slot[seg]] = t;
    }
}

// mu = sum_s n_s mean_s / sum_s n_s over the segments of a node (their means are current).  blockIdx.y picks the node:
// segments [0, split) -> mu0, [split, nseg) -> mu1 (both nodes of a merge in one launch)
__global__ void combine_means(const double* __restrict__ means_pool, SegDesc sd, StatSlots slots, int d, int split,
                              double* __restrict__ mu0, double* __restrict__ mu1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const int i0 = blockIdx.y == 0 ? 0 : split, i1 = blockIdx.y == 0 ? split : sd.nseg;
    if (i1 <= i0) return;
    double s = 0.0, n = 0.0;
    for (int i = i0; i < i1; ++i) {
        s += (double)sd.n[i] * means_pool[(int64_t)slots.slot[i] * d + c];
        n += (double)sd.n[i];
    }
    (blockIdx.y == 0 ? mu0 : mu1)[c] = s / n;
}

// .get_batch_magnitude (R/fastMNN.R:582-595): sqrt(sum(ave^2) / sum(colMeans(correction^2))), 0 if the latter is 0
__global__ void batch_magnitude_kernel(const double* __restrict__ overall, const double* __restrict__ msq, int d,
                                       double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double l2sq = 0.0, ave = 0.0;
    for (int c = 0; c < d; ++c) {
        ave += msq[c];
        l2sq += overall[c] * overall[c];
    }
    *out = ave == 0.0 ? 0.0 : sqrt(l2sq / ave);
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

void col_reduce2(hipStream_t stream, ReduceWorkspace& ws, const double* X, int n, int d, double scale, double* out_sum,
                 double* out_sq) {
    const int nb = std::max(1, cdiv(n, RED_ROWS));
    double* partial = ws.partial.reserve((size_t)nb * 2 * d);
    hipLaunchKernelGGL(col_reduce2_partial, dim3(nb), dim3(256), 0, stream, X, n, d, partial);
    BMX_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_reduce2_final, dim3(1), dim3(256), 0, stream, (const double*)partial, nb, d, scale, out_sum, out_sq);
    BMX_LAUNCH_CHECK();
}

void sum_vector(hipStream_t stream, const double* in, int d, double scale, double* out) {
    hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(64), 0, stream, in, d, scale, out);
    BMX_LAUNCH_CHECK();
}

bool average_correction(hipStream_t stream, ReduceWorkspace& ws, const double* L, const int32_t* lrows, const double* R,
                        const int32_t* rrows, int d, const int32_t* second_u, int U, const int32_t* partR, const int32_t* cntR,
                        int k1, double* averaged, bool with_sums, double* overall, double* msq, double* magnitude,
                        int32_t* srows, const int32_t* dup_next, const AvgShard* shard) {
    if (U <= 0) return false;
    if ((d & 1) == 0 && d <= 64 && k1 <= 32 && !dup_next) {
        const int grid = std::min(cdiv(U, 8), 1024);
        if (with_sums && shard && !srows) {
            // Several ranks, the averaging whose VECTORS nobody reads (R/fastMNN.R:480-481 wants overall.batch and the
            // magnitude only; the vectors are taken again after the centring): every rank runs its share of the workgroups and
            // the workgroups' column sums -- [grid][2 d] doubles -- are all-gathered; average_final adds them in block order as
            // ever, so the result has the single rank's bits for any number of ranks.
            const int per = cdiv(grid, shard->world);
            const int lo = std::min(grid, per * shard->rank), hi = std::min(grid, lo + per);
            double* partial = ws.partial.reserve((size_t)per * shard->world * 2 * d);
            if (hi > lo) {
                hipLaunchKernelGGL(average_correction_half, dim3(hi - lo), dim3(256), 0, stream, L, lrows, R, rrows, d, second_u, U,
                                   partR, cntR, k1, averaged, partial, srows, lo, grid);
                BMX_LAUNCH_CHECK();
            }
            shard->exchange(partial, (int64_t)per * 2 * d * (int64_t)sizeof(double));
            hipLaunchKernelGGL(average_final, dim3(1), dim3(1024), 0, stream, (const double*)partial, grid, d, 1.0 / (double)U,
                               overall, msq, magnitude);
            BMX_LAUNCH_CHECK();
            return true;
        }
        double* partial = with_sums ? ws.partial.reserve((size_t)grid * 2 * d) : nullptr;
        hipLaunchKernelGGL(average_correction_half, dim3(grid), dim3(256), 0, stream, L, lrows, R, rrows, d, second_u, U, partR,
                           cntR, k1, averaged, partial, srows, 0, grid);
        BMX_LAUNCH_CHECK();
        if (with_sums) {
            hipLaunchKernelGGL(average_final, dim3(1), dim3(1024), 0, stream, (const double*)partial, grid, d, 1.0 / (double)U,
                               overall, msq, magnitude);
            BMX_LAUNCH_CHECK();
        }
        return true;
    }
    hipLaunchKernelGGL(average_correction_kernel, dim3(cdiv(U, 4)), dim3(256), 0, stream, L, lrows, R, rrows, d, second_u, U,
                       partR, cntR, k1, averaged, dup_next);
    BMX_LAUNCH_CHECK();
    return false;  // (the caller takes the column sums / the row list in passes of their own)
}

void tricube_apply(hipStream_t stream, double* X, int n, int d, const double* averaged, const int32_t* idx,
                   const double* dist, int k, double ndist) {
    if (n <= 0 || k <= 0) return;  // k == 0: the weighted correction is all zeros (R/utils_tricube.R:22-23)
    if (k <= 32 && (d & 1) == 0)
        hipLaunchKernelGGL(tricube_apply_half<true>, dim3(cdiv(n, 8)), dim3(256), 0, stream, X, n, d, averaged, idx, dist, k,
                           ndist);
    else
        hipLaunchKernelGGL(tricube_apply_kernel<true>, dim3(cdiv(n, 4)), dim3(256), 0, stream, X, n, d, averaged, idx, dist, k,
                           ndist);
    BMX_LAUNCH_CHECK();
}

void tricube_vectors(hipStream_t stream, int n, int d, const double* averaged, const int32_t* idx, const double* dist,
                     int k, double ndist, double* correction) {
    if (n <= 0) return;
    if (k <= 0) {
        BMX_HIP(hipMemsetAsync(correction, 0, (size_t)n * d * sizeof(double), stream));
        return;
    }
    if (k <= 32 && (d & 1) == 0)
        hipLaunchKernelGGL(tricube_apply_half<false>, dim3(cdiv(n, 8)), dim3(256), 0, stream, correction, n, d, averaged, idx,
                           dist, k, ndist);
    else
        hipLaunchKernelGGL(tricube_apply_kernel<false>, dim3(cdiv(n, 4)), dim3(256), 0, stream, correction, n, d, averaged,
                           idx, dist, k, ndist);
    BMX_LAUNCH_CHECK();
}

void add_scaled_rows(hipStream_t stream, double* X, int n, int d, const double* correction, const double* scaling) {
    const int64_t total = (int64_t)n * d;
    if (total <= 0) return;
    hipLaunchKernelGGL(add_scaled_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, X, total, d,
                       correction, scaling);
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


// ---- fused row passes (host side) -------------------------------------------------------------------
void rows_multi(hipStream_t stream, ReduceWorkspace& ws, int d, const RowSeg* segs, int nseg, const double* vec_pool,
                const int* vec_ids, int nvec, bool stats, double* means_pool, double* scal) {
    if (d > 256) throw Error(BMX_ERR_ARG, "more than 256 dimensions are not supported");
    if (nvec == 0 && !stats) return;
    for (int s0 = 0; s0 < nseg; s0 += 16) {
        SegDesc sd;
        StatSlots sl;
        SegPtrs sp;
        sd.nseg = std::min(16, nseg - s0);
        int maxn = 1;
        // a segment without a pivot takes its first row: read in place by a pass that leaves the rows alone, copied out
        // first (d doubles) by one that rewrites them
        double* pivcopy = nullptr;
        for (int i = 0; i < sd.nseg; ++i) {
            const RowSeg& g = segs[s0 + i];
            sd.start[i] = g.start;
            sd.n[i] = g.n;
            sl.slot[i] = g.slot;
            sp.X[i] = g.X;
            sp.mu[i] = g.mu;
            sp.pivot[i] = g.pivot;
            maxn = std::max(maxn, g.n);
        }
        const int maxnb = cdiv(maxn, PASS_ROWS);
        double* partial = nullptr;
        if (stats) {
            partial = ws.partial.reserve((size_t)sd.nseg * maxnb * 2 * d + (size_t)16 * d);
            pivcopy = partial + (size_t)sd.nseg * maxnb * 2 * d;
            for (int i = 0; i < sd.nseg; ++i) {
                if (sp.pivot[i]) continue;
                const double* first_row = sp.X[i] + (size_t)sd.start[i] * d;
                if (nvec > 0) {
                    BMX_HIP(hipMemcpyAsync(pivcopy + (size_t)i * d, first_row, (size_t)d * sizeof(double), hipMemcpyDeviceToDevice,
                                           stream));
                    sp.pivot[i] = pivcopy + (size_t)i * d;
                } else {
                    sp.pivot[i] = first_row;
                }
            }
        }
        const dim3 grid(maxnb, sd.nseg);
        // batch vectors in launches of PASS_EMAX; the statistics ride on the last one (they describe the final rows)
        int e0 = 0;
        do {
            VecIds ids;
            ids.n = std::min(PASS_EMAX, nvec - e0);
            for (int e = 0; e < ids.n; ++e) ids.id[e] = vec_ids[e0 + e];
            const bool last = e0 + ids.n >= nvec;
            const int form = (d & 1) || d > 128 ? 0 : (d <= 64 ? 32 : 64);  // lanes per row of the 16-byte form
#define BMX_ROWS_PASS(A, S, PART)                                                                                              \
    do {                                                                                                                       \
        if (form == 32)                                                                                                        \
            hipLaunchKernelGGL((rows_pass_v2<A, S, 32>), grid, dim3(256), 0, stream, d, sd, sp, vec_pool, ids, maxnb, PART);   \
        else if (form == 64)                                                                                                   \
            hipLaunchKernelGGL((rows_pass_v2<A, S, 64>), grid, dim3(256), 0, stream, d, sd, sp, vec_pool, ids, maxnb, PART);   \
        else                                                                                                                   \
            hipLaunchKernelGGL((rows_pass<A, S>), grid, dim3(256), 0, stream, d, sd, sp, vec_pool, ids, maxnb, PART);          \
    } while (0)
            if (stats && last) {
                if (ids.n > 0)
                    BMX_ROWS_PASS(true, true, partial);
                else
                    BMX_ROWS_PASS(false, true, partial);
                hipLaunchKernelGGL(rows_stats_final, dim3(sd.nseg), dim3(256), 0, stream, partial, maxnb, d, sd, sp, sl, means_pool,
                                   scal);
            } else if (ids.n > 0) {
                BMX_ROWS_PASS(true, false, (double*)nullptr);
            }
#undef BMX_ROWS_PASS
            BMX_LAUNCH_CHECK();
            e0 += ids.n;
        } while (e0 < nvec);
    }
}

void rows_apply_stats(hipStream_t stream, ReduceWorkspace& ws, double* X, int d, const int* starts, const int* ns,
                      int nseg, const double* mu, const double* vec_pool, const int* vec_ids, int nvec,
                      const int* stat_slots, double* means_pool, double* scal) {
    std::vector<RowSeg> segs((size_t)nseg);
    for (int i = 0; i < nseg; ++i) segs[i] = RowSeg{X, starts[i], ns[i], mu, nullptr, stat_slots ? stat_slots[i] : -1};
    rows_multi(stream, ws, d, segs.data(), nseg, vec_pool, vec_ids, nvec, stat_slots != nullptr, means_pool, scal);
}

void node_means_from_segments(hipStream_t stream, const double* means_pool, const int* ns, const int* slots, int nseg0, int nseg1,
                              int d, double* mu0, double* mu1) {
    if (nseg0 + nseg1 > 16) throw Error(BMX_ERR_ARG, "node_means_from_segments: more than 16 segments");
    SegDesc sd;
    StatSlots sl;
    sd.nseg = nseg0 + nseg1;
    for (int i = 0; i < sd.nseg; ++i) {
        sd.start[i] = 0;
        sd.n[i] = ns[i];
        sl.slot[i] = slots[i];
    }
    hipLaunchKernelGGL(combine_means, dim3(cdiv(d, 64), nseg1 > 0 ? 2 : 1), dim3(64), 0, stream, means_pool, sd, sl, d, nseg0, mu0,
                       mu1);
    BMX_LAUNCH_CHECK();
}

void node_mean_from_segments(hipStream_t stream, const double* means_pool, const int* ns, const int* slots, int nseg,
                             int d, double* mu) {
    node_means_from_segments(stream, means_pool, ns, slots, nseg, 0, d, mu, nullptr);
}

void batch_magnitude(hipStream_t stream, const double* overall, const double* msq, int d, double* out) {
    hipLaunchKernelGGL(batch_magnitude_kernel, dim3(1), dim3(64), 0, stream, overall, msq, d, out);
    BMX_LAUNCH_CHECK();
}

}  // namespace bmx
