// Mutual-nearest-neighbour intersection on the device: the arithmetic of src/find_mutual_nns.cpp:8-41 without its
// per-row sort + binary search (rows are <= a few dozen ids, a linear probe of the partner's row is cheaper), plus
// the ordered stream compaction that reproduces the reference's pair order (left ascending, then neighbour rank).
#include "bmx_ops.hpp"

namespace bmx {
namespace {

constexpr int SCAN_ITEMS = 8;                     // per thread
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;      // per block

__global__ __launch_bounds__(256) void scan_blocks(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n,
                                                   int32_t* __restrict__ block_sums) {
    __shared__ int32_t wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * SCAN_BLOCK + tid * SCAN_ITEMS;
    int32_t v[SCAN_ITEMS];
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    int32_t inc = s;  // inclusive scan of per-thread sums across the wave
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    int32_t woff = 0;
    for (int i = 0; i < w; ++i) woff += wave_tot[i];
    int32_t run = woff + inc - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (tid == 255) block_sums[blockIdx.x] = woff + inc;
}

__global__ void scan_block_sums(const int32_t* __restrict__ sums, int32_t* __restrict__ offs, int nblocks,
                                int32_t* __restrict__ total_slot) {
    // one thread: nblocks is n / 2048 (a few thousand at most)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int32_t run = 0;
    for (int b = 0; b < nblocks; ++b) {
        offs[b] = run;
        run += sums[b];
    }
    *total_slot = run;
}

__global__ __launch_bounds__(256) void scan_add_offsets(int32_t* __restrict__ out, int n,
                                                        const int32_t* __restrict__ offs) {
    const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const int32_t o = offs[blockIdx.x];
#pragma unroll
    for (int t = 0; t < SCAN_ITEMS; ++t) {
        const int idx = i + t * 256;
        if (idx < n) out[idx] += o;
    }
}

__device__ __forceinline__ bool row_contains(const int32_t* __restrict__ row, int k, int32_t want) {
    bool f = false;
    for (int t = 0; t < k; ++t) f |= row[t] == want;
    return f;
}

// idxLR may hold the lists of a SUBSET of the left cells (row c = left cell lsel[c], ascending; lpos2c maps a selected
// left cell back to its row).  lsel == nullptr: every left cell has a row (c == l).
__global__ void mutual_left(const int32_t* __restrict__ idxLR, int nL, int k2, const int32_t* __restrict__ idxRL, int k1,
                            const int32_t* __restrict__ lsel, int32_t* __restrict__ cntL,
                            unsigned long long* __restrict__ maskL) {
    const int c0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c0 >= nL) return;
    const int l = lsel ? lsel[c0] : c0;
    int c = 0;
    unsigned long long m = 0;  // bit j: neighbour j is mutual (kept for emit_pairs when k2 <= 64)
    for (int j = 0; j < k2; ++j) {
        const int32_t r = idxLR[(int64_t)c0 * k2 + j];  // -1: the (seeded) search found fewer than k2 cells in reach
        const bool hit = r >= 0 && row_contains(idxRL + (int64_t)r * k1, k1, l);
        c += hit ? 1 : 0;
        if (hit && j < 64) m |= 1ull << j;
    }
    cntL[c0] = c;
    if (maskL) maskL[c0] = m;
}

// position of `want` in row[0, k), -1 if absent
__device__ __forceinline__ int row_find(const int32_t* __restrict__ row, int k, int32_t want) {
    int at = -1;
    for (int t = 0; t < k; ++t) at = row[t] == want ? t : at;
    return at;
}

// One probe serves both sides: right cell r lists left cell l; if l's row lists r at rank j2, the pair is mutual -- r
// keeps l among its partners and bit j2 of l's mask is set (64-bit atomic OR; the masks were zeroed).  With maskL == null
// (k2 > 64) only the right side is written and mutual_left does the left side.
__global__ void mutual_right(const int32_t* __restrict__ idxLR, int k2, const int32_t* __restrict__ idxRL, int nR,
                             int k1, const int32_t* __restrict__ lpos2c, int32_t* __restrict__ partR,
                             int32_t* __restrict__ cntR, unsigned long long* __restrict__ maskL) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nR) return;
    int32_t* row = partR + (int64_t)r * k1;
    int m = 0;
    for (int j = 0; j < k1; ++j) {
        const int32_t l = idxRL[(int64_t)r * k1 + j];
        const int64_t c = lpos2c ? lpos2c[l] : l;
        const int j2 = row_find(idxLR + c * k2, k2, r);
        if (j2 < 0) continue;
        if (maskL) atomicOr(maskL + c, 1ull << j2);
        int p = m++;  // insertion keeps the partners ascending = the order `rowsum` adds them in
        while (p > 0 && row[p - 1] > l) {
            row[p] = row[p - 1];
            --p;
        }
        row[p] = l;
    }
    cntR[r] = m;
}

// The same probe with the k1 <= 32 entries of a right cell's row spread over 32 lanes: the one-thread-per-cell form walks
// its row entry by entry, three dependent loads each (0.18 ms per merge at 100 000 cells, all of it latency).  A lane that
// found its pair takes its place among the row's partners by counting the hits with a smaller left cell.
__global__ __launch_bounds__(256) void mutual_right_wide(const int32_t* __restrict__ idxLR, int k2,
                                                         const int32_t* __restrict__ idxRL, int nR, int k1,
                                                         const int32_t* __restrict__ lpos2c, int32_t* __restrict__ partR,
                                                         int32_t* __restrict__ cntR, unsigned long long* __restrict__ maskL) {
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;
    const bool live = r < nR && j < k1;
    const int32_t l = live ? idxRL[(int64_t)r * k1 + j] : -1;
    bool hit = false;
    if (l >= 0) {
        const int64_t c = lpos2c ? lpos2c[l] : l;
        const int j2 = row_find(idxLR + c * k2, k2, r);
        hit = j2 >= 0;
        if (hit && maskL) atomicOr(maskL + c, 1ull << j2);
    }
    int rank = 0, m = 0;
    for (int t = 0; t < k1; ++t) {  // (k1 is uniform: every lane of the half-wave takes part in the shuffles)
        const int32_t lt = __shfl(l, t, 32);
        const int ht = __shfl((int)hit, t, 32);
        rank += (ht && lt < l) ? 1 : 0;
        m += ht;
    }
    if (hit) partR[(int64_t)r * k1 + rank] = l;  // ascending = the order `rowsum` adds them in
    if (r < nR && j == 0) cntR[r] = m;
}

__global__ void popcount_rows(const unsigned long long* __restrict__ mask, int n, int32_t* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = __popcll(mask[i]);
}

__global__ void emit_pairs_kernel(const int32_t* __restrict__ idxLR, int nL, int k2, const int32_t* __restrict__ idxRL,
                                  int k1, const int32_t* __restrict__ offL, const int32_t* __restrict__ lsel,
                                  const int32_t* __restrict__ lrows, const int32_t* __restrict__ rrows,
                                  const unsigned long long* __restrict__ maskL, int32_t* __restrict__ first,
                                  int32_t* __restrict__ second) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nL) return;
    int o = offL[c];
    if (o == offL[c + 1]) return;  // no pair starts at this left cell
    const int l = lsel ? lsel[c] : c;
    const int32_t lid = (lrows ? lrows[l] : l) + 1;
    const unsigned long long m = maskL ? maskL[c] : 0;
    for (int j = 0; j < k2; ++j) {
        const int32_t r = idxLR[(int64_t)c * k2 + j];
        if (maskL ? ((m >> j) & 1ull) != 0 : (r >= 0 && row_contains(idxRL + (int64_t)r * k1, k1, l))) {
            first[o] = lid;
            second[o] = (rrows ? rrows[r] : r) + 1;
            ++o;
        }
    }
}

__global__ void flag_positive(const int32_t* __restrict__ cnt, int n, int32_t* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = cnt[i] > 0 ? 1 : 0;
}

__global__ void mark_listed(const int32_t* __restrict__ idx, int64_t n, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[idx[i]] = 1;  // same value from every writer: no atomics needed
}

// seed[c] = max over the right cells r that list left cell l (c = row of l among the selected left cells) of
// d(r, l)^2, rounded up to f32: every right cell that can be a mutual partner of l lies within it
__global__ void seed_from_lists(const int32_t* __restrict__ idxRL, const double* __restrict__ distRL, int64_t n,
                                const int32_t* __restrict__ lpos2c, uint32_t* __restrict__ seed_bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dd = distRL[i] * distRL[i] * (1.0 + 1e-15);  // sqrt, then squared again: never below the true value
    float f = (float)dd;
    if ((double)f < dd) f = __uint_as_float(__float_as_uint(f) + 1u);  // dd >= 0: the next float up
    atomicMax(seed_bits + lpos2c[idxRL[i]], __float_as_uint(f));       // non-negative floats order like their bits
}

__global__ void compose_rows(const int32_t* __restrict__ sel, int n, const int32_t* __restrict__ rows,
                             int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rows[sel[i]];
}

__global__ void scatter_positions(const int32_t* __restrict__ flag, const int32_t* __restrict__ off, int n,
                                  int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[off[i]] = i;
}

}  // namespace

void exclusive_scan_i32(hipStream_t stream, ScanWorkspace& ws, const int32_t* in, int32_t* out, int n) {
    const int nb = std::max(1, cdiv(n, SCAN_BLOCK));
    int32_t* sums = ws.block_sums.reserve(nb);
    int32_t* offs = ws.block_offs.reserve(nb);
    hipLaunchKernelGGL(scan_blocks, dim3(nb), dim3(256), 0, stream, in, out, n, sums);
    BMX_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_block_sums, dim3(1), dim3(64), 0, stream, sums, offs, nb, out + n);
    BMX_LAUNCH_CHECK();
    if (nb > 1) {
        hipLaunchKernelGGL(scan_add_offsets, dim3(nb), dim3(256), 0, stream, out, n, offs);
        BMX_LAUNCH_CHECK();
    }
}

void mutual_counts(hipStream_t stream, const int32_t* idxLR, int nL, int k2, const int32_t* idxRL, int nR, int k1,
                   int32_t* cntL, int32_t* partR, int32_t* cntR, const int32_t* lsel, const int32_t* lpos2c,
                   unsigned long long* maskL) {
    // k2 <= 64 (and a mask buffer): the right cells' probe finds every mutual pair once and marks it on both sides;
    // otherwise the left side is probed separately
    const bool fused = maskL != nullptr && k2 <= 64;
    if (fused && nL > 0) BMX_HIP(hipMemsetAsync(maskL, 0, (size_t)nL * sizeof(unsigned long long), stream));
    if (!fused && nL > 0) {
        hipLaunchKernelGGL(mutual_left, dim3(cdiv(nL, 256)), dim3(256), 0, stream, idxLR, nL, k2, idxRL, k1, lsel,
                           cntL, (unsigned long long*)nullptr);
        BMX_LAUNCH_CHECK();
    }
    if (nR > 0) {
        if (k1 <= 32)
            hipLaunchKernelGGL(mutual_right_wide, dim3(cdiv(nR, 8)), dim3(256), 0, stream, idxLR, k2, idxRL, nR, k1, lpos2c,
                               partR, cntR, fused ? maskL : nullptr);
        else
            hipLaunchKernelGGL(mutual_right, dim3(cdiv(nR, 256)), dim3(256), 0, stream, idxLR, k2, idxRL, nR, k1, lpos2c,
                               partR, cntR, fused ? maskL : nullptr);
        BMX_LAUNCH_CHECK();
    }
    if (fused && nL > 0) {
        hipLaunchKernelGGL(popcount_rows, dim3(cdiv(nL, 256)), dim3(256), 0, stream, (const unsigned long long*)maskL, nL,
                           cntL);
        BMX_LAUNCH_CHECK();
    }
}

void emit_pairs(hipStream_t stream, const int32_t* idxLR, int nL, int k2, const int32_t* idxRL, int k1,
                const int32_t* offL, const int32_t* lrows, const int32_t* rrows, int32_t* first, int32_t* second,
                const int32_t* lsel, const unsigned long long* maskL) {
    if (nL <= 0) return;
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(cdiv(nL, 256)), dim3(256), 0, stream, idxLR, nL, k2, idxRL, k1, offL,
                       lsel, lrows, rrows, k2 <= 64 ? maskL : nullptr, first, second);
    BMX_LAUNCH_CHECK();
}

void select_listed_rows(hipStream_t stream, ScanWorkspace& ws, const int32_t* idx, int64_t n_entries, int n_rows,
                        int32_t* flag, int32_t* off, int32_t* sel) {
    BMX_HIP(hipMemsetAsync(flag, 0, (size_t)n_rows * sizeof(int32_t), stream));
    if (n_entries > 0) {
        hipLaunchKernelGGL(mark_listed, dim3(cdiv(n_entries, 256)), dim3(256), 0, stream, idx, n_entries, flag);
        BMX_LAUNCH_CHECK();
    }
    exclusive_scan_i32(stream, ws, flag, off, n_rows);
    hipLaunchKernelGGL(scatter_positions, dim3(cdiv(n_rows, 256)), dim3(256), 0, stream, flag, off, n_rows, sel);
    BMX_LAUNCH_CHECK();
}

void seed_thresholds(hipStream_t stream, const int32_t* idxRL, const double* distRL, int64_t n_entries,
                     const int32_t* lpos2c, int nsel, float* seed) {
    if (nsel <= 0) return;
    BMX_HIP(hipMemsetAsync(seed, 0, (size_t)nsel * sizeof(float), stream));
    if (n_entries > 0) {
        hipLaunchKernelGGL(seed_from_lists, dim3(cdiv(n_entries, 256)), dim3(256), 0, stream, idxRL, distRL, n_entries,
                           lpos2c, reinterpret_cast<uint32_t*>(seed));
        BMX_LAUNCH_CHECK();
    }
}

void compose_row_list(hipStream_t stream, const int32_t* sel, int n, const int32_t* rows, int32_t* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(compose_rows, dim3(cdiv(n, 256)), dim3(256), 0, stream, sel, n, rows, out);
    BMX_LAUNCH_CHECK();
}

void compact_mnn_cells(hipStream_t stream, ScanWorkspace& ws, const int32_t* cntR, int nR, int32_t* flagR,
                       int32_t* offR, int32_t* second_u) {
    if (nR <= 0) return;
    hipLaunchKernelGGL(flag_positive, dim3(cdiv(nR, 256)), dim3(256), 0, stream, cntR, nR, flagR);
    BMX_LAUNCH_CHECK();
    exclusive_scan_i32(stream, ws, flagR, offR, nR);
    hipLaunchKernelGGL(scatter_positions, dim3(cdiv(nR, 256)), dim3(256), 0, stream, flagR, offR, nR, second_u);
    BMX_LAUNCH_CHECK();
}

}  // namespace bmx
