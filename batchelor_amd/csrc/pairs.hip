// Mutual-nearest-neighbour intersection on the device: the arithmetic of src/find_mutual_nns.cpp:8-41 without its
// per-row sort + binary search (rows are <= a few dozen ids, a linear probe of the partner's row is cheaper), plus
// the ordered stream compaction that reproduces the reference's pair order (left ascending, then neighbour rank).
#include "bmx_ops.hpp"
#include "scan_lookback.hpp"

namespace bmx {
namespace {

// ---- compactions as single launches (scan_lookback.hpp) -------------------------------------------------------------
// One list: value(i) in {0, 1, ...}; off[i] = exclusive prefix, off[n] = total; optionally the positions of the non-zero
// entries (sel), and per selected position the reset of what the next steps accumulate into (seed, mask).
// MODE 0: value = (stamp[i] == gen)   (rows some neighbour list names: mark_listed stamps them with the search's number)
// MODE 1: value = popcount(mask[i])   (pairs per selected left cell)
// MODE 2: value = in[i]               (counts)
// MODE 3: value = in[i] > 0           (right cells with at least one pair)
template <int MODE>
__device__ __forceinline__ void compact_block(const scan::Chain ch, int bid, int nblocks, const int32_t* __restrict__ in,
                                              const unsigned long long* __restrict__ mask, int gen, int n,
                                              int32_t* __restrict__ off, int32_t* __restrict__ sel,
                                              int32_t* __restrict__ total_out, uint32_t* __restrict__ zero32,
                                              unsigned long long* __restrict__ zero64, unsigned int* sh4) {
    const int base = bid * scan::BLOCK + threadIdx.x * scan::ITEMS;
    unsigned int v[scan::ITEMS], mine = 0;
#pragma unroll
    for (int i = 0; i < scan::ITEMS; ++i) {
        unsigned int x = 0;
        if (base + i < n) {
            if (MODE == 0) x = in[base + i] == gen ? 1u : 0u;
            if (MODE == 1) x = (unsigned int)__popcll(mask[base + i]);
            if (MODE == 2) x = (unsigned int)in[base + i];
            if (MODE == 3) x = in[base + i] > 0 ? 1u : 0u;
        }
        v[i] = x;
        mine += x;
    }
    unsigned int total = 0;
    bool last = false;
    unsigned int run = scan::exclusive_prefix(ch, bid, nblocks, mine, sh4, &total, &last);
#pragma unroll
    for (int i = 0; i < scan::ITEMS; ++i) {
        if (base + i < n) {
            if (off) off[base + i] = (int32_t)run;
            if ((MODE == 0 || MODE == 3) && v[i]) {
                if (sel) sel[run] = base + i;
                if (zero32) zero32[run] = 0u;
                if (zero64) zero64[run] = 0ull;
            }
        }
        run += v[i];
    }
    if (last && threadIdx.x == 0) {
        if (off) off[n] = (int32_t)total;
        if (total_out) *total_out = (int32_t)total;
    }
}

__global__ __launch_bounds__(256) void select_listed_kernel(scan::Chain ch, const int32_t* __restrict__ stamp, int gen, int n,
                                                            int32_t* __restrict__ off, int32_t* __restrict__ sel,
                                                            int32_t* __restrict__ total_out, uint32_t* __restrict__ seed,
                                                            unsigned long long* __restrict__ mask) {
    __shared__ unsigned int sh4[8];
    __shared__ int sh_b;
    const int bid = scan::take_ticket(ch, &sh_b);
    compact_block<0>(ch, bid, gridDim.x, stamp, nullptr, gen, n, off, sel, total_out, seed, mask, sh4);
}

// Two lists in one launch: workgroups [0, nbA) scan the pairs per selected left cell (popcount of its mask, or cntL where
// k2 > 64 has no masks), workgroups [nbA, nbA + nbB) compact the right cells that have a pair.
__global__ __launch_bounds__(256) void pair_scans_kernel(scan::Chain chA, int nbA, const unsigned long long* __restrict__ maskL,
                                                         const int32_t* __restrict__ cntL, int nsel, int32_t* __restrict__ offL,
                                                         int32_t* __restrict__ totalP, scan::Chain chB, int nbB,
                                                         const int32_t* __restrict__ cntR, int nR, int32_t* __restrict__ offR,
                                                         int32_t* __restrict__ second_u, int32_t* __restrict__ totalU) {
    __shared__ unsigned int sh4[8];
    __shared__ int sh_b;
    if ((int)blockIdx.x < nbA) {  // (physical block ranges only pick the list; logical numbers come from the list's ticket)
        const int bid = scan::take_ticket(chA, &sh_b);
        if (maskL)
            compact_block<1>(chA, bid, nbA, nullptr, maskL, 0, nsel, offL, nullptr, totalP, nullptr, nullptr, sh4);
        else
            compact_block<2>(chA, bid, nbA, cntL, nullptr, 0, nsel, offL, nullptr, totalP, nullptr, nullptr, sh4);
    } else {
        const int bid = scan::take_ticket(chB, &sh_b);
        compact_block<3>(chB, bid, nbB, cntR, nullptr, 0, nR, offR, second_u, totalU, nullptr, nullptr, sh4);
    }
}

__global__ __launch_bounds__(256) void scan_counts_kernel(scan::Chain ch, const int32_t* __restrict__ in, int n,
                                                          int32_t* __restrict__ off) {
    __shared__ unsigned int sh4[8];
    __shared__ int sh_b;
    const int bid = scan::take_ticket(ch, &sh_b);
    compact_block<2>(ch, bid, gridDim.x, in, nullptr, 0, n, off, nullptr, nullptr, nullptr, nullptr, sh4);
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
                                                         int32_t* __restrict__ cntR, unsigned long long* __restrict__ maskL,
                                                         const double* __restrict__ distRL, const double* __restrict__ kthL) {
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;
    const bool live = r < nR && j < k1;
    const int32_t l = live ? idxRL[(int64_t)r * k1 + j] : -1;
    bool hit = false;
    if (l >= 0) {
        const int64_t c = lpos2c ? lpos2c[l] : l;
        // the left cell's row holds its k2 nearest right cells, the farthest at kthL[c]: a right cell farther than that is
        // not in it -- 8 bytes read instead of the row's 4 k2 for the ~93 % of the probes that fail (the distance is the same
        // double from either side: the FP64 sum is symmetric bit for bit).  At or inside it (ties included): the row decides.
        if (!(kthL && distRL[(int64_t)r * k1 + j] > kthL[c])) {
            const int j2 = row_find(idxLR + c * k2, k2, r);
            hit = j2 >= 0;
            if (hit && maskL) atomicOr(maskL + c, 1ull << j2);
        }
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

// ---- k2 > 64 (prop.k runs: k = 1 000 at 100 000 cells) -------------------------------------------------------------
// The probes above read a whole row per question: nL * k2 * k1 words, 0.83 s of a 1.7 s step at k = 1 000.  The reference
// sorts nothing either but asks a per-cell set (src/find_mutual_nns.cpp:23-36); here every row is sorted once
// (sort_rows_kernel) and a question is a binary search.  One wave per row; the hits of 64 entries are ranked by a ballot,
// so a right cell's partners come out ascending (its sorted row is walked in order) and a left cell's pairs in neighbour
// rank order (its row is walked as the search wrote it) -- the orders the linear kernels produce.
__global__ __launch_bounds__(256) void sort_rows_kernel(const int32_t* __restrict__ in, int n_rows, int k, int np2,
                                                        int32_t* __restrict__ out) {
    extern __shared__ int32_t sh_row[];
    for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
        for (int i = threadIdx.x; i < np2; i += 256) sh_row[i] = i < k ? in[(int64_t)row * k + i] : INT32_MAX;
        __syncthreads();
        for (int size = 2; size <= np2; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = threadIdx.x; t < (np2 >> 1); t += 256) {
                    const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                    const int32_t a = sh_row[i], b = sh_row[j];
                    if ((a > b) == ((i & size) == 0)) {
                        sh_row[i] = b;
                        sh_row[j] = a;
                    }
                }
                __syncthreads();
            }
        for (int i = threadIdx.x; i < k; i += 256) out[(int64_t)row * k + i] = sh_row[i];
        __syncthreads();
    }
}

__device__ __forceinline__ bool sorted_contains(const int32_t* __restrict__ row, int k, int32_t want) {
    int lo = 0, hi = k;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    return lo < k && row[lo] == want;
}

__global__ __launch_bounds__(256) void mutual_left_sorted(const int32_t* __restrict__ idxLR, int nL, int k2,
                                                          const int32_t* __restrict__ sortedRL, int k1,
                                                          const int32_t* __restrict__ lsel, int32_t* __restrict__ cntL,
                                                          unsigned long long* __restrict__ hits) {
    const int c0 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c0 >= nL) return;
    const int l = lsel ? lsel[c0] : c0;
    const int W = (k2 + 63) >> 6;
    int c = 0;
    for (int base = 0; base < k2; base += 64) {
        const int j = base + lane;
        const int32_t r = j < k2 ? idxLR[(int64_t)c0 * k2 + j] : -1;
        const bool hit = r >= 0 && sorted_contains(sortedRL + (int64_t)r * k1, k1, l);
        const unsigned long long b = __ballot(hit);
        c += __popcll(b);
        if (lane == 0) hits[(int64_t)c0 * W + (base >> 6)] = b;  // (bit j of word base / 64: neighbour base + j is mutual -- for emit_pairs)
    }
    if (lane == 0) cntL[c0] = c;
}

__global__ __launch_bounds__(256) void mutual_right_sorted(const int32_t* __restrict__ sortedLR, int k2,
                                                           const int32_t* __restrict__ sortedRL, int nR, int k1,
                                                           const int32_t* __restrict__ lpos2c, int32_t* __restrict__ partR,
                                                           int32_t* __restrict__ cntR) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= nR) return;
    int m = 0;
    for (int base = 0; base < k1; base += 64) {
        const int j = base + lane;
        const int32_t l = j < k1 ? sortedRL[(int64_t)r * k1 + j] : -1;
        bool hit = false;
        if (l >= 0) {
            const int64_t c = lpos2c ? lpos2c[l] : l;
            hit = sorted_contains(sortedLR + c * k2, k2, r);
        }
        const unsigned long long b = __ballot(hit);
        if (hit) partR[(int64_t)r * k1 + m + __popcll(b & ((1ull << lane) - 1ull))] = l;  // ascending left cells
        m += __popcll(b);
    }
    if (lane == 0) cntR[r] = m;
}

__global__ __launch_bounds__(256) void emit_pairs_sorted(const int32_t* __restrict__ idxLR, int nL, int k2,
                                                         const unsigned long long* __restrict__ hits,
                                                         const int32_t* __restrict__ offL, const int32_t* __restrict__ lsel,
                                                         const int32_t* __restrict__ lrows, const int32_t* __restrict__ rrows,
                                                         int32_t* __restrict__ first, int32_t* __restrict__ second) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= nL) return;
    int o = offL[c];
    if (o == offL[c + 1]) return;  // no pair starts at this left cell (uniform over the wave)
    const int l = lsel ? lsel[c] : c;
    const int32_t lid = (lrows ? lrows[l] : l) + 1;
    const int W = (k2 + 63) >> 6;
    for (int base = 0; base < k2; base += 64) {
        const int j = base + lane;
        const unsigned long long b = hits[(int64_t)c * W + (base >> 6)];  // (what mutual_left_sorted found: no second search)
        const bool hit = (b >> lane) & 1ull;
        const int32_t r = hit ? idxLR[(int64_t)c * k2 + j] : -1;
        if (hit) {
            const int at = o + __popcll(b & ((1ull << lane) - 1ull));
            first[at] = lid;
            second[at] = (rrows ? rrows[r] : r) + 1;
        }
        o += __popcll(b);
    }
}

// stamp[row] = gen for every row some list names (the stamps are never cleared: each search brings its own number)
__global__ void mark_listed(const int32_t* __restrict__ idx, int64_t n, int32_t* __restrict__ stamp, int gen) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stamp[idx[i]] = gen;  // same value from every writer: no atomics needed
}

// seed[c] = max over the right cells r that list left cell l (c = row of l among the selected left cells) of
// d(r, l)^2, rounded up to f32: every right cell that can be a mutual partner of l lies within it
// Several ranks: a rank searches its slice of the selected left cells only (bmx_shard_range over their number, which is
// still on the device here: nsel_dev), so it takes the seeds of that slice only -- 7 of 8 entries end after one lookup.
__global__ void seed_from_lists(const int32_t* __restrict__ idxRL, const double* __restrict__ distRL, int64_t n,
                                const int32_t* __restrict__ lpos2c, uint32_t* __restrict__ seed_bits,
                                const int32_t* __restrict__ nsel_dev, int rank, int world) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t c_ = lpos2c[idxRL[i]];
    if (world > 1) {
        const int nsel = *nsel_dev, per = (nsel + world - 1) / world;
        const int lo = min(nsel, per * rank), hi = min(nsel, lo + per);
        if (c_ < lo || c_ >= hi) return;
    }
    const double dd = distRL[i] * distRL[i] * (1.0 + 1e-15);  // sqrt, then squared again: never below the true value
    float f = (float)dd;
    if ((double)f < dd) f = __uint_as_float(__float_as_uint(f) + 1u);  // dd >= 0: the next float up
    // non-negative floats order like their bits.  A cell is listed ~25 times at the first merge; only the few entries that
    // raise its running maximum need the atomic (a stale read merely sends one too many)
    uint32_t* slot = seed_bits + c_;
    const uint32_t bits = __float_as_uint(f);
    if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
}

__global__ void compose_rows(const int32_t* __restrict__ sel, int n, const int32_t* __restrict__ rows,
                             int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rows[sel[i]];
}

}  // namespace

scan::Chain ScanWorkspace::chain(hipStream_t stream, int which, int nblocks) {
    if (!ticket.p) {
        ticket.reserve(8);
        BMX_HIP(hipMemsetAsync(ticket.p, 0, 8 * sizeof(unsigned int), stream));
    }
    if ((size_t)nblocks > chain_cap) {  // both chains grow together (stale words carry old epochs: harmless, but a fresh
        chain_cap = (size_t)nblocks + 64;  // block holds anything -- clear it once)
        status.reserve(2 * chain_cap);
        BMX_HIP(hipMemsetAsync(status.p, 0, 2 * chain_cap * sizeof(unsigned long long), stream));
    }
    epoch = epoch >= 0x3FFFFFF0u ? 1u : epoch + 1u;
    return scan::Chain{status.p + (size_t)which * chain_cap, ticket.p + which, epoch};
}

void exclusive_scan_i32(hipStream_t stream, ScanWorkspace& ws, const int32_t* in, int32_t* out, int n) {
    const int nb = std::max(1, cdiv(n, scan::BLOCK));
    hipLaunchKernelGGL(scan_counts_kernel, dim3(nb), dim3(256), 0, stream, ws.chain(stream, 0, nb), in, n, out);
    BMX_LAUNCH_CHECK();
}

bool sorted_rows_apply(int k1, int k2) { return k2 > 64 && k2 <= 8192 && k1 <= 8192; }

// rows of `in` [n_rows][k], each sorted ascending (-1 = "no neighbour" first), into buf
static const int32_t* sort_rows(hipStream_t stream, const int32_t* in, int n_rows, int k, DevBuf<int32_t>& buf) {
    int32_t* out = buf.reserve(std::max<size_t>(1, (size_t)n_rows * k));
    if (n_rows <= 0) return out;
    int np2 = 2;
    while (np2 < k) np2 <<= 1;
    hipLaunchKernelGGL(sort_rows_kernel, dim3(std::min(n_rows, 1 << 16)), dim3(256), (size_t)np2 * sizeof(int32_t), stream, in,
                       n_rows, k, np2, out);
    BMX_LAUNCH_CHECK();
    return out;
}

void mutual_counts(hipStream_t stream, const int32_t* idxLR, int nL, int k2, const int32_t* idxRL, int nR, int k1,
                   int32_t* cntL, int32_t* partR, int32_t* cntR, const int32_t* lsel, const int32_t* lpos2c,
                   unsigned long long* maskL, bool mask_is_clear, const double* distRL, const double* kthL,
                   SortedRows* sorted) {
    // k2 <= 64 (and a mask buffer): the right cells' probe finds every mutual pair once and marks it on both sides (the
    // pairs of a left cell are then the popcount of its mask); otherwise the left side is probed separately into cntL
    const bool fused = maskL != nullptr && k2 <= 64;
    if (sorted && sorted_rows_apply(k1, k2)) {
        const int32_t* sLR = sort_rows(stream, idxLR, nL, k2, sorted->lr);
        const int32_t* sRL = sort_rows(stream, idxRL, nR, k1, sorted->rl);
        if (nL > 0) {
            unsigned long long* hits = sorted->hits.reserve((size_t)nL * ((k2 + 63) / 64));
            hipLaunchKernelGGL(mutual_left_sorted, dim3(cdiv(nL, 4)), dim3(256), 0, stream, idxLR, nL, k2, sRL, k1, lsel, cntL, hits);
            BMX_LAUNCH_CHECK();
        }
        if (nR > 0) {
            hipLaunchKernelGGL(mutual_right_sorted, dim3(cdiv(nR, 4)), dim3(256), 0, stream, sLR, k2, sRL, nR, k1, lpos2c,
                               partR, cntR);
            BMX_LAUNCH_CHECK();
        }
        return;
    }
    if (fused && nL > 0 && !mask_is_clear) BMX_HIP(hipMemsetAsync(maskL, 0, (size_t)nL * sizeof(unsigned long long), stream));
    if (!fused && nL > 0) {
        hipLaunchKernelGGL(mutual_left, dim3(cdiv(nL, 256)), dim3(256), 0, stream, idxLR, nL, k2, idxRL, k1, lsel,
                           cntL, (unsigned long long*)nullptr);
        BMX_LAUNCH_CHECK();
    }
    if (nR > 0) {
        if (k1 <= 32)
            hipLaunchKernelGGL(mutual_right_wide, dim3(cdiv(nR, 8)), dim3(256), 0, stream, idxLR, k2, idxRL, nR, k1, lpos2c,
                               partR, cntR, fused ? maskL : nullptr, distRL, kthL);
        else
            hipLaunchKernelGGL(mutual_right, dim3(cdiv(nR, 256)), dim3(256), 0, stream, idxLR, k2, idxRL, nR, k1, lpos2c,
                               partR, cntR, fused ? maskL : nullptr);
        BMX_LAUNCH_CHECK();
    }
}

void pair_scans(hipStream_t stream, ScanWorkspace& ws, const unsigned long long* maskL, const int32_t* cntL, int nsel, int k2,
                int32_t* offL, int32_t* totalP, const int32_t* cntR, int nR, int32_t* offR, int32_t* second_u,
                int32_t* totalU) {
    const int nbA = std::max(1, cdiv(nsel, scan::BLOCK)), nbB = std::max(1, cdiv(nR, scan::BLOCK));
    const scan::Chain chA = ws.chain(stream, 0, std::max(nbA, nbB));
    const scan::Chain chB = ws.chain(stream, 1, std::max(nbA, nbB));
    hipLaunchKernelGGL(pair_scans_kernel, dim3(nbA + nbB), dim3(256), 0, stream, chA, nbA, k2 <= 64 ? maskL : nullptr, cntL, nsel,
                       offL, totalP, chB, nbB, cntR, nR, offR, second_u, totalU);
    BMX_LAUNCH_CHECK();
}

void emit_pairs(hipStream_t stream, const int32_t* idxLR, int nL, int k2, const int32_t* idxRL, int k1,
                const int32_t* offL, const int32_t* lrows, const int32_t* rrows, int32_t* first, int32_t* second,
                const int32_t* lsel, const unsigned long long* maskL, const SortedRows* sorted) {
    if (nL <= 0) return;
    if (sorted && sorted_rows_apply(k1, k2)) {  // (sorted->hits: what mutual_counts found for these lists)
        hipLaunchKernelGGL(emit_pairs_sorted, dim3(cdiv(nL, 4)), dim3(256), 0, stream, idxLR, nL, k2,
                           (const unsigned long long*)sorted->hits.p, offL, lsel, lrows, rrows, first, second);
        BMX_LAUNCH_CHECK();
        return;
    }
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(cdiv(nL, 256)), dim3(256), 0, stream, idxLR, nL, k2, idxRL, k1, offL,
                       lsel, lrows, rrows, k2 <= 64 ? maskL : nullptr, first, second);
    BMX_LAUNCH_CHECK();
}

void select_listed_rows(hipStream_t stream, ScanWorkspace& ws, const int32_t* idx, int64_t n_entries, int n_rows,
                        int32_t* stamp, int gen, int32_t* off, int32_t* sel, int32_t* total_out, float* seed_zero,
                        unsigned long long* mask_zero) {
    if (n_entries > 0) {
        hipLaunchKernelGGL(mark_listed, dim3(cdiv(n_entries, 256)), dim3(256), 0, stream, idx, n_entries, stamp, gen);
        BMX_LAUNCH_CHECK();
    }
    const int nb = std::max(1, cdiv(n_rows, scan::BLOCK));
    hipLaunchKernelGGL(select_listed_kernel, dim3(nb), dim3(256), 0, stream, ws.chain(stream, 0, nb), (const int32_t*)stamp, gen,
                       n_rows, off, sel, total_out, reinterpret_cast<uint32_t*>(seed_zero), mask_zero);
    BMX_LAUNCH_CHECK();
}

void seed_thresholds(hipStream_t stream, const int32_t* idxRL, const double* distRL, int64_t n_entries,
                     const int32_t* lpos2c, int nsel, float* seed, const int32_t* nsel_dev, int rank, int world) {
    if (nsel <= 0) return;
    if (!nsel_dev) world = 1;
    if (n_entries > 0) {  // (seed[0, nsel) was zeroed by select_listed_rows)
        hipLaunchKernelGGL(seed_from_lists, dim3(cdiv(n_entries, 256)), dim3(256), 0, stream, idxRL, distRL, n_entries,
                           lpos2c, reinterpret_cast<uint32_t*>(seed), nsel_dev, rank, world);
        BMX_LAUNCH_CHECK();
    }
}

void compose_row_list(hipStream_t stream, const int32_t* sel, int n, const int32_t* rows, int32_t* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(compose_rows, dim3(cdiv(n, 256)), dim3(256), 0, stream, sel, n, rows, out);
    BMX_LAUNCH_CHECK();
}

}  // namespace bmx
