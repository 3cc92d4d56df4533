// Device-level building blocks of the merge loop (implemented in pairs.hip / correct.hip / legacy.hip).
// All pointers are device pointers, all matrices row-major [cells x d] FP64, all launches go to `stream`.
#pragma once
#include <functional>
#include "bmx_common.hpp"

namespace bmx {

// ---- scan / small utilities (pairs.hip) ------------------------------------------------------------
namespace scan {
struct Chain;
}
// Chains of the single-pass scans (scan_lookback.hpp): two, so that one launch can compact two lists.  Never cleared
// between launches: every launch brings a fresh epoch.
struct ScanWorkspace {
    DevBuf<unsigned long long> status;
    DevBuf<unsigned int> ticket;
    size_t chain_cap = 0;
    unsigned int epoch = 0;
    scan::Chain chain(hipStream_t stream, int which, int nblocks);
};
// out[i] = sum_{j<i} in[j] for i in [0, n]; out has n + 1 entries (out[n] = total).  in/out may not alias.  One launch.
void exclusive_scan_i32(hipStream_t stream, ScanWorkspace& ws, const int32_t* in, int32_t* out, int n);

// ---- mutual nearest neighbours (pairs.hip) ---------------------------------------------------------
// idxLR [nL][k2]: for each left cell the positions of its nearest right cells (rank order);
// idxRL [nR][k1]: for each right cell the positions of its nearest left cells.
// partR [nR][k1]: mutual left partners of each right cell, ascending; cntR[r] their number.
// maskL (nL words, k2 <= 64): bit j of word c = neighbour j of row c of idxLR is mutual (zeroed here unless the caller says
// it is clear); without it (k2 > 64) cntL[c] = number of mutual partners of row c.
// idxLR may cover a SUBSET of the left cells: row c belongs to left cell lsel[c] (ascending) and lpos2c[l] is the row
// of a selected left cell l (both nullptr: one row per left cell).  nL = number of rows of idxLR.
// SortedRows (nullable; used where sorted_rows_apply(k1, k2): 64 < k2 <= 8192, k1 <= 8192): mutual_counts sorts each row of
// both lists into it and answers every "does row x list y" by binary search instead of reading the row -- k1 * k2 words a
// cell otherwise; emit_pairs handed the same object reads the left rows' hits as bits instead of searching again.  Results are identical.
struct SortedRows {
    DevBuf<int32_t> lr, rl;
    DevBuf<unsigned long long> hits;  // [nL][ceil(k2 / 64)]: bit j of word w of row c = neighbour 64 w + j of row c is mutual
};
bool sorted_rows_apply(int k1, int k2);
void mutual_counts(hipStream_t stream, const int32_t* idxLR, int nL, int k2, const int32_t* idxRL, int nR, int k1,
                   int32_t* cntL, int32_t* partR, int32_t* cntR, const int32_t* lsel = nullptr,
                   const int32_t* lpos2c = nullptr, unsigned long long* maskL = nullptr, bool mask_is_clear = false,
                   const double* distRL = nullptr, const double* kthL = nullptr, SortedRows* sorted = nullptr);
// (distRL [nR][k1] + kthL [nL], both or neither: the right cells' exact distances to their listed left cells and each
// left row's largest distance, +inf where unknown -- the probe then rejects without reading the row where it can)
// ONE launch behind mutual_counts: offL [nsel + 1] = exclusive scan of the pairs per row of idxLR (popcount of maskL, or
// cntL where k2 > 64), *totalP = their number; second_u = ascending positions r with cntR[r] > 0, offR [nR + 1] their
// exclusive scan, *totalU their number (device words).
void pair_scans(hipStream_t stream, ScanWorkspace& ws, const unsigned long long* maskL, const int32_t* cntL, int nsel, int k2,
                int32_t* offL, int32_t* totalP, const int32_t* cntR, int nR, int32_t* offR, int32_t* second_u, int32_t* totalU);
// Pairs in the reference order (src/find_mutual_nns.cpp:23-36): left ascending, then the left cell's neighbour rank.
// offL = exclusive scan of the pairs per row.  Ids written are lrows[l] + 1 / rrows[r] + 1 (1-based rows in the node;
// identity if null).  maskL (nullable): emit_pairs then skips the lookups.
void emit_pairs(hipStream_t stream, const int32_t* idxLR, int nL, int k2, const int32_t* idxRL, int k1,
                const int32_t* offL, const int32_t* lrows, const int32_t* rrows, int32_t* first, int32_t* second,
                const int32_t* lsel = nullptr, const unsigned long long* maskL = nullptr,
                const SortedRows* sorted = nullptr);
// Rows (of n_rows) that occur in idx[0, n_entries): they are stamped with `gen` in stamp [n_rows] (a buffer that is zero
// when first used and never cleared: every call brings a larger gen), off = exclusive scan of "row is listed" (n_rows + 1
// entries: off[r] = position of a listed row in sel, off[n_rows] = their number, also written to the device word
// *total_out), sel = the listed rows, ascending.  seed_zero / mask_zero (nullable): word `position` of each is zeroed for
// every selected row -- what seed_thresholds and mutual_counts accumulate into.  Two launches.
void select_listed_rows(hipStream_t stream, ScanWorkspace& ws, const int32_t* idx, int64_t n_entries, int n_rows,
                        int32_t* stamp, int gen, int32_t* off, int32_t* sel, int32_t* total_out, float* seed_zero = nullptr,
                        unsigned long long* mask_zero = nullptr);
// seed[c] (c < nsel, zero on entry) = the largest squared distance (rounded up to f32) at which a right cell lists the c-th
// selected left cell: idxRL / distRL [n_entries] = the right cells' neighbour lists with their Euclidean distances, lpos2c
// maps a listed left cell to its row among the selected ones.  No mutual partner of that left cell lies farther.
// (nsel_dev / rank / world: with several ranks only the seeds of this rank's slice of the selected cells are taken)
void seed_thresholds(hipStream_t stream, const int32_t* idxRL, const double* distRL, int64_t n_entries,
                     const int32_t* lpos2c, int nsel, float* seed, const int32_t* nsel_dev = nullptr, int rank = 0,
                     int world = 1);
// out[i] = rows[sel[i]]
void compose_row_list(hipStream_t stream, const int32_t* sel, int n, const int32_t* rows, int32_t* out);

// ---- correction primitives (correct.hip) -----------------------------------------------------------
struct ReduceWorkspace {
    DevBuf<double> partial;
};
// mode 0: sum x, 1: sum x^2, 2: sum (x - centre[c])^2 over rows [r0, r1) of X (optionally through a row list).
// out[c] = scale * sum.  Deterministic two-stage reduction.
void col_reduce(hipStream_t stream, ReduceWorkspace& ws, const double* X, const int32_t* rows, int r0, int r1, int d,
                int mode, const double* centre, double scale, double* out);
// out_sum[c] = scale * sum_r X[r][c], out_sq[c] = scale * sum_r X[r][c]^2 in one pass over X [n][d]
void col_reduce2(hipStream_t stream, ReduceWorkspace& ws, const double* X, int n, int d, double scale, double* out_sum,
                 double* out_sq);
// Fused row pass over the segments (row ranges) of one node, in place on X [*][d]:
//   * nvec > 0: centre along the batch vectors vec_pool[vec_ids[e]] in turn, x <- x - ((x - mu) . v^) v^  with mu [d]
//     the column mean over the node's restrict rows BEFORE the pass (it is invariant under every step), i.e.
//     .center_along_batch_vector / .orthogonalize_other (R/fastMNN.R:626-647) without their three passes per vector;
//   * stat_slots != nullptr: per segment i, the column means of the rows as written go to
//     means_pool[stat_slots[i]][d] and the sum over columns of the sample variance (.compute_perbatch_var,
//     R/fastMNN.R:651-658) to scal[stat_slots[i]] -- one pass, shifted sums.
void rows_apply_stats(hipStream_t stream, ReduceWorkspace& ws, double* X, int d, const int* starts, const int* ns,
                      int nseg, const double* mu, const double* vec_pool, const int* vec_ids, int nvec,
                      const int* stat_slots, double* means_pool, double* scal);
// The general form: the segments may live in different matrices (the left and the right node of a merge go through one
// launch) and bring their own column mean and variance shift.
struct RowSeg {
    double* X;            // the matrix the rows [start, start + n) belong to
    int start, n;
    const double* mu;     // nvec > 0: column mean over the restrict rows of the segment's node
    const double* pivot;  // stats: a vector near the segment's mean (its earlier mean); nullptr: its first row
    int slot;             // stats: where its means / total variance go
};
void rows_multi(hipStream_t stream, ReduceWorkspace& ws, int d, const RowSeg* segs, int nseg, const double* vec_pool,
                const int* vec_ids, int nvec, bool stats, double* means_pool, double* scal);
// mu0 / mu1 [d] = row-weighted means of the segment means of two nodes (segments [0, nseg0) and [nseg0, nseg0 + nseg1); at
// most 16 in all) -- one launch
void node_means_from_segments(hipStream_t stream, const double* means_pool, const int* ns, const int* slots, int nseg0, int nseg1,
                              int d, double* mu0, double* mu1);
// mu [d] = row-weighted mean of the segment means means_pool[slots[i]] (at most 16 segments)
void node_mean_from_segments(hipStream_t stream, const double* means_pool, const int* ns, const int* slots, int nseg,
                             int d, double* mu);
// .get_batch_magnitude (R/fastMNN.R:582-595) from overall = colMeans(averaged), msq = colMeans(averaged^2)
void batch_magnitude(hipStream_t stream, const double* overall, const double* msq, int d, double* out);
// out[0] = scale * sum_c in[c]   (single thread; d is tiny)
void sum_vector(hipStream_t stream, const double* in, int d, double scale, double* out);

// .average_correction (R/fastMNN.R:567-580): for the u-th MNN-involved right cell (position second_u[u]) the mean of
// L[lrows[l]] - R[rrows[r]] over its partners l (ascending).  averaged [U][d].  Returns true when the fused form ran: then
// (with_sums) overall [d] = colMeans(averaged), msq [d] = colMeans(averaged^2) and *magnitude = .get_batch_magnitude
// (R/fastMNN.R:481,582-595; magnitude nullable) have come out of the same pass, and srows (nullable) [U] holds each cell's
// row in its node (rrows[second_u[u]]); false: the caller does those itself.
// (shard: a rank of a multi-GPU run takes its share of the workgroups of the averaging whose vectors nobody reads and
// all-gathers the workgroups' column sums through `exchange` (buffer, bytes per rank): section 5 of DESIGN.md)
struct AvgShard {
    int rank, world;
    std::function<void(void*, int64_t)> exchange;
};
bool average_correction(hipStream_t stream, ReduceWorkspace& ws, const double* L, const int32_t* lrows, const double* R,
                        const int32_t* rrows, int d, const int32_t* second_u, int U, const int32_t* partR, const int32_t* cntR,
                        int k1, double* averaged, bool with_sums = false, double* overall = nullptr, double* msq = nullptr,
                        double* magnitude = nullptr, int32_t* srows = nullptr, const int32_t* dup_next = nullptr,
                        const AvgShard* shard = nullptr);
// (dup_next, nullable: right cells named by several positions of the restrict list -- position r's cell continues at
// dup_next[r], -1 ends the chain; second_u then holds first positions only and a cell's pairs are those of its whole chain)

// .compute_tricube_average + add (R/utils_tricube.R:1-27, R/fastMNN.R:606-607) in place on X [n][d].
// idx [n][k] positions into `averaged` rows, dist [n][k] ascending Euclidean distances.
void tricube_apply(hipStream_t stream, double* X, int n, int d, const double* averaged, const int32_t* idx,
                   const double* dist, int k, double ndist);

// correction [n][d] = .compute_tricube_average alone (no add): the per-cell correction vectors var.adj rescales
void tricube_vectors(hipStream_t stream, int n, int d, const double* averaged, const int32_t* idx, const double* dist,
                     int k, double ndist, double* correction);
// X[i] += pmax(scaling[i], 1) * correction[i]   (R/mnnCorrect.R:479-480, :345)
void add_scaled_rows(hipStream_t stream, double* X, int n, int d, const double* correction, const double* scaling);
// layout helpers
void transpose_cm_to_rm(hipStream_t stream, const double* cm, int n, int d, double* rm);  // [n x d] col-major -> row-major
void transpose_rm_to_cm(hipStream_t stream, const double* rm, int n, int d, double* cm, int ld_cm, int row_off);

// ---- upstream of the engine (prepca.hip): cosineNorm + PCA projection, x is genes x cells column-major ----------
void cosine_l2_device(hipStream_t stream, const double* x, int G, int n, double* l2);
void apply_cosine_norm_device(hipStream_t stream, const double* x, int G, int n, const double* l2, double* out);
// out [n x d] column-major = crossprod(x / pmax(1e-8, l2) - centers, u)   (u [G x d] column-major); cos_norm = 0 skips
// the division.  One pass over x.  l2_out (nullable) receives the column norms; cu_scratch: d doubles.
void cosnorm_project_device(hipStream_t stream, const double* x, int G, int n, const double* u, int d,
                            const double* centers, int cos_norm, double* out, double* l2_out, double* cu_scratch);

// ---- multiBatchPCA on the device (pca.hip) -----------------------------------------------------------------------
class Pca;
Pca* pca_create(int device, int G);
void pca_destroy(Pca* p);
void pca_add_batch(Pca* p, const double* x_host, int64_t n, double weight, int cos_norm);
void pca_begin_batch(Pca* p, int64_t n, double weight, int cos_norm);
void pca_add_block(Pca* p, const double* x_block_host, int64_t m);
// tol > 0: until the relative Ritz residual of the d wanted pairs is <= tol, at most max_applies applications of the
// operator (throws if not reached); tol <= 0: exactly max_applies plain subspace steps
void pca_fit(Pca* p, int d, double tol, int max_applies, double* centers, double* rotation, double* sdev, int* applies_used,
             double* resid);
void pca_project(Pca* p, int batch, double* out_host);

// ---- legacy natives (legacy.hip) -------------------------------------------------------------------
void smooth_gaussian_kernel_device(hipStream_t stream, const double* averaged, int g, int U, const int32_t* index,
                                   const double* mat, int gd, int n, double sigma2, double* out, double* ws_density);
// What a call of these sizes runs and needs -- a pure function of its arguments (and of the testing hook "asv_fast"):
// the caller reserves main_doubles + extra_doubles of scratch and hands the same plan to the launch.
struct AsvPlan {
    int exact = 1;   // 1: asv_exact_kernel (the reference's order of operations literally), 0: the tiled FP64-MFMA form
    int blocks = 1;  // workgroups
    int npad = 1;    // exact form: nr1 rounded up to a power of two
    int lcap = 0;    // tiled form: addends a chain of the literal re-run of a flagged cell may keep (0: no re-run)
    size_t main_doubles = 0, extra_doubles = 0;
};
AsvPlan adjust_shift_variance_plan(int g, int n2, int nr1, int nr2, int vect_row_major);
// counters of the tiled form on the current device: {cells re-run literally, flagged cells beyond lcap, all cells}
void asv_tally_read(unsigned long long out[3], bool reset);
void asv_ticks_read(unsigned long long out[8]);  // diagnostics: 100 MHz ticks in the stream / at the round barrier / per-cell phase / -; literal cells: selection + re-evaluation, chains, kept addends, tiles
void asv_modes_read(unsigned char* dst, size_t n);  // testing hook "asv_modes": the way each cell of the last call went
// out[c] for the cells c in [cell_begin, cell_end) only (cell_end < 0: n2) -- the unit a multi-GPU run shards by
void adjust_shift_variance_device(hipStream_t stream, const double* data1, int g, int n1, const double* data2, int n2,
                                  const double* vect, double sigma2, const int32_t* restrict1, int nr1,
                                  const int32_t* restrict2, int nr2, double* out, double* ws_pairs, const AsvPlan& plan,
                                  int vect_row_major = 0, int cell_begin = 0, int cell_end = -1);

}  // namespace bmx
