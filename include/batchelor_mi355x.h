/*
 * batchelor_mi355x.h -- C ABI of the MI355X-native fastMNN / reducedMNN hot path (libbatchelor_mi355x.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch / R types.  Each entry point names the
 * reference interface it replaces (paths relative to the batchelor source tree).  Matrices crossing the boundary
 * use R's memory layout: column-major double, int32 indices, 1-based cell / batch ids unless stated otherwise.
 * All pointers are HOST pointers unless the name says `_dev`.  Calls return BMX_OK or a negative code; the message
 * (identical to the R / Rcpp error text where the reference has one) is read with bmx_last_error() so that the
 * R shim can hand it to Rf_error() verbatim (src/RcppExports.cpp:11,21 BEGIN_RCPP/END_RCPP).
 */
#ifndef BATCHELOR_MI355X_H
#define BATCHELOR_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMX_OK 0
#define BMX_ERR_HIP (-1)        /* HIP runtime failure (no GPU, out of memory, launch error) */
#define BMX_ERR_DIM_GENES (-2)  /* "number of genes do not match up between matrices"  src/adjust_shift_variance.cpp:35 */
#define BMX_ERR_DIM_CELLS (-3)  /* "number of cells do not match up between matrices"  src/adjust_shift_variance.cpp:40 */
#define BMX_ERR_SUBSET (-4)     /* "subset indices out of range"                        src/utils.cpp:9 */
#define BMX_ERR_INDEX_LEN (-5)  /* "'index' must have length equal to number of rows in 'averaged'" src/smooth_gaussian_kernel.cpp:19 */
#define BMX_ERR_ARG (-6)        /* invalid argument (message says which) */
#define BMX_ERR_TREE (-7)       /* "invalid leaf nodes specified in 'merge.order'" R/MNN_tree.R:104 and friends */
#define BMX_ERR_NO_PAIRS (-8)   /* no MNN pairs in a merge (R raises an error there: R/fastMNN.R:588-589 on NaN) */
#define BMX_ERR_EXCHANGE (-9)   /* the multi-GPU exchange callback failed */

/* Message of the last failing call on this thread. */
const char* bmx_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU); never throws. */
int32_t bmx_device_count(void);
/* Frees arrays returned through `int32_t**` / `double**` out-parameters. */
void bmx_free(void* p);
/* The one-shot entry points (bmx_fast_mnn, the single primitives) run on the calling thread's CURRENT HIP device; a host
 * without HIP bindings of its own picks it here (hipSetDevice). */
int32_t bmx_set_device(int32_t device);
/* Engines park their device blocks in a process-wide pool when they go (at most 16 GB / 256 blocks: a host that calls
 * fastMNN() again and again then pays its hipMallocs once); an allocation that fails inside the library empties the pool
 * by itself, other allocators of the process (torch, RCCL, the host's own hipMalloc) call this when they need the memory.
 * The (at most 8 per device) streams parked by engines that are gone are destroyed too. */
void bmx_trim_caches(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Legacy .Call kernels (R_CallMethodDef table, src/RcppExports.cpp:48-58)
 * ---------------------------------------------------------------------------------------------------------------- */

/* Replaces _batchelor_find_mutual_nns (src/find_mutual_nns.cpp:8-41; R stub R/RcppExports.R:8-10).
 * left [nL x k2], right [nR x k1]: column-major, 1-based.  Pairs come out for left cell ascending and, within a
 * left cell, in the order of its row of `left`.  *out_left / *out_right are malloc'ed (bmx_free). */
int32_t bmx_find_mutual_nns(const int32_t* left, int32_t nL, int32_t k2, const int32_t* right, int32_t nR, int32_t k1,
                            int32_t** out_left, int32_t** out_right, int64_t* npairs);

/* Replaces _batchelor_smooth_gaussian_kernel (src/smooth_gaussian_kernel.cpp:11-118; caller R/mnnCorrect.R:458).
 * averaged [g x U], index [index_len] 0-based columns of mat, mat [gd x n], out [g x n] caller-allocated. */
int32_t bmx_smooth_gaussian_kernel(const double* averaged, int32_t g, int32_t U, const int32_t* index,
                                   int32_t index_len, const double* mat, int32_t gd, int32_t n, double sigma2,
                                   double* out);

/* Replaces _batchelor_adjust_shift_variance (src/adjust_shift_variance.cpp:30-164; caller R/mnnCorrect.R:477).
 * data1 [g1 x n1], data2 [g2 x n2], vect [vrow x vcol] (= n2 x g), restrict1/2 0-based, out [n2]. */
int32_t bmx_adjust_shift_variance(const double* data1, int32_t g1, int32_t n1, const double* data2, int32_t g2,
                                  int32_t n2, const double* vect, int32_t vrow, int32_t vcol, double sigma2,
                                  const int32_t* restrict1, int32_t nr1, const int32_t* restrict2, int32_t nr2,
                                  double* out);

/* Which form of adjust_shift_variance a call of these sizes takes (a pure function of the sizes): 1 = the reference's order
 * of operations literally (bit-equal to the CPU restatement, every cell; up to 4e7 (cell, restricted cell) pairs), 2 =
 * 16-cell tiles on the FP64 matrix cores with a histogram quantile (beyond that: a cell whose quantile walk is decided on
 * the last bits may land on the neighbouring quantile).  The engine's var_adj merges go through the same switch. */
int32_t bmx_adjust_shift_variance_form(int32_t n2, int32_t nr1, int32_t nr2);

/* ------------------------------------------------------------------------------------------------------------------
 * Third-party contract on the path: BiocNeighbors::queryKNN / findMutualNN (re-export R/findMutualNN.R:1-3; call
 * sites R/MNN_tree.R:129, R/fastMNN.R:605).  Exact Euclidean search, ascending distance, ties by lowest index.
 * ---------------------------------------------------------------------------------------------------------------- */

/* queryKNN(X, query, k): X [nx x d], query [nq x d] column-major; index [nq x k] 1-based, distance [nq x k]
 * (either output may be NULL).  k is clamped to nx by the caller (safe.k, R/fastMNN.R:604). */
int32_t bmx_query_knn(const double* X, int32_t nx, const double* query, int32_t nq, int32_t d, int32_t k,
                      int32_t* index, double* distance);

/* findMutualNN(data1, data2, k1, k2) -> first / second, 1-based, malloc'ed (bmx_free). */
int32_t bmx_find_mutual_nn(const double* data1, int32_t n1, const double* data2, int32_t n2, int32_t d, int32_t k1,
                           int32_t k2, int32_t** first, int32_t** second, int64_t* npairs);

/* Diagnostics of the last kNN call on this thread: queries that needed the exact FP64 re-scan. */
int64_t bmx_last_knn_exact_fallbacks(void);
/* Testing hook: non-zero routes every kNN query through the exact FP64 re-scan. */
void bmx_set_force_exact_knn(int32_t on);
/* Testing hooks, process-wide, set by an explicit call only: NOTHING in the environment of the host process changes what
 * the library computes, which tier of the search or which form of adjust_shift_variance runs (the library reads two
 * environment variables: BMX_DEBUG=1 prints the shape decisions of every search to stderr, BMX_HOST_THREADS=n sizes the
 * pool of host threads behind the pinned staging ring).  Knobs: "knn_tier" (1 / 2: that candidate tier only, 3: the exact
 * FP64 scan only), "sample" (rows of a candidate pass's threshold sample, -1 = automatic), "split_c" / "force_c"
 * (reference ranges of the tail / of every query block), "no_margin" (fp16 tier: lists cut at their KS-th best only),
 * "asv_fast" (the tiled form of adjust_shift_variance at any size), "exchange_always" (a single rank goes through its
 * exchange transport too), "refine_wave" (the exact re-rank spends a whole wave on every query), "asv_cap" (tiled
 * adjust_shift_variance: addends a chain of the literal re-run of an ill-conditioned cell may keep; -1 = default, 0 = no
 * re-run), "asv_modes" (n: the tiled form records which way each of the first n cells of a call went, see
 * bmx_dev_get_bytes), "asv_sync" (0: the tiled form's workgroups do not wait for each other at the start of a round of
 * tiles -- the default; 1 = they do, measured slower), "sample_split" (ranges the threshold sample of a search with few
 * query blocks is split into; -1 = automatic, the default; 0 = never), "lk_seed" (searches with k beyond the tiers' lists: 1 = the
 * reference's partitions after the first are searched within the first's 36th distance, the default; 0 = all plainly), "tau_replay" (developer experiment: 1 = every search of
 * a run records its queries' final thresholds, 2 = the same sequence of searches starts its full passes from them), "reset"
 * (all back to their defaults).
 * Unknown name: BMX_ERR_ARG. */
int32_t bmx_dev_set(const char* name, int32_t value);
/* Counters for tests and bench.py, current device: "asv_tiled_cells" (cells the tiled form of adjust_shift_variance has
 * handled), "asv_literal_cells" (of those, re-run in the reference's order of operations: bit-equal to the exact form),
 * "asv_fallback_cells" (ill-conditioned cells with too many significant pairs for the re-run: histogram quantile, may pick a
 * neighbouring quantile), "asv_tally_reset" (zeroes them and the ticks); 100 MHz device ticks of the tiled form added up over its
 * workgroups: "asv_ticks_stream", "asv_ticks_wait", "asv_ticks_cells" (the stream, the testing hook's round barrier, the per-cell
 * phase), "asv_ticks_literal" / "asv_ticks_chains" (the re-run cells' selection + re-evaluation + sort, their chains and walks),
 * "asv_literal_addends" (addends the re-run cells kept), "asv_chain_tiles" (tiles with a re-run cell).  Waits for the device. */
int32_t bmx_dev_get(const char* name, int64_t* value);
/* "asv_modes" (after bmx_dev_set "asv_modes" = n): which way each of the first n cells of the LAST tiled adjust_shift_variance
 * call went -- 0 the histogram quantile (a well-conditioned cell), 1 re-run in the reference's order of operations, 2
 * ill-conditioned with more significant pairs than the re-run holds; 255 beyond what was recorded. */
int32_t bmx_dev_get_bytes(const char* name, void* dst, int64_t n);
/* HIP-event milliseconds of the device kernels of this thread's last bmx_smooth_gaussian_kernel / bmx_adjust_shift_variance
 * call (the host transfers of the call excluded): what bench.py's roofline of the two legacy natives is taken over. */
double bmx_last_native_kernel_ms(void);
/* Testing hook, needs no GPU: dst[0, bytes) = src[0, bytes) by the pool of host threads that moves the boundary's
 * matrices between the caller's memory and the pinned staging buffers (csrc/host_xfer.hpp) -- lets a CPU test hammer that
 * pool from several threads at once. */
int32_t bmx_dev_host_copy(void* dst, const void* src, int64_t bytes);

/* ------------------------------------------------------------------------------------------------------------------
 * The merge engine: replaces .fast_mnn / .fast_mnn_core (R/fastMNN.R:398-562) and everything they call
 * (R/MNN_tree.R, R/utils_tricube.R, R/utils_reorder.R) with one device-resident run.  The natural cut is the call
 * `.fast_mnn(batches, k, prop.k, restrict, ndist, merge.order, auto.merge, min.batch.skip)` made from
 * R/fastMNN.R:356,380 and R/reducedMNN.R:83,89.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct bmx_engine bmx_engine_t;

typedef struct {
    int32_t struct_size;   /* sizeof(bmx_params_t) as the CALLER's header has it: the library reads that many bytes and
                              gives the fields a later header appends their defaults (var_adj = 0, sigma = 0.1), so a
                              caller built against any header that HAS this field keeps working (the layout before it
                              -- round 2's, without struct_size -- is not readable and is refused for k < 36; fields
                              are only ever appended from here on); smaller than the fields up to auto_merge, or
                              larger than the library's own struct with non-zero bytes beyond: BMX_ERR_ARG */
    int32_t k;             /* k = 20 */
    double prop_k;         /* NaN = NULL (R/MNN_tree.R:140-146) */
    double ndist;          /* ndist = 3 */
    double min_batch_skip; /* NaN = NA_real_: batch.size not computed, nothing skipped (R/fastMNN.R:484) */
    int32_t auto_merge;    /* non-zero: R/MNN_tree.R:154-226 instead of the predefined tree */
    int32_t var_adj;       /* non-zero: rescale every cell's correction vector as mnnCorrect(var.adj=TRUE) does --
                              pmax(adjust_shift_variance(left, right, correction, sigma), 1) * correction
                              (R/mnnCorrect.R:331-342,462-481).  fastMNN() has no such switch: 0 is its behaviour */
    double sigma;          /* bandwidth handed to adjust_shift_variance as `sigma2` (R/mnnCorrect.R:477), 0.1 */
} bmx_params_t;

/* Multi-GPU exchange: called by the engine when `buf` (a DEVICE buffer of world * bytes_per_rank bytes whose slice
 * [rank * bytes_per_rank, +bytes_per_rank) this rank has filled) must become identical on all ranks (an in-place
 * all-gather; RCCL over xGMI in production).  The engine's stream is idle when it is called.  Return 0 on success. */
typedef int32_t (*bmx_allgather_fn)(void* ctx, void* buf, int64_t bytes_per_rank);

/* Creates an engine on HIP device `device` (its workspaces persist across runs). */
int32_t bmx_engine_create(int32_t device, bmx_engine_t** out);
void bmx_engine_destroy(bmx_engine_t* e);
/* Every kNN search is split by query rows over `world` ranks and completed with `fn`; all ranks hold all batches. */
int32_t bmx_engine_set_shard(bmx_engine_t* e, int32_t rank, int32_t world, bmx_allgather_fn fn, void* ctx);
/* Production exchange: RCCL called from inside the engine, in place on the engine's stream (no host round trip, no
 * staging copy).  bmx_rccl_load resolves the RCCL entry points from `librccl_path` (the RCCL the process already
 * uses -- one RCCL per process; NULL searches the global scope); rank 0 makes the 128-byte id with
 * bmx_rccl_unique_id and the host side hands it to every rank (MPI, torch.distributed, a file: not this library's
 * business); bmx_engine_init_rccl is collective over the ranks and replaces any callback set with set_shard. */
int32_t bmx_rccl_load(const char* librccl_path);
int32_t bmx_rccl_unique_id(void* id_out, int32_t bytes);
int32_t bmx_engine_init_rccl(bmx_engine_t* e, int32_t rank, int32_t world, const void* unique_id, int32_t bytes);
/* All-gathers issued and bytes received by this rank since the last bmx_engine_run started. */
int32_t bmx_engine_exchange_stats(bmx_engine_t* e, int64_t* calls, int64_t* bytes);
/* Copies the batches to HBM.  data[b]: nrows[b] x d column-major; restrict_idx[b]: 1-based, any order, a cell may be named
 * more than once (any R subsetting vector, R/checkInputs.R:96-120: the search then sees it as that many points), 
 * n_restrict[b] entries, or NULL / n_restrict[b] < 0 for "all cells". */
int32_t bmx_engine_upload(bmx_engine_t* e, int32_t nbatches, int32_t d, const double* const* data,
                          const int32_t* nrows, const int32_t* const* restrict_idx, const int32_t* n_restrict);
/* Runs all merges on the resident inputs (asynchronous launches; returns after the final stream synchronisation).
 * tree: the binary merge tree in post-order, leaf = 1-based batch id, 0 = "merge the two nodes on top of the stack,
 * deeper one is the left/reference child" -- e.g. list(list(1,2),3) is {1,2,0,3,0}.  Ignored when auto_merge. */
int32_t bmx_engine_run(bmx_engine_t* e, const bmx_params_t* params, const int32_t* tree, int32_t tree_len);
/* Results of the last run.  corrected: N x d column-major, rows in input batch order (R/fastMNN.R:541-547);
 * batch: N batch ids (1-based); merge_left / merge_right: (B-1) x B row-major, batch ids of each merge's left and
 * right sets padded with 0; batch_size, skipped: B-1; lost_var: (B-1) x B column-major.  Any pointer may be NULL. */
int32_t bmx_engine_download(bmx_engine_t* e, double* corrected, int32_t* batch, int32_t* merge_left,
                            int32_t* merge_right, double* batch_size, int32_t* skipped, double* lost_var);
/* MNN pairs of merge `merge` (0-based), 1-based OUTPUT-row indices (R/fastMNN.R:533-547); malloc'ed (bmx_free). */
int32_t bmx_engine_pairs(bmx_engine_t* e, int32_t merge, int32_t** left, int32_t** right, int64_t* npairs);
/* The same into arrays the caller owns (an R shim allocates its INTSXPs first and saves a copy): *npairs receives the
 * number of pairs; with left == right == NULL nothing else happens (size query), else capacity must be >= *npairs. */
int32_t bmx_engine_pairs_into(bmx_engine_t* e, int32_t merge, int32_t* left, int32_t* right, int64_t capacity,
                              int64_t* npairs);
/* Every merge's pairs in one call (what the shim does after a run: merge.info's `pairs` list, R/fastMNN.R:550-561):
 * left[m] / right[m] receive merge m's pairs, capacity[m] >= its pair count (bmx_engine_pairs_into's size query);
 * nmerges must be the number of merges of the last run.  One pass of the host threads over all lists instead of one
 * per list. */
int32_t bmx_engine_pairs_all_into(bmx_engine_t* e, int32_t nmerges, int32_t* const* left, int32_t* const* right,
                                  const int64_t* capacity);
/* Sizes of merge `merge`: out[0..5] = {cells searched on the left, on the right, MNN-involved right cells U,
 * pairs P, all left cells, all right cells} -- the inputs of the algorithmic flop / byte counts. */
int32_t bmx_engine_merge_stats(bmx_engine_t* e, int32_t merge, int64_t* out6);
/* Hang safety.  The candidate kernels synchronise their waves through LDS words in unbounded poll loops, so the HOST
 * never waits without a deadline: every wait of a run polls the engine's stream against base_ms plus a term scaled from
 * the work queued (about 1e4 times its expected duration).  When a deadline passes the call returns BMX_ERR_HIP with a
 * message starting "watchdog:", and the engine is dead: every later call on it fails at once and bmx_engine_destroy
 * abandons its stream and device memory instead of waiting for them -- only a fresh process gets the GPU back.
 * Default base 60 000 ms; base_ms <= 0 switches the watchdog off (plain hipStreamSynchronize). */
int32_t bmx_engine_set_watchdog(bmx_engine_t* e, double base_ms);
/* Testing hook of the watchdog: queues a kernel that keeps the engine's stream busy for `ms` milliseconds and then ends
 * by itself (the GPU stays healthy). */
int32_t bmx_engine_debug_stall(bmx_engine_t* e, int32_t ms);
/* With profiling on, every launch of a candidate-pass kernel (knn_topk_f16 / knn_topk_bf16, bmx_engine_knn_kernel names
 * the last one) is bracketed by HIP events on the engine's stream; after a run: total milliseconds, number of launches, and queries that needed the exact re-scan. */
int32_t bmx_engine_set_profiling(bmx_engine_t* e, int32_t on);
int32_t bmx_engine_profile(bmx_engine_t* e, double* topk_ms, int64_t* topk_launches, int64_t* exact_fallbacks);
/* Diagnostics for full-size parity checks: the next runs keep a device copy of the two matrices that merge `merge`
 * (0-based; -1 = off) hands to findMutualNN -- the left and right node after orthogonalisation
 * (R/fastMNN.R:473-477).  bmx_engine_snapshot copies them out ROW-major ([n x d], cells in node order); pass NULL
 * matrices to read the sizes first. */
int32_t bmx_engine_set_snapshot(bmx_engine_t* e, int32_t merge);
int32_t bmx_engine_snapshot(bmx_engine_t* e, double* left_rm, double* right_rm, int64_t* n_left, int64_t* n_right);
/* With params.var_adj, the snapshot merge also keeps what its adjust_shift_variance call (R/mnnCorrect.R:462-481) was handed
 * and what it returned: the centred left and right nodes and the right cells' correction vectors (ROW-major [n x d]), the
 * scalings [n_right] before pmax(., 1), the two restrict vectors (0-based rows of the nodes).  sizes4 = {n_left, n_right,
 * length(restrict1), length(restrict2)}; NULL pointers are skipped.  Lets a test hold one merge of a large tree against
 * src/adjust_shift_variance.cpp on the same inputs, a sample of cells at a time. */
int32_t bmx_engine_snapshot_var_adj(bmx_engine_t* e, double* left_rm, double* right_rm, double* corr_rm, double* scaling,
                                    int32_t* restrict1, int32_t* restrict2, int64_t* sizes4);
/* Testing hooks of var_adj runs made with bmx_dev_set("asv_modes", n > 0) (they wait for the device after every merge):
 * out3 = cells merge `merge`'s tiled adjust_shift_variance call re-ran in the reference's order of operations / flagged as
 * ill-conditioned but beyond the re-run / handled in all (-1: not recorded, e.g. the exact form ran); and the way every right
 * cell of the SNAPSHOT merge went (dst[0, n): 0 histogram quantile, 1 re-run, 2 flagged beyond it; 255 not recorded). */
int32_t bmx_engine_var_adj_tally(bmx_engine_t* e, int32_t merge, int64_t* out3);
int32_t bmx_engine_snapshot_var_adj_modes(bmx_engine_t* e, uint8_t* dst, int64_t n);
/* Per kernel class (profiling on, since the last run started): out[0], out[1] = milliseconds and launches of the fp16
 * full pass, out[2], out[3] of the split-bf16 full pass, out[4], out[5] of the sample passes, out[6] = milliseconds of
 * the merges' streaming sections (everything that is not a kNN search), out[7] = queries that took the exact FP64
 * path, out[8] = queries the first tier handed to the second, out[9] = runs of this engine that were repeated with
 * host-checked searches (a run first leaves the counts of its uncertified queries on the device; when a search cannot be
 * completed that way -- hundreds of uncertified queries, lists overflowing with ties -- it starts over; the ranks of a
 * sharded run agree on that through a flag that travels with every search's lists, and start over together). */
int32_t bmx_engine_profile_detail(bmx_engine_t* e, double* out10);
/* The last profiled run's adjust_shift_variance calls (params.var_adj): out[0] = milliseconds (HIP events on the engine's
 * stream around each call), out[1] = calls, out[2] = (cell, restricted cell) pairs this rank evaluated. */
int32_t bmx_engine_profile_var_adj(bmx_engine_t* e, double* out3);
/* Measurement hook: ONE rank's share of an N-rank run on one GPU.  mode 1: the next run (a single rank) records what every
 * exchange of the run would have gathered; mode 2: the following runs are rank `rank` of `world` -- each search covers that
 * rank's slice of the query rows (bmx_shard_range), every replicated kernel runs in full, and an exchange fills the other
 * ranks' slices from the recording by a device copy (the collective itself is not timed: bmx_engine_exchange_stats gives its
 * calls and bytes); results are those of the recorded run; mode 0: back to normal.  Predefined merge trees only (auto-merge
 * deals whole searches over the ranks); the engine must have no transport (bmx_engine_init_rccl / _set_shard). */
int32_t bmx_engine_emulate(bmx_engine_t* e, int32_t mode, int32_t rank, int32_t world);
/* Name of the full-pass candidate kernel the engine launched last, as rocprofv3 prints it (template arguments
 * included): lets a benchmark check that a stored counter measurement belongs to the kernel it has just timed. */
int32_t bmx_engine_knn_kernel(bmx_engine_t* e, char* buf, int32_t n);
/* Candidate-pass kernel used by the engine's last MFMA-path search: 2 = knn_topk_bf16 (split-bf16 MFMA, LDS ring),
 * 3 = knn_topk_f16 (single fp16 product, LDS ring), -1 = none yet (bmx_dev_set "knn_tier" restricts the search to one
 * tier for A/B runs). */
int32_t bmx_engine_knn_variant(bmx_engine_t* e);

/* One-shot convenience (what the R shim calls): create + upload + run + download + pairs stay queryable on *out_engine
 * until bmx_engine_destroy. */
int32_t bmx_fast_mnn(int32_t nbatches, int32_t d, const double* const* data, const int32_t* nrows,
                     const int32_t* const* restrict_idx, const int32_t* n_restrict, const bmx_params_t* params,
                     const int32_t* tree, int32_t tree_len, double* corrected, int32_t* batch, int32_t* merge_left,
                     int32_t* merge_right, double* batch_size, int32_t* skipped, double* lost_var,
                     bmx_engine_t** out_engine);

/* Row range [begin, end) of `n` query rows owned by `rank` of `world` (pure host arithmetic, no GPU). */
void bmx_shard_range(int64_t n, int32_t rank, int32_t world, int64_t* begin, int64_t* end);
/* Bytes each rank contributes to the in-place all-gather of a list with bytes_per_row bytes per query row (the slices
 * are padded to equal length: the gathered buffer holds world * this many bytes).  The engine sizes its exchanges with
 * the same function. */
int64_t bmx_shard_gather_bytes(int64_t n, int32_t world, int64_t bytes_per_row);

/* ------------------------------------------------------------------------------------------------------------------
 * Single primitives of the merge step, host in / host out, for parity tests that read like the reference's own
 * (tests/testthat/test-fast-mnn.R:7-92).  Same kernels as the engine.
 * ---------------------------------------------------------------------------------------------------------------- */
/* .center_along_batch_vector (R/fastMNN.R:626-640): mat [n x d] column-major in/out; restrict 1-based or NULL. */
int32_t bmx_center_along_batch_vector(double* mat, int32_t n, int32_t d, const double* batch_vec,
                                      const int32_t* restrict_idx, int32_t n_restrict);
/* .tricube_weighted_correction (R/fastMNN.R:599-608): curdata [n x d] in/out, correction [U x d], in_mnn [U] 1-based. */
int32_t bmx_tricube_weighted_correction(double* curdata, int32_t n, int32_t d, const double* correction,
                                        const int32_t* in_mnn, int32_t U, int32_t k, double ndist);
/* .average_correction (R/fastMNN.R:567-580) fed by findMutualNN on the same data: averaged [U x d] and second [U]
 * are malloc'ed; also returns the pairs. */
int32_t bmx_mnn_average_correction(const double* refdata, int32_t n1, const double* curdata, int32_t n2, int32_t d,
                                   int32_t k1, int32_t k2, int32_t** first, int32_t** second, int64_t* npairs,
                                   double** averaged, int32_t** second_u, int32_t* U);
/* .compute_perbatch_var (R/fastMNN.R:651-658) for one batch: sum over dims of the sample variance of data [n x d]. */
int32_t bmx_total_variance(const double* data, int32_t n, int32_t d, double* out);

/* ------------------------------------------------------------------------------------------------------------------
 * Directly upstream of the engine in fastMNN() (R/fastMNN.R:348-354).  x is genes x cells, column-major.
 * ---------------------------------------------------------------------------------------------------------------- */
/* cosineNorm(x, mode=) (R/cosineNorm.R:53-82): l2 [n] and / or the normalised matrix [G x n]; either may be NULL. */
int32_t bmx_cosine_norm(const double* x, int32_t G, int32_t n, double* l2, double* normalized);
/* One pass over x: cosine normalisation (if cos_norm != 0) fused with the PCA projection of R/multiBatchPCA.R:236-239,
 * out [n x d] = crossprod(cosineNorm(x) - centers, rotation); rotation [G x d] column-major, centers [G]. */
int32_t bmx_cosnorm_project(const double* x, int32_t G, int32_t n, const double* rotation, int32_t d,
                            const double* centers, int32_t cos_norm, double* out);

/* ------------------------------------------------------------------------------------------------------------------
 * multiBatchPCA on the device (R/multiBatchPCA.R:211-322; called from fastMNN() at R/fastMNN.R:353-354).  The batches
 * (genes x cells, column-major, as the reference has them) are uploaded once and stay in HBM; neither the scaled
 * genes x N matrix nor the genes x genes Gram matrix is formed: the top subspace is found by blocked subspace
 * iteration on the FP64 matrix cores with the centring and the cosine normalisation folded in.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct bmx_pca bmx_pca_t;
int32_t bmx_pca_create(int32_t device, int32_t n_genes, bmx_pca_t** out);
void bmx_pca_destroy(bmx_pca_t* p);
/* x: n_genes x n column-major (host).  weight: the batch's weight w_b (1 = R's default: every batch counts the same
 * whatever its size, R/multiBatchPCA.R:299-334).  cos_norm != 0: cosineNorm(x) (R/cosineNorm.R:63-82) on the fly. */
int32_t bmx_pca_add_batch(bmx_pca_t* p, const double* x, int64_t n, double weight, int32_t cos_norm);
/* The same batch handed over in column blocks, so that a 32 GB batch (200 000 cells x 20 000 genes, BASELINE.json
 * configs[3]) never has to exist in host memory at once: announce the batch, then its cells in order.  x_block is
 * n_genes x n_block column-major host memory, pageable or pinned; it has been read completely when the call returns
 * (it goes through a pinned double-buffered staging ring: host copy and DMA overlap). */
int32_t bmx_pca_begin_batch(bmx_pca_t* p, int64_t n, double weight, int32_t cos_norm);
int32_t bmx_pca_add_block(bmx_pca_t* p, const double* x_block, int64_t n_block);
/* multiBatchPCA's SVD (R/multiBatchPCA.R:386-393; the reference's irlba stops at tol = 1e-5) by Chebyshev-filtered
 * subspace iteration on a block of 64 vectors (d <= 56) or 128 (d <= 120): iterates until the Ritz residuals
 * max_j |M x_j - theta_j x_j| / theta_1 of the d wanted pairs are <= tol; at most max_iters applications of the operator
 * (one pass over every batch each), then BMX_ERR_ARG with the residual reached in the message.  iters_used / residual
 * (nullable) are written in both cases.  centers [n_genes], rotation [n_genes x d] column-major (columns defined up to
 * sign, as any SVD's), sdev [d] singular values of the scaled matrix; any may be NULL.  Needs n_genes and the total
 * number of cells above the block width. */
int32_t bmx_pca_fit_tol(bmx_pca_t* p, int32_t d, double tol, int32_t max_iters, double* centers, double* rotation,
                        double* sdev, int32_t* iters_used, double* residual);
/* Fixed-count form: exactly `iters` plain subspace iterations, no convergence test (kept for callers of round 2). */
int32_t bmx_pca_fit(bmx_pca_t* p, int32_t d, int32_t iters, double* centers, double* rotation, double* sdev);
/* crossprod(cosineNorm(x_b) - centers, rotation) (R/multiBatchPCA.R:236-239): out [n_b x d] column-major. */
int32_t bmx_pca_project(bmx_pca_t* p, int32_t batch, double* out);

#ifdef __cplusplus
}
#endif
#endif /* BATCHELOR_MI355X_H */
