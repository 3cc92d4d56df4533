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

#ifdef __cplusplus
}
#endif
#endif /* BATCHELOR_MI355X_H */
