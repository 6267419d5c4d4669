/*
 * bpmf_io.h -- C ABI of the matrix file IO that the `bpmf` executable of this repo uses (also
 * exported from libbpmf_hip.so so that tests can check it against independent readers).
 *
 * Keeps the reference's file surface (c++/io.cpp:15-77,137-193): the format is chosen by the
 * extension, optionally followed by ".gz" (zlib):
 *   sparse: .mtx/.mm (MatrixMarket coordinate real|integer|pattern general), .sdm (binary:
 *           u64 nrow, ncol, nnz; u32 rows[nnz], cols[nnz] 1-based; f64 vals[nnz]), .sbm (same
 *           without values, read as 1.0)                          -- c++/io.cpp:256-314,414-522
 *   dense:  .ddm (u64 nrow, ncol; f64 column-major), .mtx/.mm (MatrixMarket array real general,
 *           column-major), .csv (nrow\n ncol\n rows of comma separated values) -- c++/io.cpp:195-254,318-409
 * Sparse matrices are returned as CSC with ascending rows per column and duplicate entries summed
 * (Eigen setFromTriplets, c++/io.cpp:282,521); explicit zeros are kept.
 * All functions return 0 or a negative code; bpmf_io_last_error() has the message
 * ("File '<name>' not found", c++/io.cpp:117).
 */
#ifndef BPMF_IO_H
#define BPMF_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#define BPMF_IO_API __attribute__((visibility("default")))
#else
#define BPMF_IO_API
#endif

BPMF_IO_API const char *bpmf_io_last_error(void);
/* arrays are malloc'ed by the library: release each with bpmf_io_free */
BPMF_IO_API int bpmf_io_read_sparse(const char *path, int64_t *nrows, int64_t *ncols, int64_t *nnz,
                                    int64_t **colptr, int32_t **rowidx, double **vals);
BPMF_IO_API int bpmf_io_write_sparse(const char *path, int64_t nrows, int64_t ncols, const int64_t *colptr,
                                     const int32_t *rowidx, const double *vals);
/* column-major */
BPMF_IO_API int bpmf_io_read_dense(const char *path, int64_t *nrows, int64_t *ncols, double **data);
BPMF_IO_API int bpmf_io_write_dense(const char *path, int64_t nrows, int64_t ncols, const double *data);
BPMF_IO_API void bpmf_io_free(void *p);

/* Assignment of the columns of a side to `nparts` ranks (host; bpmf_amd/csrc/assign.cpp).
 * _greedy: Sys::assign of the reference (c++/assign.cpp:52-201, default weights): least-loaded rank on work = 10 + nnz,
 *   three sweeps, then a renumbering that makes every rank's columns contiguous -- order[new] = old column, dom[p] .. dom[p+1]
 *   = the new range of rank p.
 * _contiguous: cuts of the ORIGINAL order at equal c0 + nnz (no permutation: samples independent of the rank count). */
BPMF_IO_API int bpmf_assign_greedy(int64_t n, const int64_t *colptr, int nparts, int64_t *order, int64_t *dom);
BPMF_IO_API int bpmf_assign_contiguous(int64_t n, const int64_t *colptr, int nparts, double c0, int64_t *dom);

#ifdef __cplusplus
}
#endif
#endif
