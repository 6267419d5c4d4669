// TEST INFRASTRUCTURE -- an Eigen-free stand-in for the part of the reference's `struct Sys` that a back-end header sees
// (/root/reference c++/bpmf.h:78-104,112-239), so that the `hip_sys.h` stub of INTEGRATION.md section 2 can be COMPILED, LINKED
// against libbpmf_hip.so and RUN (tests/test_integration_stub.py).  The reference's own header needs Eigen3, which this image
// lacks; nothing here is the reference's code: the members the stub touches are declared with the reference's names and the
// accessor names of the Eigen types they have there (data(), outerIndexPtr(), innerIndexPtr(), valuePtr(), rows(), transpose()),
// and the three non-virtual pieces a back-end relies on (the two constructors, init(), print()) are restated from
// c++/sample.cpp:101-137,179-226.  Not part of the product, not a build of the reference.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "bpmf_io.h"

#ifndef BPMF_NUMLATENT
#define BPMF_NUMLATENT 32
#endif
const int num_latent = BPMF_NUMLATENT;                                   // c++/bpmf.h:53
#define THROWERROR(msg) throw std::runtime_error(msg)                    // c++/error.h:18-30
#define BPMF_COUNTER(name) do {} while (0)                                // c++/counters.h:60-66

// column-major dense matrix with the two Eigen members the stub uses
struct DenseD {
    int64_t nrows = 0, ncols = 0;
    std::vector<double> v;
    DenseD() {}
    DenseD(int64_t r, int64_t c) : nrows(r), ncols(c), v((size_t)(r * c), 0.0) {}
    double *data() { return v.data(); }
    const double *data() const { return v.data(); }
    int64_t size() const { return nrows * ncols; }
    DenseD transpose() const
    {
        DenseD t(ncols, nrows);
        for (int64_t j = 0; j < ncols; ++j)
            for (int64_t i = 0; i < nrows; ++i) t.v[(size_t)(i * ncols + j)] = v[(size_t)(j * nrows + i)];
        return t;
    }
};

// compressed-column matrix with Eigen::SparseMatrix<double>'s raw accessors (int indices, like Eigen's default StorageIndex)
struct SparseMatrixD {
    int64_t nrows = 0, ncols = 0;
    std::vector<int> outer, inner;
    std::vector<double> val;
    int *outerIndexPtr() { return outer.data(); }
    int *innerIndexPtr() { return inner.data(); }
    double *valuePtr() { return val.data(); }
    const int *outerIndexPtr() const { return outer.data(); }
    const int *innerIndexPtr() const { return inner.data(); }
    const double *valuePtr() const { return val.data(); }
    long rows() const { return (long)nrows; }
    long cols() const { return (long)ncols; }
    long nonZeros() const { return (long)val.size(); }
    double sum() const { double s = 0; for (double x : val) s += x; return s; }
    void conservativeResize(int64_t r, int64_t c)                        // only ever grows here (c++/sample.cpp:120-121)
    {
        nrows = r;
        outer.resize((size_t)c + 1, outer.empty() ? 0 : outer.back());
        ncols = c;
    }
    SparseMatrixD transpose() const
    {
        SparseMatrixD t;
        t.nrows = ncols; t.ncols = nrows;
        t.outer.assign((size_t)nrows + 1, 0);
        for (int r : inner) ++t.outer[(size_t)r + 1];
        for (size_t i = 0; i < (size_t)nrows; ++i) t.outer[i + 1] += t.outer[i];
        t.inner.resize(inner.size()); t.val.resize(val.size());
        std::vector<int> at(t.outer.begin(), t.outer.end() - 1);
        for (int64_t c = 0; c < ncols; ++c)
            for (int p = outer[(size_t)c]; p < outer[(size_t)c + 1]; ++p) {
                const int q = at[(size_t)inner[(size_t)p]]++;
                t.inner[(size_t)q] = (int)c; t.val[(size_t)q] = val[(size_t)p];
            }
        return t;
    }
};

inline void read_matrix(const std::string &fname, SparseMatrixD &m)     // c++/io.h: read_matrix, through this repo's reader
{
    int64_t nr, nc, nnz, *cp; int32_t *ri; double *va;
    if (bpmf_io_read_sparse(fname.c_str(), &nr, &nc, &nnz, &cp, &ri, &va)) THROWERROR(bpmf_io_last_error());
    m.nrows = nr; m.ncols = nc;
    m.outer.assign(cp, cp + nc + 1); m.inner.assign(ri, ri + nnz); m.val.assign(va, va + nnz);
    bpmf_io_free(cp); bpmf_io_free(ri); bpmf_io_free(va);
}
inline void write_matrix(const std::string &fname, const SparseMatrixD &m)
{
    std::vector<int64_t> cp(m.outer.begin(), m.outer.end());
    if (bpmf_io_write_sparse(fname.c_str(), m.nrows, m.ncols, cp.data(), m.inner.data(), m.val.data())) THROWERROR(bpmf_io_last_error());
}
inline void write_matrix(const std::string &fname, const DenseD &m)
{
    if (bpmf_io_write_dense(fname.c_str(), m.nrows, m.ncols, m.data())) THROWERROR(bpmf_io_last_error());
}

struct HyperParams {                                                      // c++/bpmf.h:78-104: the four sampled members
    DenseD mu{num_latent, 1}, LambdaF{num_latent, num_latent}, LambdaU{num_latent, num_latent}, LambdaL{num_latent, num_latent};
};

struct Sys {
    static bool verbose;                                                  // c++/bpmf.h:113-119
    static int nprocs, procid;
    static int burnin, nsims;
    static double alpha;
    static std::string odirname;
    static void Init(); static void Finalize(); static void Abort(int); static void sync();       // :121-124, defined by the back-end header
    static std::ostream &cout() { return std::cout; }

    std::string name;
    int iter;                                                             // :139
    SparseMatrixD M;                                                      // :147
    double mean_rating = 0;
    int num() const { return (int)M.cols(); }
    int nnz() const { return (int)M.nonZeros(); }
    double *items_ptr = nullptr;                                          // :193
    DenseD aggrMu, aggrLambda;                                            // :209
    DenseD cov{num_latent, num_latent};                                   // :222
    double norm = 0;                                                      // :223
    HyperParams hp;
    SparseMatrixD T, Pavg, Pm2;                                           // :229-230
    double rmse = NAN, rmse_avg = NAN;
    int num_predict = 0;

    // c++/sample.cpp:112-127: train + test files, both grown to the larger of the two shapes
    Sys(std::string nm, std::string fname, std::string probename) : name(nm), iter(-1)
    {
        read_matrix(fname, M);
        read_matrix(probename, T);
        const int64_t r = std::max(M.nrows, T.nrows), c = std::max(M.ncols, T.ncols);
        M.conservativeResize(r, c); T.conservativeResize(r, c);
        Pm2 = Pavg = T;
    }
    // c++/sample.cpp:132-137: the other factor works on the transposed matrices
    Sys(std::string nm, const SparseMatrixD &Mt, const SparseMatrixD &Pt) : name(nm), iter(-1)
    {
        M = Mt.transpose();
        Pm2 = Pavg = T = Pt.transpose();
    }
    virtual ~Sys() {}
    virtual void alloc_and_init() = 0;                                    // :144
    virtual void send_item(int i) = 0;                                    // :216
    virtual void sample(Sys &in) = 0;                                     // :219 (every back-end header overrides it)

    // c++/sample.cpp:179-226 without the statistics lines: mean rating, zero factors, zero posterior sums when -o is given
    void init()
    {
        mean_rating = M.sum() / (double)M.nonZeros();
        for (int64_t i = 0; i < (int64_t)num_latent * num(); ++i) items_ptr[i] = 0.0;
        norm = 0.0;
        if (Sys::odirname.size()) {
            aggrMu = DenseD(num_latent, num());
            aggrLambda = DenseD((int64_t)num_latent * num_latent, num());
        }
        Sys::cout() << "mean rating: " << mean_rating << std::endl;
        Sys::cout() << "num " << name << ": " << num() << std::endl;
    }

    // c++/sample.cpp:101-107
    void print(double items_per_sec, double ratings_per_sec, double norm_u, double norm_m)
    {
        char buf[1024];
        snprintf(buf, sizeof buf, "%d: %s iteration %d:\t RMSE: %3.4f\tavg RMSE: %3.4f\tFU(%6.2f)\tFM(%6.2f)\titems/sec: %6.2f\tratings/sec: %6.2fM\n",
                 Sys::procid, (iter < Sys::burnin) ? "Burnin" : "Sampling", iter, rmse, rmse_avg, norm_u, norm_m, items_per_sec, ratings_per_sec / 1e6);
        Sys::cout() << buf;
    }
};
