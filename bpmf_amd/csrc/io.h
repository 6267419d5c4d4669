// io.h -- matrix files of the bpmf command line (see include/bpmf_io.h for the formats).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace bpmf {
namespace io {

struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

struct Csc {                       // sorted rows per column, duplicates summed
    int64_t nrows = 0, ncols = 0;
    std::vector<int64_t> colptr;   // ncols + 1
    std::vector<int32_t> rowidx;
    std::vector<double> vals;
    int64_t nnz() const { return colptr.empty() ? 0 : colptr.back(); }
};

struct Dense {                     // column-major
    int64_t nrows = 0, ncols = 0;
    std::vector<double> data;
};

enum class Kind { none, sdm, sbm, mtx, csv, ddm };
struct FileType { Kind kind = Kind::none; bool gz = false; };
FileType file_type(const std::string &path);

// triplets (0-based) -> CSC; rows sorted, duplicates summed, explicit zeros kept
Csc csc_from_triplets(int64_t nrows, int64_t ncols, const std::vector<int32_t> &rows, const std::vector<int32_t> &cols,
                      const std::vector<double> &vals);
Csc transpose(const Csc &m);
// grows the shape without touching the entries (conservativeResize, c++/sample.cpp:119-122)
void resize(Csc &m, int64_t nrows, int64_t ncols);

Csc read_sparse(const std::string &path);
Dense read_dense(const std::string &path);
void write_sparse(const std::string &path, const Csc &m);
void write_dense(const std::string &path, const Dense &m);

}  // namespace io
}  // namespace bpmf
