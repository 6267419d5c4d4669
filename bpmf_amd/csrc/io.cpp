// io.cpp -- readers / writers for the matrix files of the bpmf command line.
//
// Own implementation of the reference's file surface (c++/io.cpp): a byte source / sink that is
// either a plain file or a zlib stream, a whitespace tokenizer for the text formats, and CSC
// assembly by counting sort.  Formats and quirks kept: extension dispatch with optional ".gz"
// (io.cpp:31-77), 1-based u32 indices in .sdm/.sbm (:256-314), MatrixMarket coordinate
// real/integer/pattern + array, `%` comment lines (:414-522), duplicates summed (:521),
// "File '...' not found" (:117), 6 significant digits in text output (:697-718).
#include "io.h"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <sstream>

#include "../../include/bpmf_io.h"

namespace bpmf {
namespace io {
namespace {

bool ends_with(const std::string &s, const std::string &suf)
{
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// ---- byte source: whole file in memory (plain or gz) -----------------------------------------
std::string slurp(const std::string &path, bool gz)
{
    std::string out;
    if (gz) {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) throw IoError("File '" + path + "' not found");
        char buf[1 << 16];
        int n;
        while ((n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
        const bool bad = n < 0;
        gzclose(f);
        if (bad) throw IoError("Error reading compressed file: " + path);
    } else {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw IoError("File '" + path + "' not found");
        char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
        fclose(f);
    }
    return out;
}

void spill(const std::string &path, bool gz, const std::string &bytes)
{
    if (gz) {
        gzFile f = gzopen(path.c_str(), "wb");
        if (!f) throw IoError("Error opening file: " + path);
        size_t off = 0;
        while (off < bytes.size()) {
            const unsigned chunk = (unsigned)std::min<size_t>(bytes.size() - off, 1u << 30);
            if (gzwrite(f, bytes.data() + off, chunk) != (int)chunk) { gzclose(f); throw IoError("Error writing file: " + path); }
            off += chunk;
        }
        gzclose(f);
    } else {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) throw IoError("Error opening file: " + path);
        const size_t n = fwrite(bytes.data(), 1, bytes.size(), f);
        fclose(f);
        if (n != bytes.size()) throw IoError("Error writing file: " + path);
    }
}

// ---- binary cursor -------------------------------------------------------------------------------
struct Cursor {
    const std::string &b;
    size_t pos = 0;
    template <typename T>
    void get(T *dst, size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (pos + bytes > b.size()) throw IoError("unexpected end of file");
        memcpy(dst, b.data() + pos, bytes);
        pos += bytes;
    }
};

template <typename T>
void put(std::string &out, const T *src, size_t n) { out.append(reinterpret_cast<const char *>(src), n * sizeof(T)); }

// ---- text lines ---------------------------------------------------------------------------------
struct Lines {
    const std::string &b;
    size_t pos = 0;
    bool next(std::string &line)
    {
        if (pos >= b.size()) return false;
        const size_t e = b.find('\n', pos);
        line = b.substr(pos, (e == std::string::npos ? b.size() : e) - pos);
        pos = (e == std::string::npos) ? b.size() : e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
    // next line that is neither blank nor a % comment (c++/io.cpp:471-472,496-497)
    bool next_data(std::string &line)
    {
        while (next(line)) {
            size_t i = 0;
            while (i < line.size() && isspace((unsigned char)line[i])) ++i;
            if (i == line.size() || line[i] == '%') continue;
            return true;
        }
        return false;
    }
};

std::string upper(std::string s)
{
    for (auto &c : s) c = (char)toupper((unsigned char)c);
    return s;
}

struct MmHeader { bool coordinate = false, pattern = false; };

MmHeader mm_header(Lines &in)
{
    std::string line;
    if (!in.next(line) || line.compare(0, 14, "%%MatrixMarket") != 0) throw IoError("Could not process Matrix Market banner");
    std::istringstream ss(line.substr(14));
    std::string object, format, field, symmetry;
    ss >> object >> format >> field >> symmetry;
    object = upper(object); format = upper(format); field = upper(field); symmetry = upper(symmetry);
    if (object != "MATRIX") throw IoError("Invalid Matrix Market object: " + object);
    MmHeader h;
    if (format == "COORDINATE") h.coordinate = true;
    else if (format != "ARRAY") throw IoError("Invalid Matrix Market format: " + format);
    if (field == "PATTERN") h.pattern = true;
    else if (field != "REAL" && field != "INTEGER" && field != "DOUBLE") throw IoError("Unsupported Matrix Market field: " + field);
    if (symmetry != "GENERAL") throw IoError("Unsupported Matrix Market symmetry: " + symmetry);
    return h;
}

Csc read_mtx_sparse(const std::string &bytes)
{
    Lines in{bytes};
    const MmHeader h = mm_header(in);
    if (!h.coordinate) throw IoError("Matrix Market file is not in coordinate format");
    std::string line;
    if (!in.next_data(line)) throw IoError("Matrix Market size line missing");
    long long nr = 0, nc = 0, nnz = 0;
    if (sscanf(line.c_str(), "%lld %lld %lld", &nr, &nc, &nnz) != 3 || nr < 0 || nc < 0 || nnz < 0)
        throw IoError("Could not read Matrix Market size line");
    std::vector<int32_t> rows((size_t)nnz), cols((size_t)nnz);
    std::vector<double> vals((size_t)nnz);
    for (long long k = 0; k < nnz; ++k) {
        if (!in.next_data(line)) throw IoError("Matrix Market file has fewer entries than declared");
        char *p = nullptr;
        const long long r = strtoll(line.c_str(), &p, 10);
        const long long c = strtoll(p, &p, 10);
        const double v = h.pattern ? 1.0 : strtod(p, &p);
        if (r < 1 || r > nr || c < 1 || c > nc) throw IoError("Matrix Market entry out of range: " + line);
        rows[(size_t)k] = (int32_t)(r - 1); cols[(size_t)k] = (int32_t)(c - 1); vals[(size_t)k] = v;
    }
    return csc_from_triplets(nr, nc, rows, cols, vals);
}

Dense read_mtx_dense(const std::string &bytes)
{
    Lines in{bytes};
    const MmHeader h = mm_header(in);
    if (h.coordinate) throw IoError("Matrix Market file is not in array format");
    std::string line;
    if (!in.next_data(line)) throw IoError("Matrix Market size line missing");
    long long nr = 0, nc = 0;
    if (sscanf(line.c_str(), "%lld %lld", &nr, &nc) != 2 || nr < 0 || nc < 0) throw IoError("Could not read Matrix Market size line");
    Dense d;
    d.nrows = nr; d.ncols = nc; d.data.resize((size_t)nr * (size_t)nc);
    for (size_t k = 0; k < d.data.size(); ++k) {                     // column-major, one value per line (:382-409)
        if (!in.next_data(line)) throw IoError("Matrix Market file has fewer values than declared");
        d.data[k] = strtod(line.c_str(), nullptr);
    }
    return d;
}

Csc read_binary_sparse(const std::string &bytes, bool with_values, const std::string &path)
{
    Cursor in{bytes};
    uint64_t nr, nc, nnz;
    in.get(&nr, 1); in.get(&nc, 1); in.get(&nnz, 1);
    std::vector<uint32_t> r(nnz), c(nnz);
    std::vector<double> v(nnz, 1.0);
    in.get(r.data(), nnz); in.get(c.data(), nnz);
    if (with_values) in.get(v.data(), nnz);
    std::vector<int32_t> rows(nnz), cols(nnz);
    for (uint64_t k = 0; k < nnz; ++k) {
        if (r[k] < 1 || r[k] > nr || c[k] < 1 || c[k] > nc) throw IoError("index out of range in " + path);
        rows[k] = (int32_t)(r[k] - 1); cols[k] = (int32_t)(c[k] - 1);
    }
    Csc m = csc_from_triplets((int64_t)nr, (int64_t)nc, rows, cols, v);
    if (with_values && (uint64_t)m.nnz() != nnz) throw IoError("Invalid number of values");     // c++/io.cpp:284-287
    return m;
}

Dense read_csv(const std::string &bytes)
{
    Lines in{bytes};
    std::string line;
    Dense d;
    if (!in.next(line)) throw IoError("csv: missing row count");
    d.nrows = atoll(line.c_str());
    if (!in.next(line)) throw IoError("csv: missing column count");
    d.ncols = atoll(line.c_str());
    d.data.assign((size_t)d.nrows * (size_t)d.ncols, 0.0);
    for (int64_t r = 0; r < d.nrows; ++r) {
        if (!in.next(line)) throw IoError("invalid number of rows");
        std::istringstream ss(line);
        std::string cell;
        int64_t c = 0;
        while (c < d.ncols && std::getline(ss, cell, ',')) d.data[(size_t)c++ * d.nrows + r] = strtod(cell.c_str(), nullptr);
        if (c != d.ncols) throw IoError("invalid number of columns");
    }
    return d;
}

std::string fmt6(double v)                                    // default ostream formatting: %g, 6 significant digits
{
    char buf[64];
    snprintf(buf, sizeof buf, "%g", v);
    return buf;
}

}  // namespace

FileType file_type(const std::string &path)
{
    FileType t;
    std::string p = path;
    if (ends_with(p, ".gz")) { t.gz = true; p.resize(p.size() - 3); }
    if (ends_with(p, ".sdm")) t.kind = Kind::sdm;
    else if (ends_with(p, ".sbm")) t.kind = Kind::sbm;
    else if (ends_with(p, ".mtx") || ends_with(p, ".mm")) t.kind = Kind::mtx;
    else if (ends_with(p, ".csv")) t.kind = Kind::csv;
    else if (ends_with(p, ".ddm")) t.kind = Kind::ddm;
    return t;
}

Csc csc_from_triplets(int64_t nrows, int64_t ncols, const std::vector<int32_t> &rows, const std::vector<int32_t> &cols,
                      const std::vector<double> &vals)
{
    const size_t n = rows.size();
    // counting sort by column, then sort each column's run by row and fold duplicates
    std::vector<int64_t> start((size_t)ncols + 1, 0);
    for (size_t k = 0; k < n; ++k) start[(size_t)cols[k] + 1]++;
    for (int64_t c = 0; c < ncols; ++c) start[(size_t)c + 1] += start[(size_t)c];
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    std::vector<int32_t> r2(n);
    std::vector<double> v2(n);
    for (size_t k = 0; k < n; ++k) {
        const int64_t p = fill[(size_t)cols[k]]++;
        r2[(size_t)p] = rows[k]; v2[(size_t)p] = vals[k];
    }
    Csc m;
    m.nrows = nrows; m.ncols = ncols;
    m.colptr.assign((size_t)ncols + 1, 0);
    m.rowidx.reserve(n); m.vals.reserve(n);
    std::vector<size_t> perm;
    for (int64_t c = 0; c < ncols; ++c) {
        const size_t b = (size_t)start[(size_t)c], e = (size_t)start[(size_t)c + 1];
        perm.resize(e - b);
        std::iota(perm.begin(), perm.end(), b);
        std::stable_sort(perm.begin(), perm.end(), [&](size_t x, size_t y) { return r2[x] < r2[y]; });
        for (size_t q = 0; q < perm.size(); ++q) {
            const size_t k = perm[q];
            if (q > 0 && r2[k] == m.rowidx.back() && (int64_t)m.rowidx.size() > m.colptr[(size_t)c]) m.vals.back() += v2[k];
            else { m.rowidx.push_back(r2[k]); m.vals.push_back(v2[k]); }
        }
        m.colptr[(size_t)c + 1] = (int64_t)m.rowidx.size();
    }
    return m;
}

Csc transpose(const Csc &m)
{
    Csc t;
    t.nrows = m.ncols; t.ncols = m.nrows;
    t.colptr.assign((size_t)t.ncols + 1, 0);
    const size_t n = (size_t)m.nnz();
    for (size_t k = 0; k < n; ++k) t.colptr[(size_t)m.rowidx[k] + 1]++;
    for (int64_t c = 0; c < t.ncols; ++c) t.colptr[(size_t)c + 1] += t.colptr[(size_t)c];
    std::vector<int64_t> fill(t.colptr.begin(), t.colptr.end() - 1);
    t.rowidx.resize(n); t.vals.resize(n);
    for (int64_t c = 0; c < m.ncols; ++c)                          // ascending source column => ascending row in t
        for (int64_t p = m.colptr[(size_t)c]; p < m.colptr[(size_t)c + 1]; ++p) {
            const int64_t q = fill[(size_t)m.rowidx[(size_t)p]]++;
            t.rowidx[(size_t)q] = (int32_t)c; t.vals[(size_t)q] = m.vals[(size_t)p];
        }
    return t;
}

void resize(Csc &m, int64_t nrows, int64_t ncols)
{
    if (nrows < m.nrows || ncols < m.ncols) throw IoError("resize: shrinking is not supported");
    m.nrows = nrows;
    m.colptr.resize((size_t)ncols + 1, m.colptr.empty() ? 0 : m.colptr.back());
    m.ncols = ncols;
}

Csc read_sparse(const std::string &path)
{
    const FileType t = file_type(path);
    switch (t.kind) {
    case Kind::sdm: return read_binary_sparse(slurp(path, t.gz), true, path);
    case Kind::sbm: return read_binary_sparse(slurp(path, t.gz), false, path);
    case Kind::mtx: return read_mtx_sparse(slurp(path, t.gz));
    case Kind::csv: case Kind::ddm: throw IoError("Invalid matrix type");
    default: throw IoError("Unknown matrix type");
    }
}

Dense read_dense(const std::string &path)
{
    const FileType t = file_type(path);
    switch (t.kind) {
    case Kind::ddm: {
        const std::string b = slurp(path, t.gz);
        Cursor in{b};
        uint64_t nr, nc;
        in.get(&nr, 1); in.get(&nc, 1);
        Dense d;
        d.nrows = (int64_t)nr; d.ncols = (int64_t)nc; d.data.resize(nr * nc);
        in.get(d.data.data(), nr * nc);
        return d;
    }
    case Kind::mtx: return read_mtx_dense(slurp(path, t.gz));
    case Kind::csv: return read_csv(slurp(path, t.gz));
    case Kind::sdm: case Kind::sbm: throw IoError("Invalid matrix type");
    default: throw IoError("Unknown matrix type");
    }
}

void write_sparse(const std::string &path, const Csc &m)
{
    const FileType t = file_type(path);
    std::string out;
    const uint64_t nr = (uint64_t)m.nrows, nc = (uint64_t)m.ncols;
    if (t.kind == Kind::sdm || t.kind == Kind::sbm) {
        std::vector<uint32_t> r, c;
        std::vector<double> v;
        for (int64_t col = 0; col < m.ncols; ++col)
            for (int64_t p = m.colptr[(size_t)col]; p < m.colptr[(size_t)col + 1]; ++p) {
                if (t.kind == Kind::sbm && !(m.vals[(size_t)p] > 0)) continue;        // c++/io.cpp:666
                r.push_back((uint32_t)m.rowidx[(size_t)p] + 1); c.push_back((uint32_t)col + 1); v.push_back(m.vals[(size_t)p]);
            }
        const uint64_t nnz = r.size();
        put(out, &nr, 1); put(out, &nc, 1); put(out, &nnz, 1);
        put(out, r.data(), r.size()); put(out, c.data(), c.size());
        if (t.kind == Kind::sdm) put(out, v.data(), v.size());
    } else if (t.kind == Kind::mtx) {
        out = "%%MatrixMarket matrix coordinate real general\n";
        out += std::to_string(m.nrows) + " " + std::to_string(m.ncols) + " " + std::to_string(m.nnz()) + "\n";
        for (int64_t col = 0; col < m.ncols; ++col)
            for (int64_t p = m.colptr[(size_t)col]; p < m.colptr[(size_t)col + 1]; ++p)
                out += std::to_string(m.rowidx[(size_t)p] + 1) + " " + std::to_string(col + 1) + " " + fmt6(m.vals[(size_t)p]) + "\n";
    } else {
        throw IoError(t.kind == Kind::none ? "Unknown matrix type" : "Invalid matrix type");
    }
    spill(path, t.gz, out);
}

void write_dense(const std::string &path, const Dense &d)
{
    const FileType t = file_type(path);
    std::string out;
    if (t.kind == Kind::ddm) {
        const uint64_t nr = (uint64_t)d.nrows, nc = (uint64_t)d.ncols;
        put(out, &nr, 1); put(out, &nc, 1); put(out, d.data.data(), d.data.size());
    } else if (t.kind == Kind::mtx) {
        out = "%%MatrixMarket matrix array real general\n" + std::to_string(d.nrows) + " " + std::to_string(d.ncols) + "\n";
        for (double v : d.data) out += fmt6(v) + "\n";
    } else if (t.kind == Kind::csv) {
        out = std::to_string(d.nrows) + "\n" + std::to_string(d.ncols) + "\n";
        for (int64_t r = 0; r < d.nrows; ++r) {
            for (int64_t c = 0; c < d.ncols; ++c) out += (c ? "," : "") + fmt6(d.data[(size_t)c * d.nrows + r]);
            out += "\n";
        }
    } else {
        throw IoError(t.kind == Kind::none ? "Unknown matrix type" : "Invalid matrix type");
    }
    spill(path, t.gz, out);
}

}  // namespace io
}  // namespace bpmf

// ---- C ABI (include/bpmf_io.h) ----------------------------------------------------------------------
namespace {
thread_local std::string g_io_err;
template <typename T>
T *dup(const std::vector<T> &v)
{
    T *p = (T *)malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
template <typename F>
int guarded(F f)
{
    try { f(); return 0; }
    catch (const std::exception &e) { g_io_err = e.what(); return -1; }
}
}  // namespace

extern "C" const char *bpmf_io_last_error(void) { return g_io_err.c_str(); }
extern "C" void bpmf_io_free(void *p) { free(p); }

extern "C" int bpmf_io_read_sparse(const char *path, int64_t *nrows, int64_t *ncols, int64_t *nnz, int64_t **colptr,
                                   int32_t **rowidx, double **vals)
{
    return guarded([&] {
        const bpmf::io::Csc m = bpmf::io::read_sparse(path);
        *nrows = m.nrows; *ncols = m.ncols; *nnz = m.nnz();
        *colptr = dup(m.colptr); *rowidx = dup(m.rowidx); *vals = dup(m.vals);
    });
}

extern "C" int bpmf_io_write_sparse(const char *path, int64_t nrows, int64_t ncols, const int64_t *colptr,
                                    const int32_t *rowidx, const double *vals)
{
    return guarded([&] {
        bpmf::io::Csc m;
        m.nrows = nrows; m.ncols = ncols;
        m.colptr.assign(colptr, colptr + ncols + 1);
        m.rowidx.assign(rowidx, rowidx + colptr[ncols]);
        m.vals.assign(vals, vals + colptr[ncols]);
        bpmf::io::write_sparse(path, m);
    });
}

extern "C" int bpmf_io_read_dense(const char *path, int64_t *nrows, int64_t *ncols, double **data)
{
    return guarded([&] {
        const bpmf::io::Dense d = bpmf::io::read_dense(path);
        *nrows = d.nrows; *ncols = d.ncols; *data = dup(d.data);
    });
}

extern "C" int bpmf_io_write_dense(const char *path, int64_t nrows, int64_t ncols, const double *data)
{
    return guarded([&] {
        bpmf::io::Dense d;
        d.nrows = nrows; d.ncols = ncols;
        d.data.assign(data, data + (size_t)nrows * (size_t)ncols);
        bpmf::io::write_dense(path, d);
    });
}
