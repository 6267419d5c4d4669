"""INTEGRATION.md section 2 shows the back-end header (`c++/hip_sys.h`) a maintainer of the reference would add.  The
reference's own headers need Eigen3, which this image lacks, so the stub cannot be compiled into the reference here.
What CAN be checked: that the stub is valid C++ and that every call it makes matches include/bpmf_hip.h -- by
compiling it (syntax + types, -fsyntax-only) against a MOCK of exactly the members of `struct Sys` it touches
(names and types as declared at /root/reference c++/bpmf.h:113-124,139,144,193,216,219,222-223,231-232; the Eigen
types reduced to the three accessors the stub uses).  The mock lives only in this test; it is not a build of the reference."""
import os
import re
import subprocess
import tempfile

from tests.conftest import ROOT

MOCK = r'''
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <iostream>
#include <string>
#include <vector>
#include <stdexcept>
#define THROWERROR(msg) throw std::runtime_error(msg)              /* error.h:18-30 */
#define BPMF_COUNTER(name) do {} while (0)                          /* counters.h:60-66 */
static const int num_latent = 32;                                   /* bpmf.h:53 */
struct DenseMock { std::vector<double> v; double *data() { return v.data(); } DenseMock transpose() const { return *this; } };
struct SparseMatrixD {                                              /* Eigen::SparseMatrix<double>: the accessors the stub uses */
    int *outerIndexPtr(); int *innerIndexPtr(); double *valuePtr(); long rows() const;
};
struct HyperParams { DenseMock mu, LambdaF, LambdaU, LambdaL; };    /* bpmf.h:78-104 */
struct Sys {
    static bool verbose; static int nprocs, procid; static int burnin; static double alpha; static std::string odirname;   /* bpmf.h:113-119 */
    static void Init(); static void Finalize(); static void Abort(int); static void sync();                                /* :121-124 */
    int iter;                                                       /* :139 */
    Sys(std::string, std::string, std::string); Sys(std::string, const SparseMatrixD &, const SparseMatrixD &);
    virtual ~Sys();
    virtual void alloc_and_init() = 0;                              /* :144 */
    int num() const; void init();
    double *items_ptr;                                              /* :193 */
    virtual void send_item(int i) = 0;                              /* :216 */
    virtual void sample(Sys &in);                                   /* :219 */
    DenseMock cov; double norm;                                     /* :222-223 */
    double rmse, rmse_avg; int num_predict;                         /* :231-232 */
    double mean_rating; HyperParams hp; SparseMatrixD M, T, Pavg, Pm2;
};
'''


def test_hip_sys_stub_is_valid_cpp_against_the_c_abi():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", text, re.S)
    stub = [b for b in blocks if "struct HIP_Sys" in b]
    assert len(stub) == 1
    src = MOCK + stub[0].replace('#include "bpmf_hip.h"', '#include "bpmf_hip.h"')
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "hip_sys_check.cpp")
        open(f, "w").write(src)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused", "-I", os.path.join(ROOT, "include"), f],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
