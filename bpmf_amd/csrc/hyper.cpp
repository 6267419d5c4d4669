// hyper.cpp -- host-side Normal-Wishart hyper-parameter draw.
//
// The north-star keeps this step on the host.  It replaces HyperParams::sample
// (c++/bpmf.h:98-103) and the chain CondNormalWishart -> NormalWishart ->
// WishartChol -> WishartUnitChol / MvNormalChol_prec (c++/mvnormal.cpp:56-135).
// Seed-for-seed parity needs the *same* consumption of the Philox stream as the
// reference, so the draws go through libstdc++'s std::normal_distribution /
// std::gamma_distribution themselves (c++/mvnormal.cpp:42,68) on a URNG with
// the MicroURNG word order (philox.h).  Dense algebra is a few small
// hand-written routines (no Eigen): K is at most 128.
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/bpmf_hip.h"
#include "philox.h"

namespace {

struct Mat {   // column-major K x K view
    double *p; int K;
    double &operator()(int i, int j) const { return p[(size_t)j * K + i]; }
};

double randn(bpmf::MicroPhilox &rng) { return std::normal_distribution<>()(rng); }   // c++/mvnormal.cpp:41-43

// Advances the stream exactly as `n` calls of randn() would, without the log/sqrt of the
// accepted attempt: the reference draws `nrandn(K-i-1)` into a vector it never reads
// (c++/mvnormal.cpp:70).  Same acceptance test as libstdc++'s polar loop, evaluated un-fused.
void skip_randn(bpmf::MicroPhilox &rng, int n)
{
    for (int i = 0; i < n; ++i) {
        double r2;
        do {
            const uint32_t a0 = rng(), a1 = rng(), b0 = rng(), b1 = rng();
            const double x = 2.0 * bpmf::canonical53(a0, a1) - 1.0;
            const double y = 2.0 * bpmf::canonical53(b0, b1) - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
    }
}

// inverse through LU with row pivoting (role of Eigen's inverse(), c++/mvnormal.cpp:124);
// all inner loops run down a column (contiguous)
bool invert(int K, const double *A_in, double *inv)
{
    std::vector<double> lu(A_in, A_in + (size_t)K * K);
    std::vector<int> perm(K);
    Mat A{lu.data(), K};
    for (int i = 0; i < K; ++i) perm[i] = i;
    for (int c = 0; c < K; ++c) {
        int best = c;
        for (int r = c + 1; r < K; ++r)
            if (std::fabs(A(r, c)) > std::fabs(A(best, c))) best = r;
        if (A(best, c) == 0.0) return false;
        if (best != c) {
            for (int j = 0; j < K; ++j) std::swap(A(c, j), A(best, j));
            std::swap(perm[c], perm[best]);
        }
        const double piv = A(c, c);
        double *lc = &A(0, c);
        for (int r = c + 1; r < K; ++r) lc[r] /= piv;
        for (int j = c + 1; j < K; ++j) {
            double *aj = &A(0, j);
            const double f = aj[c];
            for (int r = c + 1; r < K; ++r) aj[r] -= lc[r] * f;
        }
    }
    for (int c = 0; c < K; ++c) {
        double *x = inv + (size_t)c * K;
        int first = K;
        for (int r = 0; r < K; ++r) { x[r] = perm[r] == c ? 1.0 : 0.0; if (perm[r] == c) first = r; }
        for (int j = first; j < K; ++j) {      // forward, unit lower: x_r -= L(r,j) x_j
            const double xj = x[j];
            const double *lj = &A(0, j);
            for (int r = j + 1; r < K; ++r) x[r] -= lj[r] * xj;
        }
        for (int j = K - 1; j >= 0; --j) {     // backward, upper
            const double *uj = &A(0, j);
            const double xj = (x[j] /= uj[j]);
            for (int r = 0; r < j; ++r) x[r] -= uj[r] * xj;
        }
    }
    return true;
}

// lower Cholesky factor from the lower triangle (sigma.llt(), c++/mvnormal.cpp:78); right-looking,
// every inner loop runs down a column (contiguous)
bool cholesky_lower(int K, const double *S_in, double *L_out)
{
    std::memset(L_out, 0, sizeof(double) * K * K);
    Mat L{L_out, K};
    for (int c = 0; c < K; ++c)
        for (int r = c; r < K; ++r) L(r, c) = S_in[(size_t)c * K + r];
    for (int c = 0; c < K; ++c) {
        double *lc = &L(0, c);
        const double d = lc[c];
        if (!(d > 0.0)) return false;
        const double sd = std::sqrt(d);
        lc[c] = sd;
        for (int r = c + 1; r < K; ++r) lc[r] /= sd;
        for (int j = c + 1; j < K; ++j) {
            double *lj = &L(0, j);
            const double f = lc[j];
            for (int r = j; r < K; ++r) lj[r] -= lc[r] * f;
        }
    }
    return true;
}

// K >= 128 (the fp32 large-K configuration): chol(X^-1).matrixU() WITHOUT the inverse.  X is
// symmetric positive definite; with its "reverse" Cholesky factorisation X = Ux Ux^T (Ux upper
// triangular, positive diagonal) X^-1 = Ux^-T Ux^-1 = R^T R with R = Ux^-1 upper triangular and
// positive on the diagonal -- the unique Cholesky factor the reference obtains as
// inverse().llt().matrixU() (c++/mvnormal.cpp:78,124).  ~K^3/2 multiply-adds instead of ~2 K^3:
// at K = 128 the host draw drops from ~630 us to ~250 us, which matters because a side's
// sampler -> statistics -> draw -> next sampler loop (not the sum of the two samplers) bounds the
// iteration there.  Different operation order than the reference's LU + LLT, hence kept to the
// configuration whose tolerance is the fp32 one (tests/test_gpu_f32.py); K <= 64 follows the
// reference's order.  Writes the LOWER factor L (T_c = L L^T, L = R^T) column-major like cholesky_lower.
bool inverse_factor_spd(int K, const double *X_in, double *L_out)
{
    // (buffers are kept per thread: a K x K double vector is exactly glibc's mmap threshold at K = 128,
    //  i.e. an mmap, 32 page faults and an munmap per call and vector)
    static thread_local std::vector<double> ux;
    ux.assign(X_in, X_in + (size_t)K * K);
    Mat U{ux.data(), K};
    for (int c = K - 1; c >= 0; --c) {                   // right-looking from the last column: contiguous inner loops
        double *uc = &U(0, c);
        const double d = uc[c];
        if (!(d > 0.0)) return false;
        const double sd = std::sqrt(d);
        uc[c] = sd;
        for (int r = 0; r < c; ++r) uc[r] /= sd;
        for (int j = 0; j < c; ++j) {                     // X(0..j, j) -= Ux(0..j, c) Ux(j, c)
            double *uj = &U(0, j);
            const double f = uc[j];
            for (int r = 0; r <= j; ++r) uj[r] -= uc[r] * f;
        }
    }
    std::memset(L_out, 0, sizeof(double) * K * K);
    static thread_local std::vector<double> x;
    x.assign(K, 0.0);
    for (int c = 0; c < K; ++c) {                         // column c of R = Ux^-1: Ux x = e_c, x_r = 0 for r > c
        for (int r = 0; r < c; ++r) x[r] = 0.0;
        x[c] = 1.0;
        for (int j = c; j >= 0; --j) {
            const double *uj = &U(0, j);
            const double xj = (x[j] /= uj[j]);
            for (int r = 0; r < j; ++r) x[r] -= uj[r] * xj;
        }
        for (int r = 0; r <= c; ++r) L_out[(size_t)r * K + c] = x[r];   // L(c, r) = R(r, c)
    }
    return true;
}

}  // namespace

extern "C" void bpmf_hip_set_error_(const char *msg);   // capi.cpp

// The random part of the draw does not depend on cov: the unit-Wishart factor `au` (gamma and
// normal draws of WishartUnitChol, c++/mvnormal.cpp:64-73, with df = K + N) and the K normals `z`
// of MvNormalChol_prec (:58) consume the Philox stream `counter` in a data-independent way.  It
// can therefore be produced ahead of time, before the sums of the half-iteration have arrived.
extern "C" int bpmf_hyper_draws(int K, int64_t N, uint32_t counter, double *au_out, double *z_out)
{
    if (K <= 0 || K > 1024 || N <= 0 || !au_out || !z_out) {
        bpmf_hip_set_error_("bpmf_hyper_draws: bad argument");
        return BPMF_HIP_EINVAL;
    }
    bpmf::MicroPhilox rng(counter);                      // rng_set_pos(iter), c++/sample.cpp:349
    const double nu_c = (double)((int64_t)K + N);          // nu + N with nu = df = K
    std::memset(au_out, 0, sizeof(double) * K * K);
    Mat AU{au_out, K};
    for (int i = 0; i < K; ++i) {                         // WishartUnitChol (c++/mvnormal.cpp:64-73)
        std::gamma_distribution<> gam(0.5 * (nu_c - i));
        AU(i, i) = std::sqrt(2.0 * gam(rng));
        skip_randn(rng, K - i - 1);                               // `VectorXd r = nrandn(...)`, drawn and dropped (:70)
        for (int j = i + 1; j < K; ++j) AU(i, j) = randn(rng);
    }
    for (int i = 0; i < K; ++i) z_out[i] = randn(rng);    // nrandn(num_latent) of MvNormalChol_prec (:58)
    return BPMF_HIP_OK;
}

// The part that needs cov: CondNormalWishart's posterior parameters, WishartChol's product and
// the triangular solve of MvNormalChol_prec (c++/mvnormal.cpp:56-61,75-92,116-135), LambdaF.
extern "C" int bpmf_hyper_finish(int K, int64_t N, const double *cov, const double *Um, const double *au, const double *z_in,
                                 double *mu, double *LambdaU, double *LambdaF)
{
    if (K <= 0 || K > 1024 || N <= 0 || !cov || !au || !z_in || !mu || !LambdaU || !LambdaF) {
        bpmf_hip_set_error_("bpmf_hyper_finish: bad argument");
        return BPMF_HIP_EINVAL;
    }
    const size_t KK = (size_t)K * K;
    // fixed prior (c++/bpmf.h:80-96): mu0 = 0, kappa = b0 = 2, T = WI = I, nu = df = K
    const double kappa = 2.0, dN = (double)N;
    static thread_local std::vector<double> mu_m, mu_c, X, Tc, R, z, W;
    mu_m.assign(K, 0.0); mu_c.assign(K, 0.0); X.resize(KK); Tc.resize(KK); R.resize(KK); z.assign(z_in, z_in + K); W.resize(KK);
    for (int i = 0; i < K; ++i) {
        const double um = Um ? Um[i] : 0.0;
        mu_m[i] = 0.0 - um;
        mu_c[i] = (kappa * 0.0 + dN * um) / (kappa + dN);
    }
    const double kappa_c = kappa + dN;
    const double kappa_m = (kappa * dN) / (kappa + dN);
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < K; ++i)
            X[(size_t)j * K + i] = ((i == j ? 1.0 : 0.0) + dN * cov[(size_t)j * K + i]) + kappa_m * (mu_m[i] * mu_m[j]);
    if (K >= 128) {
        if (!inverse_factor_spd(K, X.data(), R.data())) {
            bpmf_hip_set_error_("bpmf_hyper_sample: posterior scale matrix not positive definite");
            return BPMF_HIP_ENUM;
        }
    } else {
    if (!invert(K, X.data(), Tc.data())) {
        bpmf_hip_set_error_("bpmf_hyper_sample: singular posterior scale matrix");
        return BPMF_HIP_ENUM;
    }
    // WishartChol (c++/mvnormal.cpp:75-92): U = au * chol(T_c).matrixU()
    if (!cholesky_lower(K, Tc.data(), R.data())) {
        bpmf_hip_set_error_("bpmf_hyper_sample: posterior scale matrix not positive definite");
        return BPMF_HIP_ENUM;
    }
    }
    Mat U{LambdaU, K}, F{LambdaF, K};
    // U(i,j) = sum_{k=i..j} au(i,k) * matrixU(k,j), matrixU(k,j) = R(j,k).  Written as column updates
    // U(0..k, j) += au(0..k, k) * R(j,k), k ascending: every entry still adds its terms in the order
    // k = i, i+1, ..., j (bit-identical to the dot-product form) but the inner loop runs down a
    // contiguous column and vectorises without re-association.
    std::memset(LambdaU, 0, sizeof(double) * KK);
    for (int j = 0; j < K; ++j) {
        double *uj = &U(0, j);
        for (int k = 0; k <= j; ++k) {
            const double f = R[(size_t)k * K + j];
            const double *ak = au + (size_t)k * K;
            for (int i = 0; i <= k; ++i) uj[i] += ak[i] * f;
        }
    }
    // MvNormalChol_prec (c++/mvnormal.cpp:56-61)
    for (int i = K - 1; i >= 0; --i) {
        double s = z[i];
        for (int j = i + 1; j < K; ++j) s -= U(i, j) * z[j];
        z[i] = s / U(i, i);
    }
    const double sk = std::sqrt(kappa_c);
    for (int i = 0; i < K; ++i) mu[i] = z[i] / sk + mu_c[i];
    {   // LambdaF = LambdaU^T LambdaU (c++/bpmf.h:101): F(i,j) = sum_{k <= min(i,j)} U(k,i) U(k,j), again as
        // column updates F(k.., j) += W(k.., k) * U(k,j) with W = U^T (row k of U made contiguous), k ascending
        for (int c = 0; c < K; ++c)
            for (int r = 0; r < K; ++r) W[(size_t)r * K + c] = U(r, c);      // W(c, r) = U(r, c): column r of W = row r of U
        std::memset(LambdaF, 0, sizeof(double) * KK);
        for (int j = 0; j < K; ++j) {
            double *fj = &F(0, j);
            for (int k = 0; k <= j; ++k) {
                const double f = U(k, j);
                const double *wk = &W[(size_t)k * K];
                for (int i = k; i < K; ++i) fj[i] += wk[i] * f;
            }
        }
    }
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hyper_sample(int K, int64_t N, const double *cov, const double *Um, uint32_t counter,
                                 double *mu, double *LambdaU, double *LambdaF)
{
    if (K <= 0 || K > 1024 || N <= 0 || !cov || !mu || !LambdaU || !LambdaF) {
        bpmf_hip_set_error_("bpmf_hyper_sample: bad argument");
        return BPMF_HIP_EINVAL;
    }
    std::vector<double> au((size_t)K * K), z(K);
    int rc = bpmf_hyper_draws(K, N, counter, au.data(), z.data());
    if (rc) return rc;
    return bpmf_hyper_finish(K, N, cov, Um, au.data(), z.data(), mu, LambdaU, LambdaF);
}

extern "C" void bpmf_cov_from_sums(int K, int64_t N, const double *sum, const double *prod, double *cov)
{
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < K; ++i)
            cov[(size_t)j * K + i] = (prod[(size_t)j * K + i] - (sum[i] * sum[j] / (double)N)) / (double)(N - 1);
}

extern "C" void bpmf_randn_stream(uint32_t counter, int n, double *out)
{
    bpmf::MicroPhilox rng(counter);
    for (int i = 0; i < n; ++i) out[i] = randn(rng);
}
