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

// dst(c, r) = src(r, c) for the rows r >= r0(c) / all rows of a column-major K x K matrix, in 8 x 8 tiles: a plain double
// loop writes (or reads) with a stride of K doubles -- one cache line per element, 8 cycles per element at K = 128 where
// the tiles take ~1: the three transposes of the draw were a sixth of its time (tools/probes/hyper_bench.cpp)
void transpose_tiles(int K, const double *src, double *dst)
{
    constexpr int T = 8;
    for (int c0 = 0; c0 < K; c0 += T)
        for (int r0 = 0; r0 < K; r0 += T) {
            const int cn = c0 + T < K ? c0 + T : K, rn = r0 + T < K ? r0 + T : K;
            for (int c = c0; c < cn; ++c)
                for (int r = r0; r < rn; ++r) dst[(size_t)r * K + c] = src[(size_t)c * K + r];
        }
}

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
    // right-looking from the last column, contiguous inner loops.  Four pivot columns at a time: they are finished
    // among themselves first (each takes the rank-one updates of the ones before it), then every column to their left
    // takes the four updates in ONE pass -- per entry the same subtractions in the same order (c descending) as one
    // pivot column after the other, with its running value in a register across the four.
    auto finish_column = [&](int c) -> bool {            // sqrt of the pivot, scale the column above it
        double *uc = &U(0, c);
        const double d = uc[c];
        if (!(d > 0.0)) return false;
        const double sd = std::sqrt(d);
        uc[c] = sd;
        for (int r = 0; r < c; ++r) uc[r] /= sd;
        return true;
    };
    auto update_column = [&](int j, int c) {              // X(0..j, j) -= Ux(0..j, c) Ux(j, c)
        double *uj = &U(0, j);
        const double *uc = &U(0, c);
        const double f = uc[j];
        for (int r = 0; r <= j; ++r) uj[r] -= uc[r] * f;
    };
    int c = K - 1;
    for (; c >= 3; c -= 4) {
        if (!finish_column(c)) return false;
        update_column(c - 1, c); update_column(c - 2, c); update_column(c - 3, c);
        if (!finish_column(c - 1)) return false;
        update_column(c - 2, c - 1); update_column(c - 3, c - 1);
        if (!finish_column(c - 2)) return false;
        update_column(c - 3, c - 2);
        if (!finish_column(c - 3)) return false;
        const double *u0 = &U(0, c), *u1 = &U(0, c - 1), *u2 = &U(0, c - 2), *u3 = &U(0, c - 3);
        for (int j = 0; j < c - 3; ++j) {
            double *uj = &U(0, j);
            const double f0 = u0[j], f1 = u1[j], f2 = u2[j], f3 = u3[j];
            for (int r = 0; r <= j; ++r) {
                double t = uj[r];
                t -= u0[r] * f0; t -= u1[r] * f1; t -= u2[r] * f2; t -= u3[r] * f3;
                uj[r] = t;
            }
        }
    }
    for (; c >= 0; --c) {
        if (!finish_column(c)) return false;
        for (int j = 0; j < c; ++j) update_column(j, c);
    }
    static thread_local std::vector<double> rc;          // R column by column (contiguous), transposed into L_out at the end
    rc.assign((size_t)K * K, 0.0);
    for (int c = 0; c < K; ++c) {                         // column c of R = Ux^-1: Ux x = e_c, x_r = 0 for r > c
        double *x = rc.data() + (size_t)c * K;
        x[c] = 1.0;
        int j = c;
        for (; j >= 3; j -= 4) {                          // four steps of the back substitution at a time (same order per entry)
            const double *u0 = &U(0, j), *u1 = &U(0, j - 1), *u2 = &U(0, j - 2), *u3 = &U(0, j - 3);
            const double x0 = (x[j] /= u0[j]);
            x[j - 1] -= u0[j - 1] * x0;
            const double x1 = (x[j - 1] /= u1[j - 1]);
            x[j - 2] -= u0[j - 2] * x0; x[j - 2] -= u1[j - 2] * x1;
            const double x2 = (x[j - 2] /= u2[j - 2]);
            x[j - 3] -= u0[j - 3] * x0; x[j - 3] -= u1[j - 3] * x1; x[j - 3] -= u2[j - 3] * x2;
            const double x3 = (x[j - 3] /= u3[j - 3]);
            for (int r = 0; r < j - 3; ++r) {
                double t = x[r];
                t -= u0[r] * x0; t -= u1[r] * x1; t -= u2[r] * x2; t -= u3[r] * x3;
                x[r] = t;
            }
        }
        for (; j >= 0; --j) {
            const double *uj = &U(0, j);
            const double xj = (x[j] /= uj[j]);
            for (int r = 0; r < j; ++r) x[r] -= uj[r] * xj;
        }
    }
    transpose_tiles(K, rc.data(), L_out);                 // L(c, r) = R(r, c)
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
    // Four k at a time: an entry keeps its running sum in a register across the four terms instead of going through
    // memory after each -- the same additions in the same order (mul and add are separate roundings here), a quarter of
    // the loads / stores and loop set-ups of the accumulator column.
    std::memset(LambdaU, 0, sizeof(double) * KK);
    for (int j = 0; j < K; ++j) {
        double *uj = &U(0, j);
        int k = 0;
        for (; k + 3 <= j; k += 4) {
            const double f0 = R[(size_t)k * K + j], f1 = R[(size_t)(k + 1) * K + j], f2 = R[(size_t)(k + 2) * K + j], f3 = R[(size_t)(k + 3) * K + j];
            const double *a0 = au + (size_t)k * K, *a1 = a0 + K, *a2 = a1 + K, *a3 = a2 + K;
            for (int i = 0; i <= k; ++i) {
                double t = uj[i];
                t += a0[i] * f0; t += a1[i] * f1; t += a2[i] * f2; t += a3[i] * f3;
                uj[i] = t;
            }
            // ragged ends: i = k + 1 .. k + 3 take the terms of the k' >= i only
            uj[k + 1] = ((uj[k + 1] + a1[k + 1] * f1) + a2[k + 1] * f2) + a3[k + 1] * f3;
            uj[k + 2] = (uj[k + 2] + a2[k + 2] * f2) + a3[k + 2] * f3;
            uj[k + 3] = uj[k + 3] + a3[k + 3] * f3;
        }
        for (; k <= j; ++k) {
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
        transpose_tiles(K, LambdaU, W.data());               // W(c, r) = U(r, c): column r of W = row r of U
        // Only the lower triangle (i >= j) is accumulated: F(i,j) and F(j,i) are sums of the same products in the
        // same order (k ascending), i.e. bit-identical -- the upper triangle is a copy.  Half the multiply-adds of this
        // product.  (Round 3 also tried helper threads for the three stages of the draw whose result columns are
        // independent -- R = Ux^-1, U = au R^T, F -- with a spinning fork / join: bit-identical, but on the GPU boxes'
        // EPYC 259 us with one thread, 256 with three, 230-244 with six at K = 128: the 128 KB operands produced on one
        // core are cache misses on the others.  tools/probes/hyper_bench.cpp is the micro-benchmark.)
        std::memset(LambdaF, 0, sizeof(double) * KK);
        for (int j = 0; j < K; ++j) {
            double *fj = &F(0, j);
            int k = 0;
            for (; k + 3 <= j; k += 4) {                      // four k at a time (see LambdaU above): same sums, same order
                const double f0 = U(k, j), f1 = U(k + 1, j), f2 = U(k + 2, j), f3 = U(k + 3, j);
                const double *w0 = &W[(size_t)k * K], *w1 = w0 + K, *w2 = w1 + K, *w3 = w2 + K;
                for (int i = j; i < K; ++i) {
                    double t = fj[i];
                    t += w0[i] * f0; t += w1[i] * f1; t += w2[i] * f2; t += w3[i] * f3;
                    fj[i] = t;
                }
            }
            for (; k <= j; ++k) {
                const double f = U(k, j);
                const double *wk = &W[(size_t)k * K];
                for (int i = j; i < K; ++i) fj[i] += wk[i] * f;
            }
        }
        {   // the upper triangle: a copy of the lower one, tile by tile
            constexpr int T = 8;
            for (int j0 = 0; j0 < K; j0 += T)
                for (int i0 = j0; i0 < K; i0 += T) {
                    const int jn = j0 + T < K ? j0 + T : K, in = i0 + T < K ? i0 + T : K;
                    for (int j = j0; j < jn; ++j)
                        for (int i = (i0 > j + 1 ? i0 : j + 1); i < in; ++i) F(j, i) = F(i, j);
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
