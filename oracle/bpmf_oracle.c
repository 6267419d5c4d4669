/*
 * bpmf_oracle.c -- CPU restatement of the ExaScience/bpmf Gibbs hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bpmf_amd/ (the product) may import,
 * link or execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the reported CPU
 * baseline.
 *
 * PARITY STATUS: "parity unpinned" for the third-party dense arithmetic.
 *   The reference cannot be built here (needs Eigen3 + Random123, neither is
 *   in the image, no network), and its own tests assert no numerical value
 *   beyond "data/tiny Final Avg RMSE < 3" (data/tiny/run_test.sh:15).
 *   What IS pinned:
 *     - Philox4x32-10 against the Random123 known-answer vectors
 *       (tests/golden/philox_kat.json);
 *     - uniform->normal/gamma transforms against the *real* libstdc++
 *       std::normal_distribution / std::gamma_distribution driven by the same
 *       word stream (oracle/pin_libstdcxx.cpp, built by oracle/Makefile);
 *     - the whole chain against an independent numpy restatement
 *       (tests/golden/gen_golden.py -> tests/golden/*.npz);
 *     - the reference's only numeric assertion (tiny RMSE < 3).
 *   What is NOT pinned: Eigen's operation order inside LLT / inverse() /
 *   triangular solves / products (rounding-level differences only), and the
 *   Random123 MicroURNG word order (restated from its published header,
 *   Random123 >= 1.09 MicroURNG.hpp: rdata[3],rdata[2],rdata[1],rdata[0]).
 *
 * Every function cites the reference lines (relative to /root/reference/) it
 * follows.  Matrices are column-major like Eigen's MatrixNNd; K is runtime
 * here (compile-time BPMF_NUMLATENT in the reference, c++/bpmf.h:53).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off for the checker so
 * that x*x+y*y in the polar method is evaluated exactly as the un-fused
 * x86-64 reference build does; a -O3 -march=native -fopenmp build of the same
 * file is the timed CPU baseline).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Random123 philox.h, pinned by KAT; c++/mvnormal.cpp:19-23) */
/* ------------------------------------------------------------------------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

ORACLE_API void bpmf_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += PHILOX_W0; k1 += PHILOX_W1; }
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/*
 * r123::MicroURNG<Philox4x32> as used at c++/mvnormal.cpp:22-23,37:
 * counter words {c,0,0,n}, key {42,0}; n = blocks produced so far; a block's
 * words are handed out last-to-first.
 */
typedef struct {
    uint32_t c0;
    uint32_t n;
    int last;
    uint32_t r[4];
} urng_t;

static void urng_reset(urng_t *u, uint32_t c) /* rng_set_pos, c++/mvnormal.cpp:34-39 */
{
    u->c0 = c; u->n = 0; u->last = 0;
}

static uint32_t urng_next(urng_t *u)
{
    if (u->last == 0) {
        const uint32_t ctr[4] = { u->c0, 0u, 0u, u->n };
        const uint32_t key[2] = { 42u, 0u };
        bpmf_oracle_philox4x32_10(ctr, key, u->r);
        u->n++;
        u->last = 4;
    }
    return u->r[--u->last];
}

/* libstdc++ generate_canonical<double,53> with a 32-bit URNG (bits/random.tcc:3348+) */
static double canonical(urng_t *u)
{
    double sum = 0.0, tmp = 1.0;
    sum += (double)urng_next(u) * tmp; tmp *= 4294967296.0;
    sum += (double)urng_next(u) * tmp; tmp *= 4294967296.0;
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

/* state of a std::normal_distribution<double> object */
typedef struct { int saved_available; double saved; } normal_t;

/* std::normal_distribution<double>::operator() (bits/random.tcc:1803+), polar method */
static double normal_draw(normal_t *nd, urng_t *u)
{
    double ret;
    if (nd->saved_available) {
        nd->saved_available = 0;
        ret = nd->saved;
    } else {
        double x, y, r2;
        do {
            x = 2.0 * canonical(u) - 1.0;
            y = 2.0 * canonical(u) - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = sqrt(-2 * log(r2) / r2);
        nd->saved = x * mult;
        nd->saved_available = 1;
        ret = y * mult;
    }
    ret = ret * 1.0 + 0.0;
    return ret;
}

/* randn(), c++/mvnormal.cpp:41-43: a *temporary* distribution per call, so the
 * saved second variate is always thrown away. */
static double randn(urng_t *u)
{
    normal_t nd = { 0, 0.0 };
    return normal_draw(&nd, u);
}

/* std::gamma_distribution<double>(alpha, 1.0)(rng) constructed fresh, as at
 * c++/mvnormal.cpp:68-69 (bits/random.tcc:2337-2390, Marsaglia-Tsang; the
 * member normal_distribution keeps its saved value between loop trips). */
static double gamma_draw(urng_t *u, double alpha)
{
    const double beta = 1.0;
    const double malpha = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double a1_init = malpha - 1.0 / 3.0;
    const double a2 = 1.0 / sqrt(9.0 * a1_init);
    normal_t nd = { 0, 0.0 };

    double uu, v, n;
    const double a1 = malpha - 1.0 / 3.0;
    do {
        do {
            n = normal_draw(&nd, u);
            v = 1.0 + a2 * n;
        } while (v <= 0.0);
        v = v * v * v;
        uu = canonical(u);
    } while (uu > 1.0 - 0.0331 * n * n * n * n
             && (log(uu) > (0.5 * n * n + a1 * (1.0 - v + log(v)))));

    if (alpha == malpha)
        return a1 * v * beta;
    do uu = canonical(u); while (uu == 0.0);
    return pow(uu, 1.0 / alpha) * a1 * v * beta;
}

/* exported stream probes (used by tests to pin the RNG layer) */
ORACLE_API void bpmf_oracle_randn_stream(uint32_t counter, int n, double *out)
{
    urng_t u; urng_reset(&u, counter);
    for (int i = 0; i < n; ++i) out[i] = randn(&u);
}

ORACLE_API void bpmf_oracle_words_stream(uint32_t counter, int n, uint32_t *out)
{
    urng_t u; urng_reset(&u, counter);
    for (int i = 0; i < n; ++i) out[i] = urng_next(&u);
}

/* alternating gamma(alpha_i) / randn draws, the pattern of WishartUnitChol */
ORACLE_API void bpmf_oracle_gamma_stream(uint32_t counter, int n, const double *alphas, double *out_gamma, double *out_randn_after)
{
    urng_t u; urng_reset(&u, counter);
    for (int i = 0; i < n; ++i) {
        out_gamma[i] = gamma_draw(&u, alphas[i]);
        out_randn_after[i] = randn(&u);
    }
}

/* ------------------------------------------------------------------------- */
/* small dense helpers (column-major, leading dimension K)                    */
/* ------------------------------------------------------------------------- */
#define AT(A, i, j) ((A)[(size_t)(j) * K + (i)])

/* lower Cholesky of the lower triangle of A (Eigen::LLT<Lower> semantics,
 * c++/sample.cpp:306, c++/mvnormal.cpp:78).  Returns 0 on success, 1+k if the
 * k-th pivot is not positive (Eigen: info()!=Success). L's strict upper part
 * is zeroed. */
static int chol_lower(int K, const double *A, double *L)
{
    memset(L, 0, sizeof(double) * K * K);
    for (int j = 0; j < K; ++j) {
        double d = AT(A, j, j);
        for (int k = 0; k < j; ++k) d -= AT(L, j, k) * AT(L, j, k);
        if (!(d > 0.0)) return 1 + j;
        d = sqrt(d);
        AT(L, j, j) = d;
        for (int i = j + 1; i < K; ++i) {
            double s = AT(A, i, j);
            for (int k = 0; k < j; ++k) s -= AT(L, i, k) * AT(L, j, k);
            AT(L, i, j) = s / d;
        }
    }
    return 0;
}

/* general inverse by LU with partial pivoting (Eigen PartialPivLU::inverse,
 * c++/mvnormal.cpp:124).  Returns 0 on success. */
static int inverse_lu(int K, const double *A, double *Ainv)
{
    double *LU = (double *)malloc(sizeof(double) * K * K);
    int *piv = (int *)malloc(sizeof(int) * K);
    memcpy(LU, A, sizeof(double) * K * K);
    for (int i = 0; i < K; ++i) piv[i] = i;
    for (int k = 0; k < K; ++k) {
        int p = k; double best = fabs(AT(LU, k, k));
        for (int i = k + 1; i < K; ++i)
            if (fabs(AT(LU, i, k)) > best) { best = fabs(AT(LU, i, k)); p = i; }
        if (best == 0.0) { free(LU); free(piv); return 1; }
        if (p != k) {
            for (int j = 0; j < K; ++j) { double t = AT(LU, k, j); AT(LU, k, j) = AT(LU, p, j); AT(LU, p, j) = t; }
            int t = piv[k]; piv[k] = piv[p]; piv[p] = t;
        }
        for (int i = k + 1; i < K; ++i) {
            AT(LU, i, k) /= AT(LU, k, k);
            const double l = AT(LU, i, k);
            for (int j = k + 1; j < K; ++j) AT(LU, i, j) -= l * AT(LU, k, j);
        }
    }
    /* solve LU x = P e_c for every unit vector */
    for (int c = 0; c < K; ++c) {
        double *x = &AT(Ainv, 0, c);
        for (int i = 0; i < K; ++i) x[i] = (piv[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < K; ++i) {
            double s = x[i];
            for (int j = 0; j < i; ++j) s -= AT(LU, i, j) * x[j];
            x[i] = s;
        }
        for (int i = K - 1; i >= 0; --i) {
            double s = x[i];
            for (int j = i + 1; j < K; ++j) s -= AT(LU, i, j) * x[j];
            x[i] = s / AT(LU, i, i);
        }
    }
    free(LU); free(piv);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Normal-Wishart hyper-parameter draw  (host side in the product as well)    */
/* ------------------------------------------------------------------------- */

/* WishartUnitChol, c++/mvnormal.cpp:64-73 (incl. the discarded nrandn, Q5) */
static void wishart_unit_chol(int K, int df, urng_t *u, double *c)
{
    memset(c, 0, sizeof(double) * K * K);
    for (int i = 0; i < K; ++i) {
        const double g = gamma_draw(u, 0.5 * (df - i));
        AT(c, i, i) = sqrt(2.0 * g);
        for (int j = 0; j < K - i - 1; ++j) (void)randn(u);       /* VectorXd r = nrandn(K-i-1): unused */
        for (int j = i + 1; j < K; ++j) AT(c, i, j) = randn(u);
    }
}

/*
 * HyperParams::sample (c++/bpmf.h:98-103) = CondNormalWishart (c++/mvnormal.cpp:116-135)
 * -> NormalWishart (:96-114) -> WishartChol (:75-92) + MvNormalChol_prec (:56-61),
 * preceded by rng_set_pos(iter) (c++/sample.cpp:349).
 * Fixed prior: mu0=0, b0=2, WI=I, df=K (c++/bpmf.h:80-96).
 * `Um` is sum/N; the reference always passes 0 -> pass NULL.
 */
ORACLE_API int bpmf_oracle_hyper_sample(int K, int N, const double *cov, const double *Um,
                                        uint32_t counter, double *mu, double *LambdaU, double *LambdaF)
{
    const double kappa = 2.0;      /* b0 */
    const int nu = K;              /* df */
    urng_t u; urng_reset(&u, counter);

    double *mu_m = (double *)calloc(K, sizeof(double));
    double *mu_c = (double *)calloc(K, sizeof(double));
    double *X = (double *)malloc(sizeof(double) * K * K);
    double *T_c = (double *)malloc(sizeof(double) * K * K);
    double *L = (double *)malloc(sizeof(double) * K * K);
    double *au = (double *)malloc(sizeof(double) * K * K);
    double *r = (double *)malloc(sizeof(double) * K);
    int rc = 0;

    for (int i = 0; i < K; ++i) {
        const double um = Um ? Um[i] : 0.0;
        mu_m[i] = 0.0 - um;                                   /* mu - Um, mu = mu0 = 0 */
        mu_c[i] = (kappa * 0.0 + N * um) / (kappa + N);
    }
    const double kappa_c = kappa + N;
    const double kappa_m = (kappa * N) / (kappa + N);
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < K; ++i)
            AT(X, i, j) = ((i == j ? 1.0 : 0.0) + N * AT(cov, i, j)) + kappa_m * (mu_m[i] * mu_m[j]);
    if (inverse_lu(K, X, T_c)) { rc = -1; goto done; }
    const int nu_c = nu + N;

    /* WishartChol(T_c, nu_c, LamU): U = au * chol(T_c).matrixU() */
    if (chol_lower(K, T_c, L)) { rc = -2; goto done; }
    wishart_unit_chol(K, nu_c, &u, au);
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < K; ++i) {
            double s = 0.0;
            if (i <= j) for (int k = i; k <= j; ++k) s += AT(au, i, k) * AT(L, j, k);   /* matrixU(k,j) = L(j,k) */
            AT(LambdaU, i, j) = s;
        }

    /* MvNormalChol_prec(kappa_c, LamU, mu_c) */
    for (int i = 0; i < K; ++i) r[i] = randn(&u);
    for (int i = K - 1; i >= 0; --i) {
        double s = r[i];
        for (int j = i + 1; j < K; ++j) s -= AT(LambdaU, i, j) * r[j];
        r[i] = s / AT(LambdaU, i, i);
    }
    {
        const double sk = sqrt(kappa_c);
        for (int i = 0; i < K; ++i) mu[i] = (r[i] / sk) + mu_c[i];
    }

    /* LambdaF = LambdaU^T(upper) * LambdaU, c++/bpmf.h:101 */
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < K; ++i) {
            double s = 0.0;
            const int m = i < j ? i : j;
            for (int k = 0; k <= m; ++k) s += AT(LambdaU, k, i) * AT(LambdaU, k, j);
            AT(LambdaF, i, j) = s;
        }
done:
    free(mu_m); free(mu_c); free(X); free(T_c); free(L); free(au); free(r);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* per-column update: Sys::sample(long idx, Sys&) + computeMuLambda           */
/* c++/sample.cpp:248-258, 263-336                                            */
/* ------------------------------------------------------------------------- */
static inline __attribute__((always_inline)) int
sample_col(const int K, int64_t idx, const int64_t *colptr, const int32_t *rowidx, const double *vals,
           double mean_rating, double alpha, const double *other_items, int iter,
           const double *Lmu /* LambdaF*mu */, const double *LambdaF,
           double *MM, double *L, double *rr /* out: the sample */, int no_covariance,
           const double *precMu, const double *precLambda /* BPMF_REDUCE build (:289-291): K / K x K per column, or NULL */)
{
    urng_t u;
    urng_reset(&u, (uint32_t)((idx + 1) * (int64_t)K * ((int64_t)iter + 1)));   /* :266, truncated to uint32 (Q3) */

    for (int i = 0; i < K; ++i) rr[i] = Lmu[i];                                   /* :285 */
    memset(MM, 0, sizeof(double) * K * K);                                        /* :286 */

    if (precMu) {                                                                 /* #ifdef BPMF_REDUCE, :289-291 */
        for (int i = 0; i < K; ++i) rr[i] += precMu[(size_t)idx * K + i];
        const double *PL = precLambda + (size_t)idx * K * K;
        for (int j = 0; j < K; ++j)
            for (int i = 0; i < K; ++i) AT(MM, i, j) += AT(PL, i, j);
    } else
    for (int64_t p = colptr[idx]; p < colptr[idx + 1]; ++p) {                     /* :251, ascending row */
        const double *col = other_items + (size_t)rowidx[p] * K;                  /* :254 */
        const double w = (vals[p] - mean_rating) * alpha;                         /* :256 */
        for (int j = 0; j < K; ++j) {
            const double cj = col[j];
            for (int i = 0; i <= j; ++i) AT(MM, i, j) += col[i] * cj;             /* :255 upper */
        }
        for (int i = 0; i < K; ++i) rr[i] += col[i] * w;
    }
    for (int j = 0; j < K; ++j)                                                   /* :297 mirror, :298 */
        for (int i = 0; i <= j; ++i) {
            const double v = AT(LambdaF, i, j) + alpha * AT(MM, i, j);
            const double vt = AT(LambdaF, j, i) + alpha * AT(MM, i, j);
            AT(MM, i, j) = v; AT(MM, j, i) = vt;
        }

    if (no_covariance)                                                            /* BPMF_NO_COVARIANCE, :300-304: keep the diagonal */
        for (int j = 0; j < K; ++j)
            for (int i = 0; i < K; ++i)
                if (i != j) AT(MM, i, j) = 0.0;

    if (chol_lower(K, MM, L)) return 1;                                           /* :306-308 */

    for (int i = 0; i < K; ++i) {                                                 /* :321 L y = rr */
        double s = rr[i];
        for (int j = 0; j < i; ++j) s -= AT(L, i, j) * rr[j];
        rr[i] = s / AT(L, i, i);
    }
    for (int i = 0; i < K; ++i) rr[i] += randn(&u);                               /* :322 */
    for (int i = K - 1; i >= 0; --i) {                                            /* :323 L^T x = rr */
        double s = rr[i];
        for (int j = i + 1; j < K; ++j) s -= AT(L, j, i) * rr[j];
        rr[i] = s / AT(L, i, i);
    }
    return 0;
}

/* per-thread work buffers (MM | L | r | Lmu_i), kept across calls: the column loop of a half-iteration
 * is only milliseconds long on a many-core host, so nothing is allocated inside the parallel region */
static __thread double *tl_buf = NULL;
static __thread size_t tl_cap = 0;
static double *thread_buffers(int K)
{
    const size_t need = 2 * (size_t)K * K + 2 * (size_t)K;
    if (tl_cap < need) { free(tl_buf); tl_buf = (double *)malloc(sizeof(double) * need); tl_cap = tl_buf ? need : 0; }
    return tl_buf;
}

static inline __attribute__((always_inline)) int64_t
sample_side_K(const int K, int64_t from, int64_t to, const int64_t *colptr, const int32_t *rowidx,
              const double *vals, double mean_rating, double alpha, const double *other_items,
              double *items, int iter, const double *mu, const double *LambdaF,
              double *sum_out, double *prod_out, double *norm_out, int nthreads, const double *propLambda, int no_covariance,
              const double *precMu, const double *precLambda)
{
    int64_t failed = 0;
    double *Lmu = (double *)malloc(sizeof(double) * K);
    for (int i = 0; i < K; ++i) {                      /* rr = hp_LambdaF * hp.mu, :285 (same for all idx) */
        double s = 0.0;
        for (int j = 0; j < K; ++j) s += AT(LambdaF, i, j) * mu[j];
        Lmu[i] = s;
    }
#ifdef _OPENMP
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    const int nt = 1; (void)nthreads;
#endif
    const size_t per = (size_t)K * K + K + 1;
    double *part = (double *)calloc(per * nt, sizeof(double));        /* thread_vector<>, thread_vector.h:62-130 */

    /* proc_bind(spread): with OMP_PLACES=cores (bench.py's cpu_baseline leg sets it) the threads are
     * spread over the physical cores of both sockets and stay there, so a thread's slice of `items`
     * and its partial sums are first-touched on its own NUMA node */
#pragma omp parallel num_threads(nt) proc_bind(spread)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double *MM = thread_buffers(K);
        double *L = MM + (size_t)K * K;
        double *r = L + (size_t)K * K;
        double *Lmu_i = r + K;
        double *pp = part + per * tid, *ps = pp + (size_t)K * K, *pn = ps + K;
#pragma omp for schedule(guided)
        for (int64_t i = from; i < to; ++i) {                          /* c++/sample.cpp:352-372 */
            /* propagated posterior (c++/sample.cpp:272-277): the column's own Lambda replaces
             * hp.LambdaF; rr = hp_LambdaF * hp.mu keeps the GLOBAL mu (:285; propMu is loaded but
             * never used: SURVEY Q2) */
            const double *LF_i = propLambda ? propLambda + (size_t)i * K * K : LambdaF;
            const double *Lm = Lmu;
            if (propLambda) {
                for (int a = 0; a < K; ++a) {
                    double sacc = 0.0;
                    for (int b = 0; b < K; ++b) sacc += AT(LF_i, a, b) * mu[b];
                    Lmu_i[a] = sacc;
                }
                Lm = Lmu_i;
            }
            if (sample_col(K, i, colptr, rowidx, vals, mean_rating, alpha, other_items, iter, Lm, LF_i, MM, L, r, no_covariance, precMu, precLambda)) {
#pragma omp critical
                if (!failed || -(i + 1) > failed) failed = -(i + 1);
                continue;
            }
            double nn = 0.0;
            for (int b = 0; b < K; ++b) {
                for (int a = 0; a < K; ++a) AT(pp, a, b) += r[a] * r[b];
                ps[b] += r[b];
                nn += r[b] * r[b];
            }
            *pn += nn;
            memcpy(items + (size_t)i * K, r, sizeof(double) * K);      /* :324 */
        }
    }
    for (size_t q = 0; q < per; ++q) {                                  /* combine(): thread-id order */
        double s = 0.0;
        for (int t = 0; t < nt; ++t) s += part[per * t + q];
        if (q < (size_t)K * K) prod_out[q] = s;
        else if (q < (size_t)K * K + K) sum_out[q - (size_t)K * K] = s;
        else *norm_out = s;
    }
    free(part); free(Lmu);
    return failed;
}

/*
 * Sys::sample(Sys&) minus the hyper draw (c++/sample.cpp:352-384): samples columns
 * [from,to), returns the partial sums the caller turns into `cov`.  Return 0,
 * or -(idx+1) of a column whose Cholesky failed (THROWERROR, :308).
 */
ORACLE_API int64_t bpmf_oracle_sample_side(int K, int64_t from, int64_t to, const int64_t *colptr,
                                           const int32_t *rowidx, const double *vals, double mean_rating,
                                           double alpha, const double *other_items, double *items, int iter,
                                           const double *mu, const double *LambdaF, double *sum_out,
                                           double *prod_out, double *norm_out, int nthreads)
{
#define DISPATCH(KK) case KK: return sample_side_K(KK, from, to, colptr, rowidx, vals, mean_rating, alpha, other_items, items, iter, mu, LambdaF, sum_out, prod_out, norm_out, nthreads, NULL, 0, NULL, NULL)
    switch (K) {
        DISPATCH(8); DISPATCH(16); DISPATCH(32); DISPATCH(64); DISPATCH(128);
    default:
        return sample_side_K(K, from, to, colptr, rowidx, vals, mean_rating, alpha, other_items, items, iter, mu, LambdaF, sum_out, prod_out, norm_out, nthreads, NULL, 0, NULL, NULL);
    }
#undef DISPATCH
}

/*
 * One column, given by its own ratings: Sys::sample(long idx, Sys&) for global column id `idx` whose `n` ratings sit
 * in (rowidx, vals) with row ids into `other_items` -- for spot checks of matrices too big to hand over whole
 * (BASELINE configs[3]: 10M x 1M; the rows a column reads are gathered into a small `other_items` and renumbered).
 * out[K] receives the sample; returns 0 or 1 (Cholesky failed).
 */
ORACLE_API int bpmf_oracle_sample_column(int K, int64_t idx, int64_t n, const int32_t *rowidx, const double *vals,
                                         double mean_rating, double alpha, const double *other_items, int iter,
                                         const double *mu, const double *LambdaF, double *out)
{
    double *Lmu = (double *)malloc(sizeof(double) * K);
    double *MM = (double *)malloc(sizeof(double) * K * K), *L = (double *)malloc(sizeof(double) * K * K);
    int64_t *cp = (int64_t *)malloc(sizeof(int64_t) * 2);
    for (int i = 0; i < K; ++i) {                      /* rr = hp_LambdaF * hp.mu, :285 */
        double s = 0.0;
        for (int j = 0; j < K; ++j) s += AT(LambdaF, i, j) * mu[j];
        Lmu[i] = s;
    }
    cp[0] = 0; cp[1] = n;
    /* sample_col reads colptr[idx], colptr[idx + 1]: hand it a view whose element `idx` is cp[0] */
    const int rc = sample_col(K, idx, cp - idx, rowidx, vals, mean_rating, alpha, other_items, iter, Lmu, LambdaF, MM, L, out, 0, NULL, NULL);
    free(Lmu); free(MM); free(L); free(cp);
    return rc;
}

/* the same with propagated-posterior priors (-m / -l, c++/sample.cpp:152-174,272-277):
 * propLambda holds one column-major K x K matrix per column of this side (global column index) */
ORACLE_API int64_t bpmf_oracle_sample_side_prop(int K, int64_t from, int64_t to, const int64_t *colptr,
                                                const int32_t *rowidx, const double *vals, double mean_rating,
                                                double alpha, const double *other_items, double *items, int iter,
                                                const double *mu, const double *LambdaF, const double *propLambda,
                                                double *sum_out, double *prod_out, double *norm_out, int nthreads)
{
    return sample_side_K(K, from, to, colptr, rowidx, vals, mean_rating, alpha, other_items, items, iter, mu, LambdaF, sum_out,
                         prod_out, norm_out, nthreads, propLambda, 0, NULL, NULL);
}

/* the BPMF_NO_COVARIANCE build of the reference (c++/sample.cpp:300-304): only the diagonal of
 * Lambda* is kept before the factorisation */
ORACLE_API int64_t bpmf_oracle_sample_side_nocov(int K, int64_t from, int64_t to, const int64_t *colptr,
                                                 const int32_t *rowidx, const double *vals, double mean_rating,
                                                 double alpha, const double *other_items, double *items, int iter,
                                                 const double *mu, const double *LambdaF,
                                                 double *sum_out, double *prod_out, double *norm_out, int nthreads)
{
    return sample_side_K(K, from, to, colptr, rowidx, vals, mean_rating, alpha, other_items, items, iter, mu, LambdaF, sum_out,
                         prod_out, norm_out, nthreads, NULL, 1, NULL, NULL);
}


/*
 * BPMF_REDUCE build of the reference.  Sys::preComputeMuLambda (c++/sample.cpp:234-246) with computeMuLambda's
 * local_only filter (:248-258): for EVERY column i of this Sys (colptr / rowidx / vals: its whole matrix), mu and the
 * UPPER triangle of Lambda accumulated from zero over the ratings whose row lies in [other_from, other_to) -- the rows
 * this rank owns of the other side.  precMu: K per column; precLambda: K x K column-major per column (lower part zero).
 */
ORACLE_API void bpmf_oracle_precompute(int K, int64_t n, const int64_t *colptr, const int32_t *rowidx, const double *vals,
                                       double mean_rating, double alpha, const double *other_items,
                                       int64_t other_from, int64_t other_to, double *precMu, double *precLambda)
{
    for (int64_t idx = 0; idx < n; ++idx) {
        double *mu = precMu + (size_t)idx * K, *MM = precLambda + (size_t)idx * K * K;
        memset(mu, 0, sizeof(double) * K);                                            /* :240-241 */
        memset(MM, 0, sizeof(double) * K * K);
        for (int64_t p = colptr[idx]; p < colptr[idx + 1]; ++p) {
            if (rowidx[p] < other_from || rowidx[p] >= other_to) continue;            /* :253 */
            const double *col = other_items + (size_t)rowidx[p] * K;
            const double w = (vals[p] - mean_rating) * alpha;
            for (int j = 0; j < K; ++j) {
                const double cj = col[j];
                for (int i = 0; i <= j; ++i) AT(MM, i, j) += col[i] * cj;             /* :255 upper */
            }
            for (int i = 0; i < K; ++i) mu[i] += col[i] * w;                          /* :256 */
        }
    }
}

/* Sys::sample(Sys&) of the BPMF_REDUCE build: columns [from, to) from rr = LambdaF mu + precMu.col(idx), MM = precLambda
 * (c++/sample.cpp:285-291); the ratings are not read */
ORACLE_API int64_t bpmf_oracle_sample_side_prec(int K, int64_t from, int64_t to, double alpha, const double *precMu,
                                                const double *precLambda, double *items, int iter, const double *mu,
                                                const double *LambdaF, double *sum_out, double *prod_out, double *norm_out,
                                                int nthreads, int no_covariance)
{
    return sample_side_K(K, from, to, NULL, NULL, NULL, 0.0, alpha, NULL, items, iter, mu, LambdaF, sum_out, prod_out, norm_out,
                         nthreads, NULL, no_covariance, precMu, precLambda);
}

/* cov = (prod - sum sum^T / N) / (N-1), c++/sample.cpp:383-384 */
ORACLE_API void bpmf_oracle_cov(int K, int64_t N, const double *sum, const double *prod, double *cov)
{
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < K; ++i)
            AT(cov, i, j) = (AT(prod, i, j) - (sum[i] * sum[j] / (double)N)) / (double)(N - 1);
}

/*
 * Sys::predict, c++/sample.cpp:48-96.  T is CSC with one column per item of
 * `items`; Pavg/Pm2 are stored in T's nnz order.  n = iter<burnin ? 0 : iter-burnin (Q6).
 */
ORACLE_API void bpmf_oracle_predict(int K, int64_t from, int64_t to, const int64_t *tcolptr, const int32_t *trowidx,
                                    const double *tvals, const double *items, const double *other_items,
                                    double mean_rating, int n, double *Pavg, double *Pm2,
                                    double *se_out, double *se_avg_out, int64_t *nump_out, int nthreads)
{
    double se = 0.0, se_avg = 0.0;
    int64_t nump = 0;
#ifdef _OPENMP
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    (void)nthreads;
#endif
#pragma omp parallel for reduction(+ : se, se_avg, nump) num_threads(nt)
    for (int64_t k = from; k < to; ++k) {
        for (int64_t p = tcolptr[k]; p < tcolptr[k + 1]; ++p) {
            const double *m = items + (size_t)k * K;
            const double *uo = other_items + (size_t)trowidx[p] * K;
            double dot = 0.0;
            for (int i = 0; i < K; ++i) dot += m[i] * uo[i];
            const double pred = dot + mean_rating;                      /* :78 */
            se += (tvals[p] - pred) * (tvals[p] - pred);
            double avg = Pavg[p];
            const double delta = pred - avg;
            avg = (n == 0) ? pred : (avg + delta / n);                   /* :84 */
            Pavg[p] = avg;
            Pm2[p] = (n == 0) ? 0 : Pm2[p] + delta * (pred - avg);       /* :86 */
            se_avg += (tvals[p] - avg) * (tvals[p] - avg);
            nump++;
        }
    }
    *se_out = se; *se_avg_out = se_avg; *nump_out = nump;
}

static double tick(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/*
 * The Gibbs loop of main(), c++/bpmf.cpp:180-210 + the trailing extra predict
 * (:242, Q6), NO_COMM.  M_* is the train matrix with one column per movie,
 * Mt_* its transpose (one column per user); T_* / Tt_* the same for the test
 * matrix.  U is K x nusers, V is K x nmovies, both zero-initialised by the
 * caller (Sys::init, c++/sample.cpp:185).  Per-iteration outputs (arrays of
 * nsims): rmse, rmse_avg (movies.print), norm_u, norm_m (= sqrt(users.norm),
 * sqrt(movies.norm)), secs (start..stop of :182-193).  final[0] = "Final Avg
 * RMSE", final[1] = num_predict.
 * If `trace` is non-NULL it receives, per iteration, [mu_m(K) LambdaF_m(K*K)
 * mu_u(K) LambdaF_u(K*K)].
 */
ORACLE_API int64_t bpmf_oracle_gibbs(int K, int64_t nusers, int64_t nmovies,
                                     const int64_t *M_colptr, const int32_t *M_rowidx, const double *M_vals,
                                     const int64_t *Mt_colptr, const int32_t *Mt_rowidx, const double *Mt_vals,
                                     const int64_t *T_colptr, const int32_t *T_rowidx, const double *T_vals,
                                     const int64_t *Tt_colptr, const int32_t *Tt_rowidx, const double *Tt_vals,
                                     double alpha, int nsims, int burnin, int nthreads,
                                     double *U, double *V, double *Pavg, double *Pm2,
                                     double *rmse, double *rmse_avg, double *norm_u, double *norm_m, double *secs,
                                     double *final, double *trace)
{
    const int64_t nnz = M_colptr[nmovies], nnzt = T_colptr[nmovies];
    /* mean_rating = M.sum()/M.nonZeros(), per Sys (c++/sample.cpp:183, Q10) */
    double s_m = 0.0, s_u = 0.0;
    for (int64_t p = 0; p < nnz; ++p) { s_m += M_vals[p]; s_u += Mt_vals[p]; }
    const double mean_m = s_m / (double)nnz, mean_u = s_u / (double)nnz;

    const size_t KK = (size_t)K * K;
    double *cov_m = (double *)calloc(KK, sizeof(double)), *cov_u = (double *)calloc(KK, sizeof(double));
    double *mu = (double *)malloc(sizeof(double) * K), *LU = (double *)malloc(sizeof(double) * KK), *LF = (double *)malloc(sizeof(double) * KK);
    double *sum = (double *)malloc(sizeof(double) * K), *prod = (double *)malloc(sizeof(double) * KK);
    double *Pavg_u = (double *)malloc(sizeof(double) * (nnzt ? nnzt : 1)), *Pm2_u = (double *)malloc(sizeof(double) * (nnzt ? nnzt : 1));
    /* Pm2 = Pavg = T (c++/sample.cpp:123,134) */
    for (int64_t p = 0; p < nnzt; ++p) { Pavg[p] = Pm2[p] = T_vals[p]; Pavg_u[p] = Pm2_u[p] = Tt_vals[p]; }
    double nrm_m = 0.0, nrm_u = 0.0, se, se_avg; int64_t nump = 0, rc = 0;
    double last_rmse_avg = 0.0;

    for (int it = 0; it < nsims; ++it) {
        const double start = tick();
        /* movies.sample(users): iter++ -> iter == it */
        if (bpmf_oracle_hyper_sample(K, (int)nmovies, cov_m, NULL, (uint32_t)it, mu, LU, LF)) { rc = -1; break; }
        if (trace) { memcpy(trace, mu, sizeof(double) * K); memcpy(trace + K, LF, sizeof(double) * KK); trace += K + KK; }
        rc = bpmf_oracle_sample_side(K, 0, nmovies, M_colptr, M_rowidx, M_vals, mean_m, alpha, U, V, it, mu, LF, sum, prod, &nrm_m, nthreads);
        if (rc) break;
        bpmf_oracle_cov(K, nmovies, sum, prod, cov_m);
        /* users.sample(movies) */
        if (bpmf_oracle_hyper_sample(K, (int)nusers, cov_u, NULL, (uint32_t)it, mu, LU, LF)) { rc = -1; break; }
        if (trace) { memcpy(trace, mu, sizeof(double) * K); memcpy(trace + K, LF, sizeof(double) * KK); trace += K + KK; }
        rc = bpmf_oracle_sample_side(K, 0, nusers, Mt_colptr, Mt_rowidx, Mt_vals, mean_u, alpha, V, U, it, mu, LF, sum, prod, &nrm_u, nthreads);
        if (rc) break;
        bpmf_oracle_cov(K, nusers, sum, prod, cov_u);
        /* eval */
        const int n = (it < burnin) ? 0 : (it - burnin);
        bpmf_oracle_predict(K, 0, nmovies, T_colptr, T_rowidx, T_vals, V, U, mean_m, n, Pavg, Pm2, &se, &se_avg, &nump, nthreads);
        rmse[it] = sqrt(se / (double)nump); rmse_avg[it] = sqrt(se_avg / (double)nump);
        {
            double se2, sea2; int64_t np2;
            bpmf_oracle_predict(K, 0, nusers, Tt_colptr, Tt_rowidx, Tt_vals, U, V, mean_u, n, Pavg_u, Pm2_u, &se2, &sea2, &np2, nthreads);
        }
        secs[it] = tick() - start;
        norm_u[it] = sqrt(nrm_u); norm_m[it] = sqrt(nrm_m);
        last_rmse_avg = rmse_avg[it];
    }
    if (!rc && nsims > 0) {
        /* movies.predict(users, true) once more with the same iter (c++/bpmf.cpp:225,242) */
        const int it = nsims - 1;
        const int n = (it < burnin) ? 0 : (it - burnin);
        bpmf_oracle_predict(K, 0, nmovies, T_colptr, T_rowidx, T_vals, V, U, mean_m, n, Pavg, Pm2, &se, &se_avg, &nump, nthreads);
        last_rmse_avg = sqrt(se_avg / (double)nump);
    }
    final[0] = last_rmse_avg; final[1] = (double)nump;
    free(cov_m); free(cov_u); free(mu); free(LU); free(LF); free(sum); free(prod); free(Pavg_u); free(Pm2_u);
    return rc;
}
