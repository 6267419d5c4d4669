// kernels.h -- hand-written HIP kernels for gfx950 (CDNA4, wave64) behind the C ABI.
//
// The path (c++/sample.cpp:248-336 + c++/mvnormal.cpp:18-47 of the reference),
// re-designed for MI355X:
//
//   k_gram<K>        one wavefront per (column, nnz-chunk) work item.  The K-vectors
//                    of the rated rows are gathered straight into MFMA operand
//                    layout (16 lanes x 8 B = one 128-B line per 16 latent dims per
//                    rating, 4 ratings per instruction) and the upper-triangular
//                    16x16 tiles of sum_j u_j u_j^T are accumulated with
//                    v_mfma_f64_16x16x4_f64; the K-vector sum_j w_j u_j rides along
//                    on the VALU.  Columns that fit one chunk are finished in the
//                    same wave; chunks of heavy columns write their partial tiles.
//   k_finish_multi<K> sums the partial tiles of a heavy column in chunk order and
//                    finishes it.
//   finish_column<K> Lambda* = LambdaF + alpha*G into LDS, one row per lane into
//                    registers, right-looking Cholesky with the pivot column
//                    broadcast through LDS, fused forward solve, Philox/polar
//                    normal draw, backward solve, coalesced 8*K-byte store.
//   k_colstats<K>    sum x, sum x x^T of the fresh columns (again an MFMA Gram),
//                    reduced in a fixed order so results are run-to-run identical.
//   k_predict<K>     test-set dot products, running mean / M2, squared errors.
//
// Everything is fp64 like the reference (c++/bpmf.h:55-58).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.h"

namespace bpmf {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int K>
struct Geo {
    static constexpr int NT = (K + 15) / 16;             // 16-wide tiles per dimension (K=8 is zero-padded)
    static constexpr int NTRI = NT * (NT + 1) / 2;        // upper-triangular tiles incl. diagonal
    static constexpr int LD = K + 1;                      // LDS leading dimension in doubles (odd: column walks hit distinct banks)
    static constexpr int PART = NTRI * 256 + NT * 16;     // doubles in one partial: tiles in accumulator layout + rhs
    // waves per SIMD the sampler is compiled for (bounds the VGPR budget: 512 / WPS):
    // a lane keeps one K-double row of Lambda* in registers during the factorisation
    static constexpr int WPS = K <= 32 ? 4 : 2;
};

// v_mfma_f64_16x16x4_f64 operand / result layout (lane l, kq = l>>4, li = l&15):
//   A[i=li][k=kq], B[k=kq][j=li]  one double each;  D[i = kq + 4*reg][j = li], reg 0..3.
__device__ __forceinline__ d4 mfma16(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// value of `v` in lane `src` (wave-uniform src) through v_readlane_b32: no LDS traffic
__device__ __forceinline__ double bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

struct SampleArgs {
    // ratings of this rank's columns
    const int32_t *rowidx;
    const double *vals;
    // schedule
    const int32_t *wi_col;      // local column of work item
    const int64_t *wi_p0;       // first nnz of the chunk
    const int32_t *wi_len;      // nnz in the chunk
    const int32_t *wi_slot;     // partial slot, or -1: single-chunk column, finish in place
    const int32_t *mc_col;      // heavy columns: local column, first slot, number of chunks
    const int32_t *mc_slot0;
    const int32_t *mc_nchunks;
    double *partials;
    // factors
    const double *other_items;  // K x nrows
    double *items;              // K x ncols
    int64_t col_from;           // global id of local column 0
    // per-call
    const double *LambdaF;      // K x K col-major (device)
    const double *Lmu;          // LambdaF * mu (device)
    unsigned long long *fail;   // min global column id whose factorisation failed
    double mean_rating;
    double alpha;
    uint32_t iter_plus_1;
};

// ---------------------------------------------------------------------------
// K normals of the reference's per-column stream, in stream order.
// The reference draws them one after another with the polar method; every
// attempt eats exactly one Philox block, so attempt n <-> block n, and the
// j-th normal is the j-th ACCEPTED attempt.  64 lanes try blocks base..base+63
// at once, a ballot ranks the accepted ones.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double polar_r2(double x, double y)
{
#pragma clang fp contract(off)
    return x * x + y * y;      // two roundings + add, as the un-fused x86 reference build evaluates it
}

template <int NMAX>
__device__ __forceinline__ void draw_normals(uint32_t counter, int n, double *out_lds, int lane)
{
    int produced = 0;
    uint32_t base = 0;
    while (produced < n) {                                         // wave-uniform
        const Philox4 b = stream_block(counter, base + (uint32_t)lane);
        const double x = 2.0 * canonical53(b.w[3], b.w[2]) - 1.0;   // URNG order: w3, w2, w1, w0
        const double y = 2.0 * canonical53(b.w[1], b.w[0]) - 1.0;
        const double r2 = polar_r2(x, y);
        const bool acc = !(r2 > 1.0 || r2 == 0.0);
        const unsigned long long m = __ballot(acc);
        const int rank = produced + __popcll(m & ((1ull << lane) - 1ull));
        if (acc && rank < n) {
            const double mult = sqrt(-2 * log(r2) / r2);
            out_lds[rank] = y * mult;
        }
        produced += __popcll(m);
        base += 64u;
    }
}

// ---------------------------------------------------------------------------
// Gram accumulation over one chunk of a column's ratings.
// ---------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void gram_chunk(const int32_t *__restrict__ rowidx, const double *__restrict__ vals, int len,
                                           const double *__restrict__ other, double mean, double alpha,
                                           d4 (&acc)[Geo<K>::NTRI], double (&r)[Geo<K>::NT], int lane)
{
    constexpr int NT = Geo<K>::NT;
    const int kq = lane >> 4, li = lane & 15;
    constexpr int UN = 4;                                          // 4 MFMA k-steps (16 ratings) per trip
    for (int base = 0; base < len; base += 4 * UN) {
        double y[UN][NT], w[UN];
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            const int q = base + s * 4 + kq;
            const bool ok = q < len;
            const int row = ok ? rowidx[q] : 0;
            w[s] = ok ? (vals[q] - mean) * alpha : 0.0;            // c++/sample.cpp:256
            const double *col = other + (size_t)row * K + li;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                y[s][t] = (ok && (t * 16 + li < K)) ? col[t * 16] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < UN; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t) r[t] = fma(y[s][t], w[s], r[t]);
            int tri = 0;
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = I; J < NT; ++J, ++tri) acc[tri] = mfma16(y[s][I], y[s][J], acc[tri]);
        }
    }
    // the 4 k-groups of lanes hold partial rhs sums for the same latent index
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        r[t] += __shfl_xor(r[t], 16);
        r[t] += __shfl_xor(r[t], 32);
    }
}

// ---------------------------------------------------------------------------
// Everything after the Gram for one column (c++/sample.cpp:285,297-324).
// lds: K*LD + 2*K + 64 doubles.
//
// The wave holds Lambda* in registers, S = 64/K lanes per row: lane (h, i) =
// (lane / K, lane % K) owns the entries (i, j) with j = m*S + h, m = 0..M-1
// (M = K*K/64: 16 doubles at K=32).  Right-looking Cholesky: at step k the
// pivot comes from its owner lane through v_readlane, the owners scale column k
// and publish it to LDS (row-major L, reused by the backward solve), then every
// lane updates its own entries with L(i,k) * L(j,k), the second factor being a
// broadcast LDS read.  The forward solve L y = b is one more fused column.
// ---------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void finish_column(const SampleArgs &a, int col_local, const d4 (&acc)[Geo<K>::NTRI],
                                              const double (&r)[Geo<K>::NT], double *lds, int lane)
{
    constexpr int NT = Geo<K>::NT, LD = Geo<K>::LD;
    constexpr int S = 64 / K, M = K / S;
    static_assert(K * S == 64 && M * S == K, "K must be a power of two <= 64");
    const int kq = lane >> 4, li = lane & 15;
    double *sA = lds, *sb = lds + K * LD, *sz = sb + K, *sdummy = sz + K;
    const int64_t idx = a.col_from + col_local;

    // z ~ N(0, I) from stream (idx+1)*K*(iter+1) truncated to 32 bits (c++/sample.cpp:266, c++/bpmf.h:67)
    const uint32_t counter = (uint32_t)((uint64_t)(idx + 1) * (uint64_t)K * (uint64_t)a.iter_plus_1);
    draw_normals<K>(counter, K, sz, lane);

    // G (upper tiles, accumulator layout) -> LDS, mirrored (c++/sample.cpp:297); rhs partial sums -> LDS
    {
        int tri = 0;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = I; J < NT; ++J, ++tri)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = I * 16 + kq + 4 * reg, j = J * 16 + li;
                    if (i < K && j < K) {
                        sA[i * LD + j] = acc[tri][reg];
                        if (I != J) sA[j * LD + i] = acc[tri][reg];
                    }
                }
        if (kq == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t * 16 + li < K) sb[t * 16 + li] = r[t];
        }
    }
    __syncthreads();

    const int h = lane / K, i = lane % K;
    // Lambda* = LambdaF + alpha * G (:298); b = LambdaF*mu + rr (:285,:256)
    double row[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = m * S + h;
        row[m] = fma(a.alpha, sA[i * LD + j], a.LambdaF[i + j * K]);
    }
    double bi = a.Lmu[i] + sb[i];
    const double zi = sz[i];
    double my_dinv = 1.0, dmin = 1.0;
    __syncthreads();

#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int hk = k % S, mk = k / S;
        const double d = bcast(row[mk], hk * K + k);
        dmin = fmin(dmin, d);                                     // Eigen LLT: pivot <= 0 -> info() != Success (:308)
        const double dinv = 1.0 / sqrt(d);
        // owners publish column k (row k: sqrt(d); rows i>k: L(i,k)); the other lanes hit a dummy
        // slot so that the step stays branch-free (branches let LLVM sink whole FMA chains)
        double *dst = (h == hk) ? &sA[i * LD + k] : &sdummy[lane];
        *dst = row[mk] * dinv;
        my_dinv = (i == k) ? dinv : my_dinv;
        __syncthreads();
        const double lik = sA[i * LD + k];
        // fused forward solve (:321): y_k = b_k / L(k,k); b_i -= L(i,k) y_k for i>k
        const double yk = bcast(bi, k) * dinv;
        bi = (i == k) ? yk : ((i > k) ? fma(-lik, yk, bi) : bi);
        // trailing update of this lane's entries j = m*S+h > k
        {
            const double u = fma(-lik, sA[(mk * S + h) * LD + k], row[mk]);
            row[mk] = (h > hk) ? u : row[mk];
        }
#pragma unroll
        for (int m = mk + 1; m < M; ++m) row[m] = fma(-lik, sA[(m * S + h) * LD + k], row[m]);
        // Pin this step's results: otherwise instruction selection defers every FMA chain to
        // the step that finally needs row[m] and keeps (spills) all the L(j,k) it loaded meanwhile.
#pragma unroll
        for (int m = mk; m < M; ++m) asm volatile("" : "+v"(row[m]));
    }

    bi += zi;                                                     // rr += nrandn(K)  (:322)

    // backward solve L^T x = rr (:323): x_k = rr_k / L(k,k), then rr_i -= L(k,i) x_k for i<k
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
        const double xk = bcast(bi * my_dinv, k);
        const double lki = (i < k) ? sA[k * LD + i] : 0.0;
        bi = (i == k) ? xk : fma(-lki, xk, bi);
    }

    if (h == 0) a.items[(size_t)idx * K + i] = bi;                // items().col(idx) = rr (:324)
    const bool bad = !(dmin > 0.0) || !(fabs(bi) <= 1.79769313486231570815e+308);
    if (__any(bad) && lane == 0) atomicMin(a.fail, (unsigned long long)idx);
}

template <int K>
__global__ __launch_bounds__(64, Geo<K>::WPS) void k_gram(SampleArgs a)
{
    constexpr int NT = Geo<K>::NT, NTRI = Geo<K>::NTRI, LD = Geo<K>::LD, PART = Geo<K>::PART;
    __shared__ double lds[K * LD + 2 * K + 64];
    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    const int col = a.wi_col[w];
    const int64_t p0 = a.wi_p0[w];
    const int len = a.wi_len[w];
    const int slot = a.wi_slot[w];

    d4 acc[NTRI];
    double r[NT];
#pragma unroll
    for (int t = 0; t < NTRI; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NT; ++t) r[t] = 0.0;

    gram_chunk<K>(a.rowidx + p0, a.vals + p0, len, a.other_items, a.mean_rating, a.alpha, acc, r, lane);

    if (slot >= 0) {                                               // chunk of a heavy column: park the partial
        double *p = a.partials + (size_t)slot * PART;
#pragma unroll
        for (int t = 0; t < NTRI; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) p[(t * 4 + reg) * 64 + lane] = acc[t][reg];
        if (lane < 16) {
#pragma unroll
            for (int t = 0; t < NT; ++t) p[NTRI * 256 + t * 16 + lane] = r[t];
        }
        return;
    }
    finish_column<K>(a, col, acc, r, lds, lane);
}

template <int K>
__global__ __launch_bounds__(64, Geo<K>::WPS) void k_finish_multi(SampleArgs a)
{
    constexpr int NT = Geo<K>::NT, NTRI = Geo<K>::NTRI, LD = Geo<K>::LD, PART = Geo<K>::PART;
    __shared__ double lds[K * LD + 2 * K + 64];
    const int lane = threadIdx.x;
    const int m = blockIdx.x;
    const int col = a.mc_col[m];
    const int nch = a.mc_nchunks[m];
    const double *p = a.partials + (size_t)a.mc_slot0[m] * PART;

    d4 acc[NTRI];
    double r[NT];
#pragma unroll
    for (int t = 0; t < NTRI; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NT; ++t) r[t] = 0.0;
    for (int c = 0; c < nch; ++c, p += PART) {                     // fixed chunk order: deterministic
#pragma unroll
        for (int t = 0; t < NTRI; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[t][reg] += p[(t * 4 + reg) * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t) r[t] += p[NTRI * 256 + t * 16 + (lane & 15)];
    }
    finish_column<K>(a, col, acc, r, lds, lane);
}

// ---------------------------------------------------------------------------
// sum x, sum x x^T over the columns [c0, c1) of `items` (thread_vector reducers,
// c++/sample.cpp:345-347,359-362,379-381).  Wave w takes a contiguous slice and
// writes a partial in accumulator layout; k_colstats_final adds the partials
// in wave order and unpacks to column-major prod | sum | norm.
// ---------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(64) void k_colstats(const double *__restrict__ items, int64_t c0, int64_t c1, int nwaves,
                                                 double *__restrict__ partials)
{
    constexpr int NT = Geo<K>::NT, NTRI = Geo<K>::NTRI, PART = Geo<K>::PART;
    const int lane = threadIdx.x, kq = lane >> 4, li = lane & 15;
    const int w = blockIdx.x;
    const int64_t n = c1 - c0;
    const int64_t per = (((n + nwaves - 1) / nwaves) + 3) & ~(int64_t)3;
    const int64_t b = c0 + w * per;
    const int64_t e = (b + per < c1) ? b + per : c1;

    d4 acc[NTRI];
    double r[NT];
#pragma unroll
    for (int t = 0; t < NTRI; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NT; ++t) r[t] = 0.0;

    for (int64_t c = b; c < e; c += 8) {
        double y[2][NT];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t col = c + s * 4 + kq;
            const bool ok = col < e;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                y[s][t] = (ok && (t * 16 + li < K)) ? items[(size_t)col * K + t * 16 + li] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t) r[t] += y[s][t];
            int tri = 0;
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = I; J < NT; ++J, ++tri) acc[tri] = mfma16(y[s][I], y[s][J], acc[tri]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        r[t] += __shfl_xor(r[t], 16);
        r[t] += __shfl_xor(r[t], 32);
    }
    double *p = partials + (size_t)w * PART;
#pragma unroll
    for (int t = 0; t < NTRI; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) p[(t * 4 + reg) * 64 + lane] = acc[t][reg];
    if (lane < 16) {
#pragma unroll
        for (int t = 0; t < NT; ++t) p[NTRI * 256 + t * 16 + lane] = r[t];
    }
}

// out: prod[K*K] col-major | sum[K] | norm
template <int K>
__global__ __launch_bounds__(256) void k_colstats_final(const double *__restrict__ partials, int nwaves, double *__restrict__ out)
{
    constexpr int NT = Geo<K>::NT, NTRI = Geo<K>::NTRI, PART = Geo<K>::PART;
    __shared__ double diag[K];
    for (int e = threadIdx.x; e < K * K + K; e += blockDim.x) {
        int off;
        if (e < K * K) {
            int i = e % K, j = e / K;
            if (i > j) { const int t = i; i = j; j = t; }          // symmetric: read the upper tile
            const int I = i >> 4, J = j >> 4;
            const int tri = I * NT - (I * (I - 1)) / 2 + (J - I);
            const int ii = i & 15, jj = j & 15;
            off = (tri * 4 + (ii >> 2)) * 64 + (ii & 3) * 16 + jj;
        } else {
            off = NTRI * 256 + (e - K * K);
        }
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += partials[(size_t)w * PART + off];
        out[e] = s;
        if (e < K * K && (e % K) == (e / K)) diag[e % K] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double nn = 0.0;
        for (int i = 0; i < K; ++i) nn += diag[i];
        out[K * K + K] = nn;
    }
}

// ---------------------------------------------------------------------------
// Sys::predict (c++/sample.cpp:48-96): 16 lanes per test rating.
// ---------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_predict(const int32_t *__restrict__ tcol, const int32_t *__restrict__ trow,
                                                 const double *__restrict__ tval, int64_t nnz, int64_t per_wave,
                                                 const double *__restrict__ items, const double *__restrict__ other,
                                                 int64_t col_from, double mean, int n, double *__restrict__ pavg,
                                                 double *__restrict__ pm2, double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, kq = lane >> 4, li = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t b = wave * per_wave;
    const int64_t e = (b + per_wave < nnz) ? b + per_wave : nnz;
    double se = 0.0, se_avg = 0.0;
    for (int64_t q0 = b; q0 < e; q0 += 4) {
        const int64_t q = q0 + kq;
        const bool ok = q < e;
        double dot = 0.0;
        if (ok) {
            const double *m = items + (size_t)(col_from + tcol[q]) * K;
            const double *u = other + (size_t)trow[q] * K;
#pragma unroll
            for (int t = 0; t < (K + 15) / 16; ++t)
                if (t * 16 + li < K) dot = fma(m[t * 16 + li], u[t * 16 + li], dot);
        }
        dot += __shfl_xor(dot, 8);
        dot += __shfl_xor(dot, 4);
        dot += __shfl_xor(dot, 2);
        dot += __shfl_xor(dot, 1);
        if (ok && li == 0) {
            const double pred = dot + mean;                         // :78
            const double v = tval[q];
            se += (v - pred) * (v - pred);
            double avg = pavg[q];
            const double delta = pred - avg;
            avg = (n == 0) ? pred : (avg + delta / n);              // :84 (n, not n+1: reference quirk)
            pavg[q] = avg;
            pm2[q] = (n == 0) ? 0.0 : pm2[q] + delta * (pred - avg);   // :86
            se_avg += (v - avg) * (v - avg);
        }
    }
    // lanes 0,16,32,48 hold this wave's partial sums
    se += __shfl_xor(se, 16);      se += __shfl_xor(se, 32);
    se_avg += __shfl_xor(se_avg, 16); se_avg += __shfl_xor(se_avg, 32);
    if (lane == 0) { partial[2 * wave] = se; partial[2 * wave + 1] = se_avg; }
}

__global__ void k_predict_final(const double *__restrict__ partial, int64_t nwaves, double *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double se = 0.0, sa = 0.0;
        for (int64_t w = 0; w < nwaves; ++w) { se += partial[2 * w]; sa += partial[2 * w + 1]; }
        out[0] = se; out[1] = sa;
    }
}

// test probe: the first n normals of stream `counter`
__global__ __launch_bounds__(64) void k_randn_probe(uint32_t counter, int n, double *out)
{
    __shared__ double z[128];
    draw_normals<128>(counter, n, z, threadIdx.x);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) out[i] = z[i];
}

}  // namespace bpmf
