// fp32 large-K form of the hot path (K = 128): BASELINE config "MovieLens-1M, K=128, fp32" -- the pieces around the
// sampler (kernels_wg2.h): tile geometry and MFMA traits, column statistics, prediction.
//
// The reference computes in fp64 throughout (c++/bpmf.h:55-58); this path keeps the factors,
// the Gram, the factorisation and the solves in fp32 and everything that leaves the column loop
// (hyper-parameters, column statistics, prediction sums, the normal draws) in fp64.  It is the
// "mixed-precision tolerance study" of the north star: tests/test_gpu_f32.py states what the
// fp32 arithmetic costs against the fp64 restatement of the reference.
// Operand / result layout of v_mfma_f32_16x16x4_f32 (tools/probes/layout16f32_probe.hip):
//   A lane 16 k + i, B lane 16 k + j (one float each);  D[i = 4 (lane / 16) + reg][j = lane % 16].
#pragma once
#include "kernels.h"

namespace bpmf {

typedef float f4 __attribute__((ext_vector_type(4)));

// element type of the factors / tiles: float (K = 128) or double (K = 64: the reference's arithmetic)
template <typename T> struct WgTraits;
template <> struct WgTraits<float> {
    typedef f4 acc_t;
    __device__ static __forceinline__ f4 mfma(float x, float y, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0); }
    __device__ static __forceinline__ int drow(int kq, int reg) { return 4 * kq + reg; }      // D[i = 4 (lane / 16) + reg][j = lane % 16]
    __device__ static __forceinline__ float rsqrt_acc(float d) { return __frsqrt_rn(d); }                // v_rsq_f32 (1 ulp)
    __device__ static __forceinline__ float bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
};
template <> struct WgTraits<double> {
    typedef d4 acc_t;
    __device__ static __forceinline__ d4 mfma(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }
    __device__ static __forceinline__ int drow(int kq, int reg) { return kq + 4 * reg; }      // D[i = lane / 16 + 4 reg][j = lane % 16]
    __device__ static __forceinline__ double rsqrt_acc(double d) { return 1.0 / sqrt(d); }
    __device__ static __forceinline__ double bcast(double v, int l) { return ::bpmf::bcast(v, l); }
};


template <int K>
struct GeoF {
    static constexpr int NT = K / 16;                    // 16-wide tiles per dimension
    static constexpr int NTRI = NT * (NT + 1) / 2;
    // R (upper, Lambda* = R^T R) lives in LDS by block rows: block row s is 16 x (K - 16 s) floats
    // with a row stride of K - 16 s + 4: 40 960 B of LDS per K = 128 fp32 workgroup, i.e. four per CU (+ 8: 43 008 B, three per CU, 5 % slower although its operand reads conflict less)
    __host__ __device__ static constexpr int width(int s) { return K - 16 * s; }
    __host__ __device__ static constexpr int ld(int s) { return width(s) + 4; }
    __host__ __device__ static constexpr int roff(int s) { return 16 * s * (K + 4) - 128 * s * (s - 1); }   // 16 * sum_{t<s} ld(t)
    static constexpr int RWORDS = roff(NT);
    template <typename T> static constexpr size_t lds_bytes() { return (size_t)K * 8 + (size_t)RWORDS * sizeof(T) + 2 * K * sizeof(T); }
    // row-major upper index of tile (I, J), I <= J; with NW waves its owner is wave tri % NW, slot tri / NW
    __host__ __device__ static constexpr int tri(int I, int J) { return I * NT - (I * (I - 1)) / 2 + (J - I); }
    // ... and back: block row / column of tile `t`
    __host__ __device__ static constexpr int tile_i(int t) { int I = 0; while (t >= NT - I) { t -= NT - I; ++I; } return I; }
    __host__ __device__ static constexpr int tile_j(int t) { int I = 0; while (t >= NT - I) { t -= NT - I; ++I; } return I + t; }
};

// ---------------------------------------------------------------------------
// sum x, sum x x^T (fp64 accumulation of the fp32 columns): workgroup w takes a contiguous slice
// of columns, thread t owns the outputs e = t, t + 256, ... of  prod[K*K] | sum[K]; the partials
// are added in workgroup order by k_colstats_f32_final, whose last block publishes the blob.
// ---------------------------------------------------------------------------
// sum x x^T = X X^T on the 16x16x4 f64 MFMA (the fp32 entries widened: products exact, sums fp64).  One single-wave
// workgroup per (slice of columns, upper 16 x 16 tile): nsl x 36 waves spread over the chip, each loading only the two
// row blocks of its tile, four k-steps (16 columns) of loads in flight.  The diagonal tiles also sum x.  Partial of a
// slice: the tiles in accumulator layout | sum[K]; k_colstats_f32_final adds the slices in order and un-tiles.
// (First form: one output per thread and column on the VALU, 128 partials of 132 KB: 55 us alone, 0.2 ms beside the
// next sampler.  With the select wrapped around each load the loads of a k-step were serialised: see kernels_wg2.h.)
// (T = double: the fp64 K = 128 context -- the same pass over fp64 columns)
template <int K, typename T = float>
__global__ __launch_bounds__(64) void k_colstats_f32(const T *__restrict__ items, int64_t c0, int64_t c1, int nsl,
                                                     double *__restrict__ partials)
{
    constexpr int NT = K / 16, NTRI = NT * (NT + 1) / 2, PARTW = NTRI * 256 + K;
    const int tri = blockIdx.x % NTRI, sl = blockIdx.x / NTRI;
    int I = 0, t = tri;
    while (t >= NT - I) { t -= NT - I; ++I; }
    const int J = I + t;
    const int lane = threadIdx.x, kq = lane >> 4, li = lane & 15;
    const int64_t n = c1 - c0;
    const int64_t per = ((n + nsl - 1) / nsl + 3) / 4 * 4;
    const int64_t b = c0 + sl * per, e = (b + per < c1) ? b + per : c1;
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    double r = 0.0;
    const T *xi = items + 16 * I + li, *xj = items + 16 * J + li;
    // 16 columns per trip; the loads of the next trip are issued before the MFMAs of the current one
    T fa[4], fb[4], na[4], nb[4];
    auto fetch = [&](int64_t c, T (&a4)[4], T (&b4)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t cc = c + 4 * u + kq;
            const size_t at = (size_t)((cc < e) ? cc : b) * K;       // (beyond the slice: any valid column, masked below)
            a4[u] = xi[at];
            b4[u] = xj[at];
        }
    };
    if (b < e) fetch(b, fa, fb);
    for (int64_t c = b; c < e; c += 16) {
        fetch(c + 16 < e ? c + 16 : b, na, nb);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = c + 4 * u + kq < e;
            const double ya = ok ? (double)fa[u] : 0.0, yb = ok ? (double)fb[u] : 0.0;
            acc = mfma16(ya, yb, acc);
            r += ya;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { fa[u] = na[u]; fb[u] = nb[u]; }
    }
    double *p = partials + (size_t)sl * PARTW;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) p[tri * 256 + reg * 64 + lane] = acc[reg];
    if (I == J) {                                                     // (workgroup-uniform)
        r += __shfl_xor(r, 16);
        r += __shfl_xor(r, 32);
        if (kq == 0) p[NTRI * 256 + 16 * I + li] = r;
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_colstats_f32_final(const double *__restrict__ partials, int nsl,
                                                            const unsigned long long *__restrict__ fail_in,
                                                            double *__restrict__ out, unsigned *ticket, unsigned *flag, unsigned seq)
{
    constexpr int NT = K / 16, NTRI = NT * (NT + 1) / 2, PARTW = NTRI * 256 + K, NOUT = K * K + K;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o < NOUT) {
        int at;
        if (o < K * K) {                                              // prod(gi, gj) at gi + gj K: from tile (min, max) of the block pair
            int gi = o % K, gj = o / K;
            if (gi / 16 > gj / 16) { const int x = gi; gi = gj; gj = x; }
            const int I = gi / 16, J = gj / 16, ii = gi % 16;
            at = (I * NT - (I * (I - 1)) / 2 + (J - I)) * 256 + (ii >> 2) * 64 + (ii & 3) * 16 + (gj % 16);   // D[kq + 4 reg][li] of the f64 16x16x4 shape
        } else {
            at = NTRI * 256 + (o - K * K);
        }
        double s = 0.0;
        for (int q0 = 0; q0 < nsl; q0 += 8) {                         // eight loads in flight, added in slice order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)((q0 + u < nsl) ? q0 + u : q0) * PARTW + at];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (q0 + u < nsl) ? v[u] : 0.0;
        }
        __hip_atomic_store(&out[o], s, BPMF_RLX_SYSTEM);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long fw = *fail_in;
        __hip_atomic_store(&out[NOUT], (fw == ~0ull) ? 0.0 : (double)(fw + 1ull), BPMF_RLX_SYSTEM);
        __hip_atomic_store(&reinterpret_cast<unsigned long long *>(out)[NOUT + 1], fw, BPMF_RLX_SYSTEM);
    }
    publish_when_last(ticket, gridDim.x, flag, seq);
}

// The same pass as RIDERS at the head of the next sampler launch (k_sample_wg2, workgroups of NW waves): job = (slice, tile)
// per wave as in k_colstats_f32, partials written through, a ticket per workgroup; the last arrivers wait until every
// partial has landed (every workgroup takes its ticket before it waits: the count completes as soon as all riders have
// run, and they are the first workgroups of the grid) and add the slices in order, one output per thread -- the sums of
// k_colstats_f32_final, bit for bit.  Why: as kernels of their own on the side's stream the pass needed a head start over
// the partner's sampler, bought with two cross-queue event hops of ~15 us each per half-iteration (DESIGN.md section 4,
// "what the K = 128 iteration is made of"); as riders the partner's launch follows its predecessor on ONE queue.
template <int K, int NW, typename T = float>
__device__ __forceinline__ void colstats_f32_rider(const StatRiders &r, int rb, int tid)
{
    constexpr int NT = K / 16, NTRI = NT * (NT + 1) / 2, PARTW = NTRI * 256 + K, NOUT = K * K + K, NTH = 64 * NW;
    __shared__ unsigned stk;
    const int wave = tid >> 6, lane = tid & 63, kq = lane >> 4, li = lane & 15;
    const int job = rb * NW + wave;
    if (job < r.nsl * NTRI) {                                         // wave-uniform
        const int tri = job % NTRI, sl = job / NTRI;
        int I = 0, t = tri;
        while (t >= NT - I) { t -= NT - I; ++I; }
        const int J = I + t;
        const int64_t n = r.c1 - r.c0;
        const int64_t per = ((n + r.nsl - 1) / r.nsl + 3) / 4 * 4;
        const int64_t b = r.c0 + sl * per, e = (b + per < r.c1) ? b + per : r.c1;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        double rs = 0.0;
        const T *xi = reinterpret_cast<const T *>(r.items) + 16 * I + li, *xj = reinterpret_cast<const T *>(r.items) + 16 * J + li;
        T fa[4], fb[4], na[4], nb[4];
        auto fetch = [&](int64_t c, T (&a4)[4], T (&b4)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t cc = c + 4 * u + kq;
                const size_t at = (size_t)((cc < e) ? cc : b) * K;
                a4[u] = xi[at]; b4[u] = xj[at];
            }
        };
        if (b < e) fetch(b, fa, fb);
        for (int64_t c = b; c < e; c += 16) {
            fetch(c + 16 < e ? c + 16 : b, na, nb);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = c + 4 * u + kq < e;
                const double ya = ok ? (double)fa[u] : 0.0, yb = ok ? (double)fb[u] : 0.0;
                acc = mfma16(ya, yb, acc);
                rs += ya;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { fa[u] = na[u]; fb[u] = nb[u]; }
        }
        double *p = r.partials + (size_t)sl * PARTW;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) __hip_atomic_store(&p[tri * 256 + reg * 64 + lane], acc[reg], BPMF_RLX_AGENT);
        if (I == J) {
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            if (kq == 0) __hip_atomic_store(&p[NTRI * 256 + 16 * I + li], rs, BPMF_RLX_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's partial has landed
    __syncthreads();
    if (tid == 0) stk = __hip_atomic_fetch_add(r.ticket, 1u, BPMF_RLX_AGENT);
    __syncthreads();
    const int tk = (int)stk;
    const int nfin = r.nblocks < (NOUT + NTH - 1) / NTH ? r.nblocks : (NOUT + NTH - 1) / NTH;
    if (tk < r.nblocks - nfin) return;
    const int f = tk - (r.nblocks - nfin);                            // finisher 0 .. nfin-1
    if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(r.ticket, BPMF_RLX_AGENT) < (unsigned)r.nblocks) {
            __builtin_amdgcn_s_sleep(1);
            if (r.wait_ticks && wall_clock64() - t0 > r.wait_ticks) { flag_timeout(r.tmo, BPMF_TMO_STATS); break; }
        }
    }
    __syncthreads();
    for (int o = f * NTH + tid; o < NOUT; o += nfin * NTH) {
        int at;
        if (o < K * K) {                                              // prod(gi, gj) at gi + gj K: from tile (min, max) of the block pair
            int gi = o % K, gj = o / K;
            if (gi / 16 > gj / 16) { const int x = gi; gi = gj; gj = x; }
            const int I = gi / 16, J = gj / 16, ii = gi % 16;
            at = (I * NT - (I * (I - 1)) / 2 + (J - I)) * 256 + (ii >> 2) * 64 + (ii & 3) * 16 + (gj % 16);
        } else {
            at = NTRI * 256 + (o - K * K);
        }
        double s = 0.0;
        for (int q0 = 0; q0 < r.nsl; q0 += 8) {                       // eight loads in flight, added in slice order (as k_colstats_f32_final)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(&r.partials[(size_t)((q0 + u < r.nsl) ? q0 + u : q0) * PARTW + at], BPMF_RLX_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (q0 + u < r.nsl) ? v[u] : 0.0;
        }
        __hip_atomic_store(&r.out[o], s, BPMF_RLX_SYSTEM);
    }
    if (f == 0 && tid == 0) {
        const unsigned long long fw = *r.fail_in;
        __hip_atomic_store(&r.out[NOUT], (fw == ~0ull) ? 0.0 : (double)(fw + 1ull), BPMF_RLX_SYSTEM);
        __hip_atomic_store(&reinterpret_cast<unsigned long long *>(r.out)[NOUT + 1], fw, BPMF_RLX_SYSTEM);
    }
    publish_when_last(r.ticket + 1, (unsigned)nfin, r.flag, r.seq, r.ticket);
}

// ---------------------------------------------------------------------------
// Sys::predict (c++/sample.cpp:48-96) on fp32 factors: fp64 accumulation of the dot product and
// of everything behind it; same partial / publish scheme as k_predict.
// ---------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_predict_f32(const int32_t *__restrict__ tcol, const int32_t *__restrict__ trow,
                                                     const double *__restrict__ tval, int64_t nnz,
                                                     const float *__restrict__ items, const float *__restrict__ other,
                                                     int64_t col_from, double mean, int n, double *__restrict__ pavg,
                                                     double *__restrict__ pm2, double *partial, double *__restrict__ out,
                                                     unsigned *ticket, unsigned *flag, unsigned seq)
{
    __shared__ double red[2][4];
    __shared__ double fin[2][256];
    __shared__ unsigned last;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double se = 0.0, se_avg = 0.0;
    if (q < nnz) {
        const float4 *m = reinterpret_cast<const float4 *>(items + (size_t)(col_from + tcol[q]) * K);
        const float4 *u = reinterpret_cast<const float4 *>(other + (size_t)trow[q] * K);
        double d0 = 0.0, d1 = 0.0;
#pragma unroll 8
        for (int t = 0; t < K / 4; ++t) {
            const float4 x = m[t], y = u[t];
            d0 = fma((double)x.x, (double)y.x, d0);
            d1 = fma((double)x.y, (double)y.y, d1);
            d0 = fma((double)x.z, (double)y.z, d0);
            d1 = fma((double)x.w, (double)y.w, d1);
        }
        const double pred = (d0 + d1) + mean;                       // :78
        const double v = tval[q];
        se = (v - pred) * (v - pred);
        double avg = pavg[q];
        const double delta = pred - avg;
        avg = (n == 0) ? pred : (avg + delta / n);                  // :84 (n, not n+1: reference quirk)
        pavg[q] = avg;
        pm2[q] = (n == 0) ? 0.0 : pm2[q] + delta * (pred - avg);    // :86
        se_avg = (v - avg) * (v - avg);
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        se += __shfl_xor(se, sh);
        se_avg += __shfl_xor(se_avg, sh);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = se; red[1][wv] = se_avg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partial[2 * blockIdx.x], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), BPMF_RLX_AGENT);
        __hip_atomic_store(&partial[2 * blockIdx.x + 1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]), BPMF_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, BPMF_RLX_AGENT);
        last = (t == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    const int64_t nblocks = gridDim.x;
    double sa = 0.0, sb = 0.0;
    for (int64_t wgi = threadIdx.x; wgi < nblocks; wgi += 256) {
        sa += __hip_atomic_load(&partial[2 * wgi], BPMF_RLX_AGENT);
        sb += __hip_atomic_load(&partial[2 * wgi + 1], BPMF_RLX_AGENT);
    }
    fin[0][threadIdx.x] = sa; fin[1][threadIdx.x] = sb;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) { fin[0][threadIdx.x] += fin[0][threadIdx.x + st]; fin[1][threadIdx.x] += fin[1][threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&out[0], fin[0][0], BPMF_RLX_SYSTEM);
        __hip_atomic_store(&out[1], fin[1][0], BPMF_RLX_SYSTEM);
        __hip_atomic_store(ticket, 0u, BPMF_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, seq, BPMF_RLX_SYSTEM);
    }
}

}  // namespace bpmf
