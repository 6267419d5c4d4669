// Calibration of rocprofv3's FETCH_SIZE for the sampler's access pattern (MI355X_MICROARCH.md: the
// counter halves wide coalesced streaming reads; other widths are uncalibrated).  Three kernels read
// the same 1 GiB exactly once: (a) streaming, 16 B per lane, a wave reads 1 KiB contiguous;
// (b) the sampler's gather: lane (slot = l >> 2, x = l & 3) reads 16 B at row[slot] * 256 + 64 h + 16 x,
// h = 0..3, rows in a pseudo-random order; (c) 8-byte gathers (lane li of 16 reads 8 B of a row).
// Two more kernels write 1 GiB once (streaming 16 B per lane; 256-B rows in random order) for WRITE_SIZE.
// run: rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o p -- ./fetch_calib   (and --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
static constexpr size_t BYTES = 1ull << 30, ROWS = BYTES / 256;

__device__ __forceinline__ size_t perm(size_t i) { return (i * 2654435761ull + 12345) % ROWS; }   // odd multiplier: a bijection mod 2^22

__global__ __launch_bounds__(256) void k_stream(const d2 *__restrict__ p, double *out)
{
    const size_t n = BYTES / 16, stride = (size_t)gridDim.x * 256;
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { const d2 v = p[i]; s += v.x + v.y; }
    if (s == 1.2345) out[0] = s;
}
__global__ __launch_bounds__(64) void k_gather16(const double *__restrict__ p, double *out)
{
    const int lane = threadIdx.x, slot = lane >> 2, x = lane & 3;
    double s = 0;
    for (size_t g = blockIdx.x; g < ROWS / 16; g += gridDim.x) {
        const size_t row = perm(g * 16 + slot);
        const d2 *q = reinterpret_cast<const d2 *>(p + row * 32 + 2 * x);
#pragma unroll
        for (int h = 0; h < 4; ++h) { const d2 v = q[4 * h]; s += v.x + v.y; }
    }
    if (s == 1.2345) out[0] = s;
}
__global__ __launch_bounds__(64) void k_gather8(const double *__restrict__ p, double *out)
{
    const int lane = threadIdx.x, kq = lane >> 4, li = lane & 15;
    double s = 0;
    for (size_t g = blockIdx.x; g < ROWS / 4; g += gridDim.x) {
        const size_t row = perm(g * 4 + kq);
        s += p[row * 32 + li] + p[row * 32 + 16 + li];
    }
    if (s == 1.2345) out[0] = s;
}
__global__ __launch_bounds__(256) void k_wstream(d2 *__restrict__ p)
{
    const size_t n = BYTES / 16, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = d2{1.0, 2.0};
}
__global__ __launch_bounds__(64) void k_wrows(double *__restrict__ p)          // the sampler's result store: 32 lanes x 8 B per row
{
    for (size_t g = blockIdx.x; g < ROWS; g += gridDim.x) { const size_t row = perm(g); if (threadIdx.x < 32) p[row * 32 + threadIdx.x] = 3.0; }
}
int main()
{
    double *p, *out;
    hipMalloc(&p, BYTES); hipMalloc(&out, 64);
    hipMemset(p, 0, BYTES);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const d2 *)p, out);
        hipLaunchKernelGGL(k_gather16, dim3(16384), dim3(64), 0, 0, (const double *)p, out);
        hipLaunchKernelGGL(k_gather8, dim3(16384), dim3(64), 0, 0, (const double *)p, out);
        hipLaunchKernelGGL(k_wstream, dim3(4096), dim3(256), 0, 0, (d2 *)p);
        hipLaunchKernelGGL(k_wrows, dim3(16384), dim3(64), 0, 0, p);
    }
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes once\n", BYTES);
    return 0;
}
