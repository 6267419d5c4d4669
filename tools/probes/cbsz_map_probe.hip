// What do cbsz / abid do to v_mfma_f64_4x4x4_4b_f64 on gfx950?  Brute force: A = unit vector at lane p, B = unit vector at lane q,
// for all 64 x 64 (p, q): out lane r is 1 iff (A lane p) x (B lane q) contributes to D lane r.  Prints, per setting, for each
// output lane r = 16 i + 4 b + j the contributing (p, q) pairs, compactly (only where it differs from the plain instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/cbsz_map_probe.hip -o tools/probes/cbsz_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CB, int AB>
__global__ void k(unsigned long long *out)      // out[p * 64 + q] = ballot of lanes with a non-zero result
{
    const int l = threadIdx.x;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q) {
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == p ? 1.0 : 0.0, l == q ? 1.0 : 0.0, 0.0, CB, AB, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (l == 0) out[p * 64 + q] = m;
        }
}
template <int CB, int AB>
void run(unsigned long long *o, const std::vector<unsigned long long> *base, std::vector<unsigned long long> *keep)
{
    k<CB, AB><<<1, 64>>>(o);
    std::vector<unsigned long long> h(4096);
    hipMemcpy(h.data(), o, 4096 * 8, hipMemcpyDeviceToHost);
    if (keep) *keep = h;
    int diff = 0;
    if (base) for (int t = 0; t < 4096; ++t) diff += h[t] != (*base)[t];
    printf("cbsz %d abid %d: %d of 4096 (p, q) products land elsewhere than in the plain instruction\n", CB, AB, base ? diff : 0);
    // for output lanes r = 0 (i0 b0 j0), 5 (i0 b1 j1), 22 (i1 b1 j2), 63: list the (p, q) that feed them
    for (int r : {0, 5, 22, 47, 63}) {
        printf("   D lane %2d (i%d b%d j%d) <-", r, r >> 4, (r >> 2) & 3, r & 3);
        for (int p = 0; p < 64; ++p) for (int q = 0; q < 64; ++q) if (h[p * 64 + q] >> r & 1) printf(" A%d(k%d b%d i%d)*B%d(k%d b%d j%d)", p, p >> 4, (p >> 2) & 3, p & 3, q, q >> 4, (q >> 2) & 3, q & 3);
        printf("\n");
    }
}
int main()
{
    unsigned long long *o; hipMalloc(&o, 4096 * 8);
    std::vector<unsigned long long> base;
    run<0, 0>(o, nullptr, &base);
    run<2, 0>(o, &base, nullptr); run<2, 1>(o, &base, nullptr); run<2, 3>(o, &base, nullptr); run<1, 1>(o, &base, nullptr);
    return 0;
}
