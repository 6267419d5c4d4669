// host check of polar_mult (bpmf_amd/csrc/philox.h) against the long-double evaluation of sqrt(-2 log(r2) / r2):
//   g++ -O2 -ffp-contract=off tools/probes/polar_mult_check.cpp -o /tmp/pmc && /tmp/pmc [n]
// prints the largest error in ulp of the exact value over n random r2 of the polar method's own distribution
// (r2 = x^2 + y^2 of canonical doubles, accepted ones), a log-uniform sweep down to 2^-104, and the edge values.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "../../bpmf_amd/csrc/philox.h"

static double ulp_err(double got, long double want)
{
    if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
    int e;
    frexpl(want, &e);
    const long double ulp = ldexpl(1.0L, e - 53);
    return (double)(fabsl((long double)got - want) / ulp);
}
static long double exact(double r2) { return sqrtl(-2.0L * logl((long double)r2) / (long double)r2); }

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    double worst = 0.0, worst_at = 0.0, worst_libm = 0.0;
    auto probe = [&](double r2) {
        const long double want = exact(r2);
        const double e = ulp_err(bpmf::polar_mult(r2), want);
        if (e > worst) { worst = e; worst_at = r2; }
        const double el = ulp_err(std::sqrt(-2 * std::log(r2) / r2), want);
        if (el > worst_libm) worst_libm = el;
    };
    uint32_t blk = 0;
    long done = 0;
    while (done < n) {
        const bpmf::Philox4 b = bpmf::stream_block(7u, blk++);
        const double x = 2.0 * bpmf::canonical53(b.w[3], b.w[2]) - 1.0, y = 2.0 * bpmf::canonical53(b.w[1], b.w[0]) - 1.0;
        const double r2 = x * x + y * y;
        if (r2 > 1.0 || r2 == 0.0) continue;
        probe(r2);
        ++done;
    }
    for (int e = 0; e >= -104; --e)
        for (int t = 0; t < 2000; ++t) {
            const bpmf::Philox4 b = bpmf::stream_block(11u, (uint32_t)(-e * 2000 + t));
            probe(std::ldexp(0.5 + 0.5 * bpmf::canonical53(b.w[1], b.w[0]), e));
        }
    const double edges[] = {1.0, std::nextafter(1.0, 0.0), 0.5, std::nextafter(0.5, 0.0), std::nextafter(0.5, 1.0), 0.70710678118654752440,
                            std::nextafter(0.70710678118654752440, 0.0), std::nextafter(0.70710678118654752440, 1.0), std::ldexp(1.0, -104),
                            std::ldexp(1.0, -52), 0.25, 0.75, 0.99999, 1e-300};
    for (double r2 : edges) probe(r2);
    printf("polar_mult: max error %.3f ulp (at r2 = %.17g); libm expression: %.3f ulp; polar_mult(1) = %g\n", worst, worst_at, worst_libm,
           bpmf::polar_mult(1.0));
    return worst <= 2.0 && bpmf::polar_mult(1.0) == 0.0 ? 0 : 1;
}
