// Sweeps pv_sincos_small (audiomod_amd/csrc/pv_sincos.h, the code the synthesis kernels run) against double-precision
// sin / cos: every float in a band around each multiple of pi/4 up to the argument bound, dense samples of [-pi, pi]
// and of the whole accepted range, special values.  Bar: 2 ulp of the result or 1.2e-7 absolute (near the zeros of
// either function the absolute term is the meaningful one, as for any single-precision sine).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "pv_sincos.h"

int main() {
    long n = 0, bad = 0;
    double worst_ulp = 0, worst_abs = 0;
    auto check = [&](float x) {
        float s, c;
        pv_sincos_small(x, s, c);
        const double ws = std::sin((double)x), wc = std::cos((double)x);
        for (int k = 0; k < 2; ++k) {
            const double got = k ? c : s, want = k ? wc : ws;
            const double err = std::fabs(got - want);
            const double ulp = std::ldexp(1.0, std::ilogb(std::fabs(want) > 1e-30 ? std::fabs(want) : 1e-30) - 23);
            if (err > worst_abs) worst_abs = err;
            if (std::fabs(want) > 1e-3 && err / ulp > worst_ulp) worst_ulp = err / ulp;
            if (err > 2.0 * ulp && err > 1.2e-7) {
                if (bad < 10) printf("MISMATCH x=%a %s got=%.9g want=%.9g\n", x, k ? "cos" : "sin", got, want);
                ++bad;
            }
        }
        ++n;
    };
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<float> in_pi(-3.14159274f, 3.14159274f), wide(-PV_SINCOS_MAX_ARG, PV_SINCOS_MAX_ARG);
    for (int i = 0; i < 20000000; ++i) check(in_pi(rng));
    for (int i = 0; i < 20000000; ++i) check(wide(rng));
    for (int m = -20; m <= 20; ++m) { // neighbourhoods of the multiples of pi/4: reduction boundaries and zeros
        const float t = (float)(m * 0.78539816339744831);
        if (std::fabs(t) > PV_SINCOS_MAX_ARG) continue;
        uint32_t u = pv_sc_f2u(t);
        for (int d = -50000; d <= 50000; ++d) check(pv_sc_u2f(u + d));
    }
    for (int e = -149; e <= 0; ++e) check(std::ldexp(1.0f, e)), check(-std::ldexp(1.0f, e));
    check(0.f), check(-0.f);
    float s, c;
    pv_sincos_small(NAN, s, c);
    if (!(s != s) || !(c != c)) printf("NaN does not come out as NaN\n"), ++bad;
    pv_sincos_small(INFINITY, s, c);
    if (!(s != s) || !(c != c)) printf("infinity does not come out as NaN\n"), ++bad;
    printf("%ld checked, worst %.3f ulp (|result| > 1e-3), worst absolute %.3g, %ld mismatches\n", n, worst_ulp, worst_abs, bad);
    return bad != 0;
}
