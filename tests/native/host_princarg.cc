// tests/native/host_princarg.cc -- CPU check that the divide-free princarg used where the argument is the sum or
// difference of two wrapped phases (|a| <= 2 pi + a few float ulp) is bit-identical to the reference expression
// mod(a + pi, -2 pi) + pi with mod(x, y) = x - y * floor(x / y) in double (reference sys.h:84,91).
// Build: g++ -O2 -std=c++17 -ffp-contract=off tests/native/host_princarg.cc -o /tmp/host_princarg
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

static const double PV_PI = 3.14159265358979323846;

static double princarg_ref(double a) {
    const double x = a + PV_PI;
    const double y = -2.0 * PV_PI;
    return (x - (y * std::floor(x / y))) + PV_PI;
}
// same text as audiomod_amd/csrc/pv_kernels.hip princarg_small()
static double princarg_small(double a) {
    const double x = a + PV_PI;
    const double Y = 2.0 * PV_PI;
    const double yn = x > 0.0 ? (x > Y ? 2.0 * Y : Y) : 0.0;
    return (x - yn) + PV_PI;
}
static bool same(double p, double q) { return std::memcmp(&p, &q, sizeof p) == 0 || (p == 0.0 && q == 0.0); }

int main() {
    long bad = 0, n = 0;
    auto check = [&](double a) {
        ++n;
        const double p = princarg_ref(a), q = princarg_small(a);
        const float pf = (float)p, qf = (float)q;
        if (!same(p, q) || std::memcmp(&pf, &qf, 4) != 0) {
            if (bad < 10) std::printf("mismatch a=%.17g ref=%.17g small=%.17g\n", a, p, q);
            ++bad;
        }
    };
    // doubles around every decision point, stepped ulp by ulp
    const double pts[] = {-2.0 * PV_PI, -PV_PI, 0.0, PV_PI, 2.0 * PV_PI, -6.2831854820251465, 6.2831854820251465,
                          -3.1415927410125732, 3.1415927410125732};
    for (double c : pts) {
        double lo = c, hi = c;
        for (int i = 0; i < 20000; ++i) {
            check(lo);
            check(hi);
            lo = std::nextafter(lo, -100.0);
            hi = std::nextafter(hi, 100.0);
        }
    }
    // every float sum / difference of two float phases near the wrap points, and random ones
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<float> U(-3.14159274f, 3.14159274f);
    for (long i = 0; i < 20000000; ++i) {
        const float p = U(rng), r = U(rng);
        check((double)(p + r));
        check((double)(p - r));
    }
    const float pif = 3.14159274f;
    float f = pif;
    for (int i = 0; i < 200000; ++i) {
        check((double)(f + pif));
        check((double)(-f - pif));
        check((double)(f - f));
        check((double)(f + (-pif)));
        f = std::nextafterf(f, 0.f);
    }
    std::printf("%ld values, %ld mismatches\n", n, bad);
    return bad != 0;
}
