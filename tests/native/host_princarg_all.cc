// princarg_f (audiomod_amd/csrc/pv_kernels.hip) forms the quotient x / y of the reference's princarg (sys.h:84,91:
// mod(a + pi, -2 pi) + pi with mod(x, y) = x - y floor(x / y)) without a division: q0 = x (1 / y), r = fma(-y, q0, x),
// q1 = fma(r, 1 / y, q0).  Its argument is always a float widened to double, so the claim "q1 is the IEEE quotient"
// can be checked on EVERY argument: all 2^32 bit patterns, quotient and result, bit for bit.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static const double PV_PI = 3.14159265358979323846;
static inline double princarg_reference(double a) {
    const double x = a + PV_PI;
    const double y = -2.0 * PV_PI;
    return (x - (y * std::floor(x / y))) + PV_PI;
}
static inline double princarg_f(const float af, double *quot) {
    const double x = (double)af + PV_PI;
    const double y = -2.0 * PV_PI;
    const double inv_y = 1.0 / (-2.0 * PV_PI);
    const double q0 = x * inv_y;
    const double r = std::fma(-y, q0, x);
    const double q1 = std::fma(r, inv_y, q0);
    *quot = q1;
    return (x - (y * std::floor(q1))) + PV_PI;
}

int main() {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 8) nt = 8;
    std::atomic<long> bad{0}, badq{0}, n{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            long b = 0, bq = 0, c = 0;
            for (uint64_t u = t; u < (1ull << 32); u += nt) {
                const uint32_t v = (uint32_t)u;
                float f;
                memcpy(&f, &v, 4);
                if (!std::isfinite(f)) continue;
                double q1;
                const double got = princarg_f(f, &q1), want = princarg_reference((double)f);
                const double qd = ((double)f + PV_PI) / (-2.0 * PV_PI);
                b += memcmp(&got, &want, 8) != 0;
                bq += memcmp(&q1, &qd, 8) != 0;
                ++c;
            }
            bad += b, badq += bq, n += c;
        });
    for (auto &x : th) x.join();
    printf("%ld floats: %ld result mismatches, %ld quotient mismatches\n", n.load(), bad.load(), badq.load());
    return bad.load() != 0 || badq.load() != 0;
}
