// Sweeps pv_atan2f_fd (audiomod_amd/csrc/pv_atan2f.h, the very code the analysis kernels run) against the C library's
// atan2f, bit for bit: random bit patterns, spectrum-like magnitudes, the neighbourhoods of every reduction
// threshold, special values.  usage: host_atan2f [millions of random pairs per class, default 20]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "pv_atan2f.h"

int main(int argc, char **argv) {
    const long per = (argc > 1 ? atol(argv[1]) : 20) * 1000000L;
    std::mt19937_64 rng(12345);
    long bad = 0, n = 0, nsafe = 0;
    alignas(16) static const PvAtanBlob blob = pv_atan_make_blob();
    auto check = [&](float y, float x) {
        const float a = pv_atan2f_fd(y, x), b = atan2f(y, x);
        ++n;
        if (std::isfinite(y) && std::isfinite(x)) { // the variant the kernels use: finite arguments only
            const float c = pv_atan2f_fd_finite(y, x);
            if (pv_f2u(c) != pv_f2u(b)) {
                if (bad < 10) printf("MISMATCH (finite variant) y=%a x=%a got=%a want=%a\n", y, x, c, b);
                ++bad;
            }
            const float ay = std::fabs(y), ax = std::fabs(x);
            if (ay >= 0x1p-48f && ay < 0x1p63f && ax >= 0x1p-48f && ax < 0x1p63f) { // what the kernels' fast path is given
                const float f = pv_atan2f_fd_tab<true>(y, x, reinterpret_cast<const unsigned char *>(blob.w));
                ++nsafe;
                if (pv_f2u(f) != pv_f2u(b)) {
                    if (bad < 10) printf("MISMATCH (table variant, safe range) y=%a x=%a got=%a want=%a\n", y, x, f, b);
                    ++bad;
                }
            }
            const float d = pv_atan2f_fd_tab<false>(y, x, reinterpret_cast<const unsigned char *>(blob.w));
            if (pv_f2u(d) != pv_f2u(b)) {
                if (bad < 10) printf("MISMATCH (table variant) y=%a x=%a got=%a want=%a\n", y, x, d, b);
                ++bad;
            }
        }
        if (pv_f2u(a) != pv_f2u(b) && !(a != a && b != b)) {
            if (bad < 10) printf("MISMATCH y=%a x=%a got=%a want=%a\n", y, x, a, b);
            ++bad;
        }
    };
    for (long i = 0; i < per; ++i) {
        const uint64_t r = rng();
        check(pv_u2f((uint32_t)r), pv_u2f((uint32_t)(r >> 32)));
    }
    std::uniform_real_distribution<float> e(-8.f, 3.f);
    for (long i = 0; i < per; ++i) {
        const float y = std::pow(10.f, e(rng)) * ((rng() & 1) ? 1 : -1), x = std::pow(10.f, e(rng)) * ((rng() & 1) ? 1 : -1);
        check(y, x);
    }
    std::uniform_real_distribution<float> ew(-14.f, 18.9f); // the whole range the fast path accepts
    for (long i = 0; i < per; ++i) {
        const float y = std::pow(10.f, ew(rng)) * ((rng() & 1) ? 1 : -1), x = std::pow(10.f, ew(rng)) * ((rng() & 1) ? 1 : -1);
        check(y, x);
    }
    const float th[] = {0.4375f, 0.6875f, 1.1875f, 2.4375f, 1.0f, 3.7e-9f, 3.3554432e7f};
    for (float t : th)
        for (int d = -20000; d <= 20000; ++d) {
            const float r = pv_u2f(pv_f2u(t) + d);
            for (float x : {1.0f, 0.37f, 123.456f, -0.37f, -5e-3f}) {
                check(r * x, x);
                check(-r * x, x);
            }
        }
    const float sp[] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 3.4e38f, -3.4e38f};
    for (float y : sp)
        for (float x : sp) check(y, x);
    printf("%ld checked (%ld in the fast path's range), %ld mismatches\n", n, nsafe, bad);
    return bad != 0;
}
