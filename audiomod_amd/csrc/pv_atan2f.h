// pv_atan2f.h -- atan2f as the reference's libm computes it.
//
// The reference takes its analysis phases from libm's atan2f (FFT::forwardPolar, FFT.cc:2623-2630), and the phase
// propagation that follows is discontinuous in them: delta = omega + princarg(phase - prev_phase - omega) jumps by
// 2 pi when its argument crosses +-pi, and the advance scales that jump by a non-integer (phasevocoderprocess.cc:
// 655-664).  A phase that differs from the reference's in its last bit therefore flips, once in a few stream-minutes,
// a wrap on some spectral peak, and from there on the output is a different (equally valid) signal -- RMS 5e-2
// instead of 4e-6, found by the full-size test of stream 69 of the bench batch.  So the device must produce the
// reference's phases to the bit.  glibc up to 2.40 (the image: 2.35) computes atan2f with fdlibm's e_atan2f.c /
// s_atanf.c -- float arithmetic only, no table, no FMA variant -- whose published algorithm and constants are
// restated here.  Every operation is an IEEE float add / multiply / divide evaluated as written
// (-ffp-contract=off, correctly rounded division), so host and device agree bit for bit with libm;
// tests/native/host_atan2f.cc sweeps 2e8 arguments against atan2f().
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define PV_AT_HD __host__ __device__ __forceinline__
#else
#define PV_AT_HD static inline
#endif

PV_AT_HD uint32_t pv_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PV_AT_HD float pv_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// (written without branches or indexed constant tables: on the device a lane-dependent table index would put the
// tables in scratch memory, and the four argument reductions would run one after the other under divergence; here
// each candidate numerator / denominator is formed with the reference's own operations and ONE division serves
// whichever interval the lane is in -- x / 1 is x exactly, for the interval that is not reduced)
PV_AT_HD float pv_atanf_fd(float x) {
    const int32_t hx = (int32_t)pv_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) { // |x| >= 2^25 (or NaN)
        if (ix > 0x7f800000) return x + x;
        const float r = 1.5707962513e+00f + 7.5497894159e-08f;
        return hx > 0 ? r : -1.5707962513e+00f - 7.5497894159e-08f;
    }
    if (ix < 0x31000000) return x; // |x| < 2^-29
    const float ax = pv_u2f((uint32_t)ix);
    // id: -1 |x| < 7/16 (no reduction), 0 < 11/16, 1 < 19/16, 2 < 39/16, 3 otherwise
    const bool r0 = ix >= 0x3ee00000, r1 = ix >= 0x3f300000, r2 = ix >= 0x3f980000, r3 = ix >= 0x401c0000;
    float num = x, den = 1.0f, hi = 0.f, lo = 0.f;
    if (r0) num = 2.0f * ax - 1.0f, den = 2.0f + ax, hi = 4.6364760399e-01f, lo = 5.0121582440e-09f;
    if (r1) num = ax - 1.0f, den = ax + 1.0f, hi = 7.8539812565e-01f, lo = 3.7748947079e-08f;
    if (r2) num = ax - 1.5f, den = 1.0f + 1.5f * ax, hi = 9.8279368877e-01f, lo = 3.4473217170e-08f;
    if (r3) num = -1.0f, den = ax, hi = 1.5707962513e+00f, lo = 7.5497894159e-08f;
    const float t = num / den;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f +
                          w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f +
                          w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    if (!r0) return t - t * (s1 + s2);
    const float r = hi - ((t * (s1 + s2) - lo) - t);
    return hx < 0 ? -r : r;
}

PV_AT_HD float pv_atan2f_fd(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)pv_f2u(x), hy = (int32_t)pv_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y; // NaN
    if (hx == 0x3f800000) return pv_atanf_fd(y);          // x = 1.0
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);    // 2 * sign(x) + sign(y)
    if (iy == 0) { // y = 0
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny; // x = 0
    if (ix == 0x7f800000) { // x = inf
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny; // y = inf
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;      // |y / x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;        // |y| / x < -2^60
    else z = pv_atanf_fd(pv_u2f(pv_f2u(y / x) & 0x7fffffffu));
    switch (m) {
    case 0: return z;
    case 1: return pv_u2f(pv_f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}
