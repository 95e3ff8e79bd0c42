// pv_sincos.h -- sine and cosine of a modest argument for the synthesis kernels' polar -> cartesian conversion
// (FFT.cc:2711-2718: re = mag cos(phase), im = mag sin(phase)).
//
// The reference calls libm's sinf / cosf there; unlike the analysis side's atan2f (pv_atan2f.h) nothing downstream is
// discontinuous in these values, so the bar is the path's float tolerance, not bit identity, and the device library's
// sincosf (1-2 ulp) was what round 1 used.  Its cost on gfx950 is not its ~45 instructions but their kind: a
// Payne-Hanek branch for huge arguments, an infinity / NaN class test and a quadrant swap, all as back-to-back
// v_cndmask_b32 on VCC, which issue several times slower than arithmetic (tools/pk_probe.hip).  The phases that reach
// the conversion are wrapped to [-pi, pi] (or a small multiple of it), so this version has a two-term Cody-Waite
// reduction good to |x| <= 16, polynomials of its own fit on [-pi/4, pi/4] (least squares on Chebyshev nodes:
// sin 0.8 ulp, cos 1.2 ulp measured against double over the reduced range), and a quadrant step made of bit operations.
// NaN and infinity come out as NaN without a test (the reduction turns both into NaN).
// tests/native/host_sincos.cc sweeps it against double-precision sin / cos.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define PV_SC_HD __host__ __device__ __forceinline__
#else
#define PV_SC_HD static inline
#endif

PV_SC_HD uint32_t pv_sc_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PV_SC_HD float pv_sc_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#define PV_SINCOS_MAX_ARG 16.0f

// |x| <= PV_SINCOS_MAX_ARG (callers test; beyond it the two-term reduction loses accuracy)
PV_SC_HD void pv_sincos_small(const float x, float &sn, float &cs) {
    const float n = __builtin_rintf(x * 6.3661977237e-01f);       // x * 2 / pi, to the nearest integer
    float r = __builtin_fmaf(n, -1.5707963705e+00f, x);           // x - n * pi/2 in two pieces (fma: exact product)
    r = __builtin_fmaf(n, 4.3711388287e-08f, r);                  // pi/2 = 1.5707963705 - 4.3711388287e-08
    const int q = (int)n;
    const float u = r * r;
    // sin r = r + r^3 S(u), cos r = 1 + u C(u)
    float ps = __builtin_fmaf(u, -1.9515887834e-04f, 8.3321649581e-03f);
    ps = __builtin_fmaf(u, ps, -1.6666655242e-01f);
    const float s = __builtin_fmaf(r * u, ps, r);
    float pc = __builtin_fmaf(u, 2.4389120881e-05f, -1.3886747183e-03f);
    pc = __builtin_fmaf(u, pc, 4.1666623205e-02f);
    pc = __builtin_fmaf(u, pc, -0.5f);
    const float c = __builtin_fmaf(u, pc, 1.0f);
    // quadrant: odd n swaps the two (cos takes -sin), bit 1 of n flips sin's sign, bit 1 of n + 1 flips cos's
    const uint32_t odd = 0u - ((uint32_t)q & 1u);
    const uint32_t su = pv_sc_f2u(s), cu = pv_sc_f2u(c);
    const uint32_t s_sel = (cu & odd) | (su & ~odd);
    const uint32_t c_sel = (su & odd) | (cu & ~odd);
    sn = pv_sc_u2f(s_sel ^ (((uint32_t)q << 30) & 0x80000000u));
    cs = pv_sc_u2f(c_sel ^ ((((uint32_t)q + 1u) << 30) & 0x80000000u));
}
