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

// a / b, correctly rounded, for operands in a SAFE range: both normal with exponents in [2^-63, 2^63], so that
// neither the quotient nor the intermediate residuals can overflow, underflow or go subnormal.  On the device: the
// hardware reciprocal (1 ulp) refined once, the quotient, one exact-residual correction -- six instructions where
// the compiler's general division spends eleven on scaling and fix-up the safe range does not need.  The residual
// is exact (fma) and the refined reciprocal is good to ~2^-46, so the sum before the final rounding is within
// 2^-40 ulp of a / b, while a quotient of two floats is never closer than 2^-25 ulp to a rounding boundary: correctly
// rounded.  Checked against the compiler's division on 8.4e9 pairs (tools/divtest.hip).  On the host: a / b.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float pv_div_safe(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
#else
static inline float pv_div_safe(float a, float b) { return a / b; }
#endif

PV_AT_HD uint32_t pv_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PV_AT_HD float pv_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Written as one straight line of code with selects: no branches, no indexed constant tables.  On the device a
// lane-dependent table index would put the tables in scratch memory, and the reference's early returns would run
// one after the other under divergence (a first, literal version cost 219 vector instructions and three divisions
// per call; this one 2 divisions).  Each candidate numerator / denominator of the argument reduction is formed with
// the reference's own operations and ONE division serves whichever interval the lane is in (x / 1 is x exactly, for
// the interval that is not reduced); the special cases of e_atan2f.c overwrite the general result at the end, in
// the reference's order of precedence.  Its x == 1 shortcut (atanf(y)) gives what the general path gives and is
// dropped.

// atanf for a non-negative argument (|y / x|): s_atanf.c without the sign handling
PV_AT_HD float pv_atanf_pos_fd(const float q) {
    const int32_t ix = (int32_t)pv_f2u(q); // q >= 0 or NaN: the sign bit is clear
    const bool r0 = ix >= 0x3ee00000, r1 = ix >= 0x3f300000, r2 = ix >= 0x3f980000, r3 = ix >= 0x401c0000;
    float num = q, den = 1.0f, hi = 0.f, lo = 0.f;
    if (r0) num = 2.0f * q - 1.0f, den = 2.0f + q, hi = 4.6364760399e-01f, lo = 5.0121582440e-09f;
    if (r1) num = q - 1.0f, den = q + 1.0f, hi = 7.8539812565e-01f, lo = 3.7748947079e-08f;
    if (r2) num = q - 1.5f, den = 1.0f + 1.5f * q, hi = 9.8279368877e-01f, lo = 3.4473217170e-08f;
    if (r3) num = -1.0f, den = q, hi = 1.5707962513e+00f, lo = 7.5497894159e-08f;
    // (den is 1 or lies in [1.4, 2^25]; num is 0 or at least 2^-24 in magnitude and below 2^25: the safe range --
    // except for an argument that is not reduced and tiny, below 2^-29 . 2^-34, whose result is overwritten below)
    const float t = pv_div_safe(num, den);
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f +
                          w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f +
                          w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    const float reduced = hi - ((t * (s1 + s2) - lo) - t);
    float r = r0 ? reduced : t - t * (s1 + s2);
    if (ix < 0x31000000) r = q;                                              // |x| < 2^-29
    if (ix >= 0x4c000000) r = 1.5707962513e+00f + 7.5497894159e-08f;         // |x| >= 2^25
    if (ix > 0x7f800000) r = q + q;                                          // NaN
    return r;
}

// The same for FINITE y and x (what a transform of finite samples produces): the infinity / NaN cases of the
// reference cannot occur and are left out; zeros, signed zeros and out-of-range quotients are handled as above.
PV_AT_HD float pv_atan2f_fd_finite(const float y, const float x) {
    const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)pv_f2u(x), hy = (int32_t)pv_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const int32_t k = (iy - ix) >> 23;
    // y / x: the short division when both exponents are within [2^-63, 2^63] (wave-uniform test, practically always
    // true for spectra of audio), the compiler's general one otherwise.  (x = 0 gives inf or NaN: overwritten below.)
    float quot;
#if defined(__HIP_DEVICE_COMPILE__)
    const bool safe = (uint32_t)(ix - 0x20000000) < 0x3f000000u && (uint32_t)(iy - 0x20000000) < 0x3f000000u;
    if (__builtin_amdgcn_ballot_w64(!safe && ix != 0 && iy != 0) == 0) quot = pv_div_safe(y, x);
    else quot = y / x;
#else
    quot = y / x;
#endif
    float z = pv_atanf_pos_fd(pv_u2f(pv_f2u(quot) & 0x7fffffffu));
    if (hx < 0 && k < -60) z = 0.0f;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    const float zq = z - pi_lo;
    float r = hx < 0 ? pi - zq : z;                              // atan(+, -) : atan(+, +)
    if (ix == 0) r = pi_o_2 + tiny;                              // x = 0
    if (iy == 0) r = hx < 0 ? pi + tiny : 0.0f;                  // y = 0: +-0 for x >= 0, +-pi for x < 0
    return pv_u2f(pv_f2u(r) | ((uint32_t)hy & 0x80000000u));    // the result takes y's sign: atan(-, .) = -atan(+, .)
}

PV_AT_HD float pv_atan2f_fd(const float y, const float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)pv_f2u(x), hy = (int32_t)pv_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2); // 2 * sign(x) + sign(y)
    // general case: atan(|y / x|), then the quadrant
    const int32_t k = (iy - ix) >> 23;
    float z = pv_atanf_pos_fd(pv_u2f(pv_f2u(y / x) & 0x7fffffffu));
    if (hx < 0 && k < -60) z = 0.0f;              // |y| / x < -2^60
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;       // |y / x| > 2^60
    const float zq = z - pi_lo;
    float r = z;                                                          // atan(+, +)
    if (m == 1) r = pv_u2f(pv_f2u(z) ^ 0x80000000u);                      // atan(-, +)
    if (m == 2) r = pi - zq;                                              // atan(+, -)
    if (m == 3) r = zq - pi;                                              // atan(-, -)
    // special values, later lines taking precedence as the reference's earlier tests do
    const bool ypos = hy >= 0;
    if (iy == 0x7f800000) r = ypos ? pi_o_2 + tiny : -pi_o_2 - tiny;      // y = inf
    if (ix == 0x7f800000) {                                               // x = inf
        const float qinf = hx >= 0 ? pi_o_4 + tiny : 3.0f * pi_o_4 + tiny; // ... and y = inf
        const float fin = hx >= 0 ? 0.0f : pi + tiny;
        const float a = iy == 0x7f800000 ? qinf : fin;
        r = ypos ? a : -a;
    }
    if (ix == 0) r = ypos ? pi_o_2 + tiny : -pi_o_2 - tiny;               // x = 0
    if (iy == 0) r = m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);       // y = 0
    if (ix > 0x7f800000 || iy > 0x7f800000) r = x + y;                    // NaN
    return r;
}

// --------------------------------------------------------------------------------------------------------------
// Table-driven form of the finite-argument variant (round 2): the same operations on the same values, with the
// interval of s_atanf.c's argument reduction looked up instead of selected.  The reduction of every interval is
//     t = (a q + b) / (c q + a)     with (a, b, c) = (1, 0, 0)  (2, -1, 1)  (1, -1, 1)  (1, -1.5, 1.5)  (0, -1, 1)
// -- the reference's q, (2q - 1) / (2 + q), (q - 1) / (q + 1), (q - 1.5) / (1 + 1.5 q), -1 / q: a q is exact, so
// a q + b rounds once like the reference's numerator; c q rounds only for c = 1.5, where the reference rounds too --
// and the result hi - ((t (s1 + s2) - lo) - t), which for the interval that is not reduced (hi = lo = 0) is
// t - t (s1 + s2) as in the reference (rounding is symmetric in sign).  All four thresholds are multiples of 2^18 in
// the float's bit pattern, so bits 18..30 of q, offset and clamped to [0, 80], index an 81-byte map to the row.  A
// tiny q needs no case of its own (t - t (s1 + s2) rounds to t = q below 2^-12), the reference's z = 0 for
// |y / x| < 2^-60 neither (q - pi_lo rounds to -pi_lo), and a quotient of 2^25 and above -- which is where its
// |y / x| > 2^60 case, an infinite quotient (x = 0) and 0 / 0 end up -- takes atanf's constant, which equals
// pi_o_2 + 0.5 pi_lo.  tests/native/host_atan2f.cc sweeps this form against atan2f() with the others.
// The blob (five 32-byte rows, then the byte map) is 256 bytes; the kernels keep a copy in LDS.
// --------------------------------------------------------------------------------------------------------------
#define PV_ATAN_BLOB_WORDS 64
#define PV_ATAN_MAP_OFFSET 160
#define PV_ATAN_KEY_BIAS 0xFB7 // (0x3ee00000 >> 18) - 1
#if defined(__cplusplus)
constexpr uint32_t pv_cf2u(float f) { return __builtin_bit_cast(uint32_t, f); }
constexpr uint32_t pv_atan_row_of_key(int k) { return 32u * ((k >= 1) + (k >= 21) + (k >= 47) + (k >= 80)); }
constexpr uint32_t pv_atan_blob_word(int i) {
    constexpr float rows[5][8] = {
        {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0, 0, 0},
        {2.0f, -1.0f, 1.0f, 4.6364760399e-01f, 5.0121582440e-09f, 0, 0, 0},
        {1.0f, -1.0f, 1.0f, 7.8539812565e-01f, 3.7748947079e-08f, 0, 0, 0},
        {1.0f, -1.5f, 1.5f, 9.8279368877e-01f, 3.4473217170e-08f, 0, 0, 0},
        {0.0f, -1.0f, 1.0f, 1.5707962513e+00f, 7.5497894159e-08f, 0, 0, 0},
    };
    if (i < 40) return pv_cf2u(rows[i / 8][i % 8]);
    uint32_t w = 0;
    for (int b = 0; b < 4; ++b) {
        const int k = 4 * (i - 40) + b;
        w |= pv_atan_row_of_key(k > 80 ? 80 : k) << (8 * b);
    }
    return w;
}
struct PvAtanBlob {
    uint32_t w[PV_ATAN_BLOB_WORDS];
};
constexpr PvAtanBlob pv_atan_make_blob() {
    PvAtanBlob b{};
    for (int i = 0; i < PV_ATAN_BLOB_WORDS; ++i) b.w[i] = pv_atan_blob_word(i);
    return b;
}

// sqrtf(a) for a normal a in [2^-96, 2^127), correctly rounded, in five instructions: y = rsq(a) (1 ulp), s0 = a y,
// the EXACT residual d = a - s0 s0 (fma), and s0 + d (y / 2) in one more fma.  With s0 = r (1 + e0) and
// y / 2 = (1 + eh) / 2r the sum before the final rounding is r (1 - e0^2 / 2 - e0 eh), i.e. within ~2^-20 ulp of
// sqrt(a) -- close enough that only a value of sqrt(a) within that distance of a rounding boundary could round the
// other way, and there is no such float: the device test sweeps EVERY float of the range against the compiler's
// correctly rounded sqrtf (pv_debug_sqrt_sweep, tests/test_gpu_parity.py: 1.87e9 values, 0 mismatches).  The
// compiler's own expansion (v_sqrt, two neighbours, two residuals, two compares, two selects) is nine instructions,
// the selects among the expensive ones (tools/pk_probe.hip).  Host: sqrtf.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float pv_sqrt_safe(const float a) {
    const float y = __builtin_amdgcn_rsqf(a);
    const float s0 = a * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-s0, s0, a);
    return __builtin_fmaf(d, h, s0);
}
#else
PV_AT_HD float pv_sqrt_safe(const float a) { return __builtin_sqrtf(a); }
#endif

// `blob`: the 256 bytes above (16-byte aligned; in LDS on the device).  SAFE: the caller has checked that both
// operands are normal with magnitudes in [2^-48, 2^63) -- y / x then goes through the short division, and the
// zero cases of the reference cannot occur.
template <bool SAFE> PV_AT_HD float pv_atan2f_fd_tab(const float y, const float x, const unsigned char *blob) {
    const float pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    // (x = 0: the short division gives NaN where the IEEE one gives infinity -- both take the constant below)
    const float quot = SAFE ? pv_div_safe(y, x) : y / x;
    const uint32_t uq = pv_f2u(quot);
    int key = (int)((uq >> 18) & 0x1fffu) - PV_ATAN_KEY_BIAS;
    key = key < 0 ? 0 : (key > 80 ? 80 : key);
#if defined(__HIP_DEVICE_COMPILE__)
    // (LDS addresses as integers: the table's offset folds into the instructions' immediate field)
    typedef __attribute__((address_space(3))) const unsigned char lds_u8;
    typedef __attribute__((address_space(3))) const float4 lds_f4;
    typedef __attribute__((address_space(3))) const float lds_f;
    const uint32_t base = (uint32_t)(uintptr_t)blob; // the caller passes the table's LDS address as a number
    const uint32_t off = *(lds_u8 *)(uintptr_t)(base + PV_ATAN_MAP_OFFSET + (uint32_t)key);
    const float4 abch = *(lds_f4 *)(uintptr_t)(base + off);
    const float a = abch.x, b = abch.y, c = abch.z, hi = abch.w;
    const float lo = *(lds_f *)(uintptr_t)(base + off + 16u);
#else
    const uint32_t off = blob[PV_ATAN_MAP_OFFSET + key];
    const float *row = reinterpret_cast<const float *>(blob + off);
    const float a = row[0], b = row[1], c = row[2], hi = row[3];
    const float lo = row[4];
#endif
    const float q = pv_u2f(uq & 0x7fffffffu);
#if defined(__HIP_DEVICE_COMPILE__)
    const float num = __builtin_fmaf(a, q, b); // a q is exact: one rounding either way
#else
    const float num = a * q + b;
#endif
    const float cq = c * q;
    const float den = cq + a;
    const float t = pv_div_safe(num, den);
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f +
                          w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f +
                          w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    float at = hi - ((t * (s1 + s2) - lo) - t);
    if (!(q < 0x1p25f)) at = 1.5707962513e+00f + 7.5497894159e-08f; // |y / x| >= 2^25, inf, NaN
    const float zq = at - pi_lo;
    float r = x < 0.0f ? pi - zq : at;             // (x = -0 is not "negative" here: atan2(y, -0) = +-pi/2)
    if (!SAFE && y == 0.0f) r = pv_u2f((uint32_t)((int32_t)pv_f2u(x) >> 31) & pv_f2u(pi)); // atan2(+-0, x): +-0, +-pi
    (void)pi_o_2;
    return pv_u2f(pv_f2u(r) | (pv_f2u(y) & 0x80000000u));
}
#endif
