// pv_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the phase-vocoder engine.
//
// Kernels (the HBM-visible stages of SURVEY.md section 8(d); the phase stage is split three ways):
//   pv_analyze_kernel : window -> fftshift -> real FFT (LDS butterflies) -> (mag, phase)
//                       replaces analyzeSlice + FFT::forwardPolar + kiss_fftr
//                       (reference phasevocoderprocess.cc:492-503, FFT.cc:2617-2631, kiss_fftr.c:67-121)
//                       + spectral peak picking of modifySlicePhaseLocked (:587-596)
//   pv_match_kernel   : phase-locked mode, parallel part: peak matching and everything about a peak's
//                       rotation that does not depend on the recurrence (:640-663)
//   pv_seq_kernel     : phase-locked mode, sequential part: the per-peak rotation chain, one workgroup
//                       per (stream, channel) row (:664-665); per-bin INIT / no-peak steps (:606-636)
//   pv_prop_kernel    : coremode 0, per-bin recurrence of modifySliceSimple (:708-753)
//   pv_synth_kernel   : freqComp gather -> mag/N * (cosf, sinf) -> inverse real FFT -> ifftshift * window
//                       replaces freqCompSlice + synthesiseSlice + FFT::inversePolar + kiss_fftri
//                       (phasevocoderprocess.cc:842-923,1001-1075, FFT.cc:2711-2721, kiss_fftr.c:123-159)
//   pv_ola_kernel     : overlap-add gather + window-sum normalisation + Speex Q4 resampling
//                       replaces the accumulate/divide/shift of synthesiseSlice/writeSlice and
//                       resampler::doresample (phasevocoderprocess.cc:1057,1073,1140-1194,
//                       speex/resample.c:353-401,462-560)
//
// Arithmetic contract: this file is compiled with -ffp-contract=off.  Every float expression
// keeps the reference's operand order and rounding points so the FFT, magnitudes, peak picking,
// OLA and resampler MACs are bit-identical to the x86 reference.  The analysis phases are too: atan2f is libm's own
// algorithm (pv_atan2f.h), because the phase propagation is discontinuous in them.  Only the synthesis' sine / cosine
// are approximations of the reference's (pv_sincos.h for wrapped phases, the device libm elsewhere: 1-2 ulp; the output is
// continuous in them).  princarg stays in double with a true IEEE divide, as the reference
// (common/system/sys.h:84,91).
#include "pv_kernels.h"
#include "pv_atan2f.h"
#include "pv_sincos.h"
#include "pv_wavefft.h"

#include <hip/hip_runtime.h>
#include <cstdlib>

namespace pv {

// -DPV_POISON (a debugging build, tools/build_variant.sh poison -DPV_POISON): every kernel first fills its whole
// dynamic LDS with a NaN pattern, and the engine fills every device buffer with 0xFF bytes instead of zeros, so that a
// read of anything the code did not write itself shows up in the output deterministically instead of depending on what
// the CU or the memory held before.  The product build contains none of this.
#ifdef PV_POISON
__device__ __forceinline__ void pv_poison_lds(char *base) {
    const uint32_t bytes = ((const __attribute__((address_space(4))) uint32_t *)__builtin_amdgcn_dispatch_ptr())[7]; // group_segment_size
    const int nt = blockDim.x * blockDim.y * blockDim.z, tid = threadIdx.x;
    for (uint32_t i = tid; i < bytes / 4; i += nt) reinterpret_cast<uint32_t *>(base)[i] = 0x7fc0dead;
    __syncthreads();
}
#define PV_POISON_LDS(base) pv_poison_lds(base)
#else
#define PV_POISON_LDS(base) ((void)0)
#endif

#define PV_PI 3.14159265358979323846

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    float2 m;
    m.x = a.x * b.x - a.y * b.y;
    m.y = a.x * b.y + a.y * b.x;
    return m;
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// princarg(a) = mod(a + pi, -2 pi) + pi with mod(x, y) = x - y*floor(x/y), in double (sys.h:84,91), for an argument
// that is a FLOAT (every call site: the sum or difference of float phases, widened).  floor() needs the correctly
// rounded quotient x / y; the divisor is a constant, so the quotient is formed without a division -- q0 = x (1 / y),
// the exact residual r = x - y q0 (fma), q1 = q0 + r (1 / y) (fma) -- and q1 IS the IEEE quotient for every one of
// the 2^32 float arguments: tests/native/host_princarg_all.cc compares it, and the function's result, with the
// reference expression bit for bit on all of them.  Three dependent operations where the division is about ten:
// this is the critical path of the rotation chain's step.
__device__ __forceinline__ double princarg_f(const float af) {
    const double x = (double)af + PV_PI;
    const double y = -2.0 * PV_PI;
    constexpr double inv_y = 1.0 / (-2.0 * PV_PI);
    const double q0 = x * inv_y;
    const double r = __builtin_fma(-y, q0, x);
    const double q1 = __builtin_fma(r, inv_y, q0);
    return (x - (y * floor(q1))) + PV_PI;
}

// atan2f's interval table (pv_atan2f.h) as the kernels read it into LDS
__device__ const PvAtanBlob pv_atan_blob_dev = pv_atan_make_blob();
__device__ __forceinline__ float pv_max3_abs(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
}
__device__ __forceinline__ float pv_min3_abs(float a, float b, float c) {
    return __builtin_fminf(__builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
}
__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in issue order; this only stops the compiler from moving LDS
    // accesses across the hand-off between two passes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// princarg where the argument is the sum or difference of two wrapped phases (|a| <= 2 pi + a few float ulp):
// floor((a + pi) / (-2 pi)) can only be 0, -1 or -2, and which one follows from comparing x = a + pi with 0 and
// 2 pi -- the correctly rounded quotient x / y lies on the same side of 0 and -1 as the exact one (x / 2 pi differs
// from 1 by at least ulp(x) / 2 pi = 1.4e-16, more than half the spacing of doubles around 1) -- and y * n is exact
// for these n.  Bit-identical to princarg_div without the divide (tests/native/host_princarg.cc sweeps it).
__device__ __forceinline__ double princarg_small(double a) {
    const double x = a + PV_PI;
    const double Y = 2.0 * PV_PI;
    const double yn = x > 0.0 ? (x > Y ? 2.0 * Y : Y) : 0.0;
    return (x - yn) + PV_PI;
}

// XCD-aware block -> (row, slice) map: blocks b and b+8 share an XCD (and its L2), so each XCD walks
// whole rows (one stream-channel) slice after slice and re-reads the overlapping input from its own L2.
__device__ __forceinline__ int ring_slot(int s0, int tl, int TR) {
    const int s = s0 + tl; // tl < TR, so one conditional wrap is enough
    return s >= TR ? s - TR : s;
}
__device__ __forceinline__ int ring_prev(int slot, int TR) { return slot == 0 ? TR - 1 : slot - 1; }

__device__ __forceinline__ bool block_to_row_slice(int Tn, int rows, int &row, int &tl) {
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    row = xcd + 8 * (q / Tn);
    tl = q % Tn;
    return row < rows;
}

// Workgroup size of the generic one-workgroup-per-frame kernels (the sizes without a wave-per-frame transform): a
// 256-point frame is 128 complex points, 32 radix-4 butterflies per stage -- one wave, not four that mostly wait at
// barriers (fft 256 at 128 streams: 1.63 -> 2.17 G samples/s).
static inline int generic_fft_threads(int nc) { return nc <= 128 ? 64 : kFftThreads; }

// Butterfly stages of the half-size complex FFT in LDS, kissfft operation order
// (kf_bfly4 kiss_fft.c:59-103, kf_bfly2 :36-57); stage 0 is the innermost recursion level.
template <bool INV>
__device__ __forceinline__ void fft_stages(float2 *buf, const DevTables &tb, const float2 *__restrict__ tw) {
    const int nt = blockDim.x;
    for (int s = 0; s < tb.nstages; ++s) {
        const int lm = tb.log2m[s], fs = tb.fstride[s];
        const int m = 1 << lm;
        if (tb.radix[s] == 4) {
            const int nb = tb.nc >> 2;
            for (int b = threadIdx.x; b < nb; b += nt) {
                const int k = b & (m - 1);
                const int base = ((b >> lm) << (lm + 2)) + k;
                float2 f0 = buf[base];
                const float2 f1 = buf[base + m], f2 = buf[base + 2 * m], f3 = buf[base + 3 * m];
                const float2 s0 = cmul(f1, tw[k * fs]);
                const float2 s1 = cmul(f2, tw[2 * k * fs]);
                const float2 s2 = cmul(f3, tw[3 * k * fs]);
                const float2 s5 = csub(f0, s1);
                f0 = cadd(f0, s1);
                const float2 s3 = cadd(s0, s2);
                const float2 s4 = csub(s0, s2);
                buf[base + 2 * m] = csub(f0, s3);
                buf[base] = cadd(f0, s3);
                if (INV) {
                    buf[base + m] = make_float2(s5.x - s4.y, s5.y + s4.x);
                    buf[base + 3 * m] = make_float2(s5.x + s4.y, s5.y - s4.x);
                } else {
                    buf[base + m] = make_float2(s5.x + s4.y, s5.y - s4.x);
                    buf[base + 3 * m] = make_float2(s5.x - s4.y, s5.y + s4.x);
                }
            }
        } else {
            const int nb = tb.nc >> 1;
            for (int b = threadIdx.x; b < nb; b += nt) {
                const int k = b & (m - 1);
                const int base = ((b >> lm) << (lm + 1)) + k;
                const float2 f0 = buf[base];
                const float2 t = cmul(buf[base + m], tw[k * fs]);
                buf[base + m] = csub(f0, t);
                buf[base] = cadd(f0, t);
            }
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------
// analysis (+ spectral peak picking for the phase-locked mode)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_peak(const float *smag, int b, int hs) {
    // modifySlicePhaseLocked (:587-596): strict local maximum over +-2 bins, b in [2, hs-3]
    if (b < 2 || b + 2 >= hs) return false;
    const float mb = smag[b];
    return mb > smag[b - 1] && mb > smag[b - 2] && mb > smag[b + 1] && mb > smag[b + 2];
}

__global__ __launch_bounds__(kFftThreads) void pv_analyze_kernel(const AnalyzeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    const DevTables &tb = a.tb;
    const int N = tb.N, hs = tb.hs, nc = tb.nc, nt = blockDim.x;
    float2 *buf = reinterpret_cast<float2 *>(smem_raw); // [nc]
    float *smag = reinterpret_cast<float *>(buf + nc);  // [hs + 1]
    int *wcnt = reinterpret_cast<int *>(smag + hs + 4); // [(hs/nt + 1) * 4]
    int row, tl;
    if (!block_to_row_slice(a.Tn, a.rows, row, tl)) return;
    const int64_t t = a.t0 + tl;
    const int slot = ring_slot(a.s0, tl, a.TR);
    const int64_t a0 = t * (int64_t)a.hop;
    // row = s*Cch + c; the host guarantees stride_s == Cch*stride_c for contiguous rows
    const float *__restrict__ in = a.ia.in + (int64_t)row * a.ia.stride_c;
    const float *__restrict__ w = tb.window;

    // windowed, fft-shifted frame packed as nc complex numbers, written in butterfly (permuted) order
    for (int j = threadIdx.x; j < nc; j += nt) {
        const int src = tb.perm[j];
        const int k0 = (2 * src + hs) & (N - 1);
        const int64_t g0 = a0 + k0;
        const float x0 = g0 < a.ia.len ? in[(uint64_t)g0 & a.ia.mask] : 0.f;
        const float x1 = g0 + 1 < a.ia.len ? in[(uint64_t)(g0 + 1) & a.ia.mask] : 0.f;
        buf[j] = make_float2(x0 * w[k0], x1 * w[k0 + 1]);
    }
    __syncthreads();
    fft_stages<false>(buf, tb, tb.tw_fwd);

    // real-FFT split (kiss_fftr.c:91-120) + polar (FFT.cc:2623-2630)
    const int64_t plane = ((int64_t)row * a.TR + slot);
    float *__restrict__ mag = a.mag + plane * tb.HP;
    float *__restrict__ ph = a.phase + plane * tb.HP;
    for (int k = threadIdx.x; k <= nc / 2; k += nt) {
        if (k == 0) {
            const float2 tdc = buf[0];
            const float r0 = tdc.x + tdc.y, rn = tdc.x - tdc.y;
            const float m0 = sqrtf(r0 * r0 + 0.f * 0.f), mn = sqrtf(rn * rn + 0.f * 0.f);
            mag[0] = m0;
            smag[0] = m0;
            ph[0] = pv_atan2f_fd_finite(0.f, r0);
            mag[nc] = mn;
            smag[nc] = mn;
            ph[nc] = pv_atan2f_fd_finite(0.f, rn);
        } else {
            const float2 fpk = buf[k];
            const float2 q = buf[nc - k];
            const float2 fpnk = make_float2(q.x, -q.y);
            const float2 f1k = cadd(fpk, fpnk);
            const float2 f2k = csub(fpk, fpnk);
            const float2 tq = cmul(f2k, tb.st_fwd[k]);
            const float xr = (f1k.x + tq.x) * 0.5f, xi = (f1k.y + tq.y) * 0.5f;
            const float yr = (f1k.x - tq.x) * 0.5f, yi = (tq.y - f1k.y) * 0.5f;
            if (k != nc - k) {
                const float m = sqrtf(xr * xr + xi * xi);
                mag[k] = m;
                smag[k] = m;
                ph[k] = pv_atan2f_fd_finite(xi, xr);
            }
            const float m2 = sqrtf(yr * yr + yi * yi);
            mag[nc - k] = m2;
            smag[nc - k] = m2;
            ph[nc - k] = pv_atan2f_fd_finite(yi, yr);
        }
    }
    if (!a.find_peaks) return;
    __syncthreads();

    // ordered compaction of the peak bins: ballot per 64-bin group, prefix over groups in bin order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nt >> 6;
    const int J = (hs + nt - 1) / nt;
    for (int j = 0; j < J; ++j) {
        const unsigned long long bm = __ballot(is_peak(smag, threadIdx.x + j * nt, hs));
        if (lane == 0) wcnt[j * nw + wave] = __popcll(bm);
    }
    __syncthreads();
    uint16_t *__restrict__ pk = a.peaks + plane * a.PKP;
    int running = 0, qi = 0;
    for (int j = 0; j < J; ++j) {
        const int mine = j * nw + wave;
        for (; qi < mine; ++qi) running += wcnt[qi];
        const int b = threadIdx.x + j * nt;
        const bool isp = is_peak(smag, b, hs);
        const unsigned long long bm = __ballot(isp);
        if (isp) pk[running + __popcll(bm & ((1ull << lane) - 1ull))] = (uint16_t)b;
    }
    if (threadIdx.x == 0) {
        int total = 0;
        for (int q = 0; q < J * nw; ++q) total += wcnt[q];
        a.npk[plane] = total;
    }
}

// --------------------------------------------------------------------------------------------
// analysis, wave-per-frame variant (N = 2048 / 4096): one 64-lane wave owns a slice, the four waves of a
// workgroup take four consecutive slices of the same row.  No workgroup barrier anywhere.
// --------------------------------------------------------------------------------------------
template <int WPB> __device__ __forceinline__ bool block_to_row_slice_w(int Tn, int rows, int &row, int &tl) {
    const int groups = (Tn + WPB - 1) / WPB;
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    row = xcd + 8 * (q / groups);
    tl = __builtin_amdgcn_readfirstlane(WPB * (q % groups) + (threadIdx.x >> 6)); // wave-uniform: keep it scalar
    return row < rows && tl < Tn;
}

// (the body is a device function of (row, slice) so that the single-launch streaming kernel can call it too)
// ATAB: the LDS address of atan2f's interval table as a compile-time number -- the kernels that call this have no
// static LDS, so their dynamic LDS starts at address 0 (checked on the host: lds_starts_at_zero) and the table's
// offsets fold into the LDS instructions' immediate fields (two vector adds fewer per bin)
template <int NC, uint32_t ATAB>
__device__ __forceinline__ void analyze_wave_role(const AnalyzeArgs &a, const int row, const int tl, cf *lds) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const unsigned char *atab = reinterpret_cast<const unsigned char *>((uintptr_t)ATAB);
    using W = WF<NC>;
    constexpr int N = 2 * NC, hs = NC, R = W::R;
    const int lane = threadIdx.x & 63;
    // atan2f's interval table (pv_atan2f.h): 256 bytes of LDS behind the waves' regions, (re)written by every wave
    // with the same words before its polar conversion reads them
    const uint32_t atab_word = pv_atan_blob_dev.w[lane];
    const DevTables &tb = a.tb;
    const int64_t t = a.t0 + tl;
    const int slot = ring_slot(a.s0, tl, a.TR);
    const int64_t a0 = t * (int64_t)a.hop;
    const float *__restrict__ in = a.ia.in + (int64_t)row * a.ia.stride_c;
    const float *__restrict__ w = tb.window;
    const cf *__restrict__ tw = reinterpret_cast<const cf *>(tb.tw_fwd);
    const cf *__restrict__ stw = reinterpret_cast<const cf *>(tb.st_fwd);
    const cf2 *__restrict__ twl = reinterpret_cast<const cf2 *>(tb.twl_fwd);

    // Every global load below is issued ahead of its use, as early as the 128-VGPR budget of four waves per SIMD
    // allows (a pass's twiddles while the previous pass finishes, the split's super-twiddles during the last pass,
    // all of them before the first store): a load placed at its use costs the wave a full memory round trip, and one
    // placed after a store also waits for that store to be acknowledged (vmcnt is a single in-order counter).
    WfTw<W> T0, T1, T2;
    wf_load_pass_tw<W, 0>(T0, lane, tw);
    // windowed, fft-shifted frame straight into the pass-0 register layout: for a fixed register the 64
    // lanes read one contiguous 512-byte span of the input (permuted among the lanes)
    cf v[R];
    {
        const int lp = wf_lane_part<W>(0, lane);
        const int lsrc = wf_src_of<W>(lp);
        const uint64_t m0 = (uint64_t)a0 & a.ia.mask;
        if (a0 + N + 4 <= a.ia.len && ((m0 + (uint64_t)(N + 3)) & a.ia.mask) == m0 + (uint64_t)(N + 3)) {
            // The whole frame (and the 16 bytes after it) lies inside the input and does not wrap the ring
            // (wave-uniform, the common case).  Aligned 16-byte loads of the span that contains the frame, times
            // the window copy delayed by the frame's offset d into its first 16 bytes, staged through LDS: nine
            // wide loads of each instead of 48 narrow permuted ones (the texture path is issue-bound), then the
            // pass-0 layout is a pair of adjacent floats per register out of LDS.
            const float *__restrict__ fp = in + m0;
            const uintptr_t fa = reinterpret_cast<uintptr_t>(fp);
            const int d = (int)((fa & 15u) >> 2);
            const float4 *__restrict__ xb = reinterpret_cast<const float4 *>(fa & ~(uintptr_t)15);
            const float4 *__restrict__ wb = reinterpret_cast<const float4 *>(tb.window_sh + (size_t)d * (N + 8));
            constexpr int Q = N / 256; // 16-byte pieces per lane (plus one more on lane 0), eight at a time
            float4 *stage4 = reinterpret_cast<float4 *>(lds);
            float4 xt = make_float4(0.f, 0.f, 0.f, 0.f), wt = xt;
            if (lane == 0) {
                xt = xb[64 * Q];
                wt = wb[64 * Q];
            }
            constexpr int QS = Q < 8 ? Q : 8;
#pragma unroll
            for (int q0 = 0; q0 < Q; q0 += QS) {
                float4 xq[QS], wq[QS];
#pragma unroll
                for (int q = 0; q < QS; ++q) {
                    xq[q] = xb[lane + 64 * (q0 + q)];
                    wq[q] = wb[lane + 64 * (q0 + q)];
                }
#pragma unroll
                for (int q = 0; q < QS; ++q)
                    stage4[lane + 64 * (q0 + q)] =
                        make_float4(xq[q].x * wq[q].x, xq[q].y * wq[q].y, xq[q].z * wq[q].z, xq[q].w * wq[q].w);
            }
            if (lane == 0) stage4[64 * Q] = make_float4(xt.x * wt.x, xt.y * wt.y, xt.z * wt.z, xt.w * wt.w);
            wave_sync();
            const float *stage = reinterpret_cast<const float *>(lds) + d; // stage[k] = x[a0 + k] * w[k]
            const int kl = (2 * lsrc + hs) & (N - 1); // the register part only sets bits the lane part leaves clear
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int kr = 2 * wf_src_of_const<W>(wf_reg_part<W>(0, r));
                const int k0 = kl ^ kr; // == (2 * (lsrc | src_r) + hs) & (N - 1)
                v[r] = cf{stage[k0], stage[k0 + 1]};
            }
            wave_sync();
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int src = lsrc | wf_src_of_const<W>(wf_reg_part<W>(0, r));
                const int k0 = (2 * src + hs) & (N - 1);
                const int64_t g0 = a0 + k0;
                const float x0 = g0 < a.ia.len ? in[(uint64_t)g0 & a.ia.mask] : 0.f;
                const float x1 = g0 + 1 < a.ia.len ? in[(uint64_t)(g0 + 1) & a.ia.mask] : 0.f;
                v[r] = cf{x0 * w[k0], x1 * w[k0 + 1]};
            }
        }
    }
    constexpr int J = NC / 128;
    const int lp0 = wf_lane_part<W>(0, lane), lp1 = wf_lane_part<W>(1, lane), lp2 = wf_lane_part<W>(2, lane);
    WfTwRaw<W, 1> raw1;
    // (PV_EXP_ANA: elimination builds for timing only, results invalid -- bit 0 no butterflies, 1 no LDS exchanges
    // between the passes, 2 no polar conversion, 3 no plane stores, 4 no peak search)
#if defined(PV_EXP_ANA) && (PV_EXP_ANA & 1)
#define PV_ANA_BFLY(x)
#else
#define PV_ANA_BFLY(x) x
#endif
#if defined(PV_EXP_ANA) && (PV_EXP_ANA & 2)
#define PV_ANA_XCHG(x)
#else
#define PV_ANA_XCHG(x) x
#endif
    wf_fetch_pass_tw<W, 1>(raw1, lane, twl);
    PV_ANA_BFLY((wf_apply_pass_stages<W, 0, false>(v, T0)));
    PV_ANA_XCHG((wf_store<W, 0>(lds, v, lp0)));
    wave_sync();
    PV_ANA_XCHG((wf_load<W, 1>(lds, v, lp1)));
    wf_unpack_pass_tw<W, 1>(T1, raw1);
    PV_ANA_BFLY((wf_apply_pass_stages<W, 1, false>(v, T1)));
    WfTwRaw<W, 2> raw2;
    wf_fetch_pass_tw<W, 2>(raw2, lane, twl);
    PV_ANA_XCHG((wf_store<W, 1>(lds, v, lp1)));
    wave_sync();
    PV_ANA_XCHG((wf_load<W, 2>(lds, v, lp2)));
    wf_unpack_pass_tw<W, 2>(T2, raw2);
    PV_ANA_BFLY((wf_apply_pass_stages<W, 2, false>(v, T2)));
    if constexpr (W::NPASS == 4) { // (512 points: the last radix-4 stage is a pass of its own)
        const int lp3 = wf_lane_part<W>(3, lane);
        WfTwRaw<W, 3> raw3;
        wf_fetch_pass_tw<W, 3>(raw3, lane, twl);
        PV_ANA_XCHG((wf_store<W, 2>(lds, v, lp2)));
        wave_sync();
        PV_ANA_XCHG((wf_load<W, 3>(lds, v, lp3)));
        wf_unpack_pass_tw<W, 3>(T2, raw3);
        PV_ANA_BFLY((wf_apply_pass_stages<W, 3, false>(v, T2)));
    }
    cf sw[J];
#pragma unroll
    for (int j = 0; j < J; ++j) sw[j] = stw[lane + 64 * j];
    const cf swmid = stw[NC / 2];
    if constexpr (W::NPASS == 4) wf_store<W, 3>(lds, v, wf_lane_part<W>(3, lane));
    else wf_store<W, 2>(lds, v, lp2);
    wave_sync();

    // real-FFT split (kiss_fftr.c:91-120) + polar (FFT.cc:2623-2630); lane handles k = lane + 64 j and NC - k
    const int64_t plane = ((int64_t)row * a.TR + slot);
    float *__restrict__ mag = a.mag + plane * tb.HP;
    float *__restrict__ ph = a.phase + plane * tb.HP;
    // The split's output bins, cartesian, go back to the wave's LDS region in natural order; the polar conversion
    // then runs as a ROLLED loop over runs of four consecutive bins.  (Unrolled over a lane's 2 J bins it was 2 J
    // inlined copies of atan2f -- libm's algorithm, pv_atan2f.h, ~100 instructions each -- and the 4096-point kernel
    // ran 2.3x slower than with the device library's shorter atan2f; a run of four per iteration keeps four
    // independent evaluations in flight and the phases / magnitudes leave as 16-byte stores.)
    cf xlo[J], xhi[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = lane + 64 * j;
        if (j == 0 && lane == 0) {
            const cf tdc = lds[W::pad(0)];
            xlo[0] = cf{tdc.x + tdc.y, 0.f}; // DC and Nyquist: imaginary part forced to 0 (kiss_fftr.c:97-102)
            xhi[0] = cf{tdc.x - tdc.y, 0.f};
        } else {
            const cf fpk = lds[W::pad(k)];
            const cf q = lds[W::pad(NC - k)];
            const cf fpnk = cf{q.x, -q.y};
            const cf f1k = wf_add(fpk, fpnk);
            const cf f2k = wf_sub(fpk, fpnk);
            const cf tq = wf_cmul(f2k, sw[j]);
            xlo[j] = cf{(f1k.x + tq.x) * 0.5f, (f1k.y + tq.y) * 0.5f};
            xhi[j] = cf{(f1k.x - tq.x) * 0.5f, (tq.y - f1k.y) * 0.5f};
        }
    }
    cf xmid = cf{0.f, 0.f};
    if (lane == 0) { // k == NC/2 pairs with itself: the second assignment of the reference loop wins
        const cf fpk = lds[W::pad(NC / 2)];
        const cf fpnk = cf{fpk.x, -fpk.y};
        const cf f1k = wf_add(fpk, fpnk);
        const cf f2k = wf_sub(fpk, fpnk);
        const cf tq = wf_cmul(f2k, swmid);
        xmid = cf{(f1k.x - tq.x) * 0.5f, (tq.y - f1k.y) * 0.5f};
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = lane + 64 * j;
        if (j == 0 && lane == 0) {
            lds[0] = xlo[0];
            lds[NC] = xhi[0];
        } else {
            lds[k] = xlo[j];
            lds[NC - k] = xhi[j];
        }
    }
    if (lane == 0) lds[NC / 2] = xmid;
    wave_sync();
    // polar (FFT.cc:2623-2630): mag = sqrtf(re^2 + im^2), phase = atan2f(im, re).  The magnitudes also stay in LDS
    // for the peak search: a run's four floats land on the first half of the 32 bytes its cartesian values
    // occupied, and every later run lies beyond them.
    float *smag = reinterpret_cast<float *>(lds);
    *(lds_u32 *)(uintptr_t)(ATAB + 4u * (uint32_t)lane) = atab_word;
    // DC and Nyquist have a zero imaginary part by construction (kiss_fftr.c:97-102): atan2f(+0, r) is +0 or pi by
    // r's sign bit and the magnitude sqrtf(r r + 0 0); lanes 0 and 1 do both at once, and the loop below sees (1, 1)
    // in the DC slot, so that its operand-range test is not tripped by that zero in every frame.
    float edge_mag = 0.f, edge_ph = 0.f;
    if (lane < 2) {
        const cf e = lds[lane == 0 ? 0 : NC];
        edge_mag = sqrtf(e.x * e.x + 0.f * 0.f);
        edge_ph = pv_u2f((uint32_t)((int32_t)pv_f2u(e.x) >> 31) & pv_f2u(3.1415927410e+00f));
    }
    wave_sync();
    if (lane == 0) lds[0] = cf{1.f, 1.f};
    wave_sync();
#pragma nounroll
    for (int q = 0; q < NC / 256; ++q) {
        const int i4 = 4 * (lane + 64 * q);
        const float4 c01 = *reinterpret_cast<const float4 *>(lds + i4);     // bins i4, i4 + 1
        const float4 c23 = *reinterpret_cast<const float4 *>(lds + i4 + 2); // bins i4 + 2, i4 + 3
        float4 m4, p4;
        const float a0 = c01.x * c01.x + c01.y * c01.y, a1 = c01.z * c01.z + c01.w * c01.w;
        const float a2 = c23.x * c23.x + c23.y * c23.y, a3 = c23.z * c23.z + c23.w * c23.w;
        // The short division and square root of pv_atan2f.h need normal operands with magnitudes in [2^-48, 2^63):
        // tested once per run of four bins and wave (true for spectra of audio unless a value is exactly zero --
        // digital silence -- or vanishingly small; the IEEE operations of the other branch serve those).
        const float mx = __builtin_fmaxf(__builtin_fmaxf(pv_max3_abs(c01.x, c01.y, c01.z), pv_max3_abs(c01.w, c23.x, c23.y)),
                                         __builtin_fmaxf(__builtin_fabsf(c23.z), __builtin_fabsf(c23.w)));
        const float mn = __builtin_fminf(__builtin_fminf(pv_min3_abs(c01.x, c01.y, c01.z), pv_min3_abs(c01.w, c23.x, c23.y)),
                                         __builtin_fminf(__builtin_fabsf(c23.z), __builtin_fabsf(c23.w)));
        const bool in_range = mx < 0x1p63f && mn >= 0x1p-48f;
#if defined(PV_EXP_ANA) && (PV_EXP_ANA & 4)
        if (true) {
            m4 = make_float4(a0, a1, a2, a3);
            p4 = make_float4(c01.y + mx, c01.w + mn, c23.y, c23.w);
        } else
#endif
        if (__builtin_amdgcn_ballot_w64(!in_range) == 0) {
            m4 = make_float4(pv_sqrt_safe(a0), pv_sqrt_safe(a1), pv_sqrt_safe(a2), pv_sqrt_safe(a3));
            p4.x = pv_atan2f_fd_tab<true>(c01.y, c01.x, atab);
            p4.y = pv_atan2f_fd_tab<true>(c01.w, c01.z, atab);
            p4.z = pv_atan2f_fd_tab<true>(c23.y, c23.x, atab);
            p4.w = pv_atan2f_fd_tab<true>(c23.w, c23.z, atab);
        } else {
            m4 = make_float4(sqrtf(a0), sqrtf(a1), sqrtf(a2), sqrtf(a3));
            p4.x = pv_atan2f_fd_tab<false>(c01.y, c01.x, atab);
            p4.y = pv_atan2f_fd_tab<false>(c01.w, c01.z, atab);
            p4.z = pv_atan2f_fd_tab<false>(c23.y, c23.x, atab);
            p4.w = pv_atan2f_fd_tab<false>(c23.w, c23.z, atab);
        }
        if (q == 0 && lane == 0) p4.x = edge_ph, m4.x = edge_mag;
#if !(defined(PV_EXP_ANA) && (PV_EXP_ANA & 8))
        *reinterpret_cast<float4 *>(ph + i4) = p4;
        *reinterpret_cast<float4 *>(mag + i4) = m4;
#else
        if (p4.x == 12345.678f) *reinterpret_cast<float4 *>(ph + i4) = p4; // (keeps the values alive)
#endif
        *reinterpret_cast<float4 *>(smag + i4) = m4;
    }
    if (lane == 1) {
        ph[NC] = edge_ph;
        mag[NC] = edge_mag;
        smag[NC] = edge_mag;
    }
    wave_sync();
#if defined(PV_EXP_ANA) && (PV_EXP_ANA & 16)
    if (lane == 0) a.npk[plane] = 100;
    return;
#endif
    if (!a.find_peaks) return;
    // ordered list first into LDS (behind the magnitudes), then out in whole 32-bit words: three coalesced
    // stores per lane instead of one sparsely populated 16-bit store per group of 64 bins
    uint16_t *slist = reinterpret_cast<uint16_t *>(smag + NC + 4); // [PKP], 16-byte aligned
    // A lane tests four consecutive bins (one 16-byte read plus the two bins either side) against the larger of their
    // four neighbours -- is_peak()'s four comparisons as one.  Two peaks are at least three bins apart, so a run of
    // four holds none, one, or its first and last bin: the ordered list position is a prefix count over two lane masks.
    // Bins 0, 1 and hs - 2, hs - 1 are never peaks (:587-596): their outer neighbours read as +infinity.
    int running = 0;
    const float inf = __builtin_inff();
#pragma unroll
    for (int q = 0; q < NC / 256; ++q) {
        const int i4 = 4 * (lane + 64 * q);
        const float4 m = *reinterpret_cast<const float4 *>(smag + i4);
        float2 lo = *reinterpret_cast<const float2 *>(smag + (q == 0 && lane == 0 ? 0 : i4 - 2));
        float2 hi = *reinterpret_cast<const float2 *>(smag + i4 + 4);
        if (q == 0 && lane == 0) lo = make_float2(inf, inf);
        if (q == NC / 256 - 1 && lane == 63) hi = make_float2(inf, inf);
        const float in01 = __builtin_fmaxf(m.x, m.y), in12 = __builtin_fmaxf(m.y, m.z), in23 = __builtin_fmaxf(m.z, m.w);
        const bool f0 = m.x > __builtin_fmaxf(__builtin_fmaxf(lo.x, lo.y), in12);
        const bool f1 = m.y > __builtin_fmaxf(__builtin_fmaxf(lo.y, m.x), in23);
        const bool f2 = m.z > __builtin_fmaxf(in01, __builtin_fmaxf(m.w, hi.x));
        const bool f3 = m.w > __builtin_fmaxf(in12, __builtin_fmaxf(hi.x, hi.y));
        const bool any = f0 || f1 || f2 || f3, two = f0 && f3;
        const unsigned long long ma = __ballot(any), mb = __ballot(two);
        const int pos = running + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u)) +
                        (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
        if (any) slist[pos] = (uint16_t)(i4 + (f0 ? 0 : f1 ? 1 : f2 ? 2 : 3));
        if (two) slist[pos + 1] = (uint16_t)(i4 + 3);
        running += __popcll(ma) + __popcll(mb);
    }
    if (lane == 0) {
        a.npk[plane] = running;
        if (running & 1) slist[running] = 0; // the odd tail shares a word with the last peak
    }
    wave_sync();
    uint32_t *__restrict__ pk32 = reinterpret_cast<uint32_t *>(a.peaks + plane * a.PKP); // PKP % 8 == 0
    const uint32_t *slist32 = reinterpret_cast<const uint32_t *>(slist);
    for (int i = lane; 2 * i < running; i += 64) pk32[i] = slist32[i];
}

// workgroup barrier for hand-overs through LDS: waits for this wave's LDS operations only (a __syncthreads() would also
// wait for its outstanding global stores to be acknowledged)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The same analysis with a frame on TWO waves (WF2048S: 4096-point frames, 128 lanes x 16 elements): the arithmetic,
// its order and the outputs are those of analyze_wave_role<2048>; what changes is who holds what.  A lane needs the
// registers of the 2048-point frame's kernel (four waves per SIMD instead of two), the passes hand over through workgroup
// barriers, and the peak list's positions need the other wave's counts.  The polar loop overwrites the cartesian bins it
// has consumed with magnitudes, so each of its iterations separates its reads from its writes by a barrier.
template <class W, uint32_t ATAB>
__device__ __forceinline__ void analyze_split_role(const AnalyzeArgs &a, const int row, const int tl, cf *lds) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const unsigned char *atab = reinterpret_cast<const unsigned char *>((uintptr_t)ATAB);
    constexpr int NC = W::N_C, N = 2 * NC, hs = NC, R = W::R, LN = W::LANES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t atab_word = pv_atan_blob_dev.w[lane];
    const DevTables &tb = a.tb;
    const int64_t t = a.t0 + tl;
    const int slot = ring_slot(a.s0, tl, a.TR);
    const int64_t a0 = t * (int64_t)a.hop;
    const float *__restrict__ in = a.ia.in + (int64_t)row * a.ia.stride_c;
    const float *__restrict__ w = tb.window;
    const cf *__restrict__ tw = reinterpret_cast<const cf *>(tb.tw_fwd);
    const cf *__restrict__ stw = reinterpret_cast<const cf *>(tb.st_fwd);
    const cf2 *__restrict__ twl = reinterpret_cast<const cf2 *>(tb.twl_fwd);

    WfTw<W> T0, T1, T2;
    wf_load_pass_tw<W, 0>(T0, tid, tw);
    cf v[R];
    {
        const int lp = wf_lane_part<W>(0, tid);
        const int lsrc = wf_src_of<W>(lp);
        const uint64_t m0 = (uint64_t)a0 & a.ia.mask;
        if (a0 + N + 4 <= a.ia.len && ((m0 + (uint64_t)(N + 3)) & a.ia.mask) == m0 + (uint64_t)(N + 3)) {
            // (workgroup-uniform) aligned 16-byte pieces of the span that holds the frame times the window copy delayed
            // by the frame's offset into its first piece, staged through LDS (analyze_wave_role)
            const float *__restrict__ fp = in + m0;
            const uintptr_t fa = reinterpret_cast<uintptr_t>(fp);
            const int d = (int)((fa & 15u) >> 2);
            const float4 *__restrict__ xb = reinterpret_cast<const float4 *>(fa & ~(uintptr_t)15);
            const float4 *__restrict__ wb = reinterpret_cast<const float4 *>(tb.window_sh + (size_t)d * (N + 8));
            constexpr int Q = N / (4 * LN); // 16-byte pieces per lane (plus one more on thread 0)
            float4 *stage4 = reinterpret_cast<float4 *>(lds);
            float4 xt = make_float4(0.f, 0.f, 0.f, 0.f), wt = xt;
            if (tid == 0) {
                xt = xb[LN * Q];
                wt = wb[LN * Q];
            }
            float4 xq[Q], wq[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                xq[q] = xb[tid + LN * q];
                wq[q] = wb[tid + LN * q];
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
                stage4[tid + LN * q] = make_float4(xq[q].x * wq[q].x, xq[q].y * wq[q].y, xq[q].z * wq[q].z, xq[q].w * wq[q].w);
            if (tid == 0) stage4[LN * Q] = make_float4(xt.x * wt.x, xt.y * wt.y, xt.z * wt.z, xt.w * wt.w);
            lds_barrier();
            const float *stage = reinterpret_cast<const float *>(lds) + d; // stage[k] = x[a0 + k] * w[k]
            const int kl = (2 * lsrc + hs) & (N - 1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int kr = 2 * wf_src_of_const<W>(wf_reg_part<W>(0, r));
                const int k0 = kl ^ kr;
                v[r] = cf{stage[k0], stage[k0 + 1]};
            }
            lds_barrier(); // (pass 0 stores over the staged frame)
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int src = lsrc | wf_src_of_const<W>(wf_reg_part<W>(0, r));
                const int k0 = (2 * src + hs) & (N - 1);
                const int64_t g0 = a0 + k0;
                const float x0 = g0 < a.ia.len ? in[(uint64_t)g0 & a.ia.mask] : 0.f;
                const float x1 = g0 + 1 < a.ia.len ? in[(uint64_t)(g0 + 1) & a.ia.mask] : 0.f;
                v[r] = cf{x0 * w[k0], x1 * w[k0 + 1]};
            }
        }
    }
    constexpr int J = NC / (2 * LN);
    const int lp0 = wf_lane_part<W>(0, tid), lp1 = wf_lane_part<W>(1, tid), lp2 = wf_lane_part<W>(2, tid);
    // (a pass loads exactly the slots it stores, so one barrier per hand-over is enough)
    WfTwRaw<W, 1> raw1;
    wf_fetch_pass_tw<W, 1>(raw1, tid, twl);
    wf_apply_pass_stages<W, 0, false>(v, T0);
    wf_store<W, 0>(lds, v, lp0);
    lds_barrier();
    wf_load<W, 1>(lds, v, lp1);
    wf_unpack_pass_tw<W, 1>(T1, raw1);
    wf_apply_pass_stages<W, 1, false>(v, T1);
    WfTwRaw<W, 2> raw2;
    wf_fetch_pass_tw<W, 2>(raw2, tid, twl);
    wf_store<W, 1>(lds, v, lp1);
    lds_barrier();
    wf_load<W, 2>(lds, v, lp2);
    wf_unpack_pass_tw<W, 2>(T2, raw2);
    wf_apply_pass_stages<W, 2, false>(v, T2);
    cf sw[J];
#pragma unroll
    for (int j = 0; j < J; ++j) sw[j] = stw[tid + LN * j];
    const cf swmid = stw[NC / 2];
    wf_store<W, 2>(lds, v, lp2);
    lds_barrier();

    // real-FFT split (kiss_fftr.c:91-120): thread handles k = tid + LN j and NC - k
    const int64_t plane = ((int64_t)row * a.TR + slot);
    float *__restrict__ mag = a.mag + plane * tb.HP;
    float *__restrict__ ph = a.phase + plane * tb.HP;
    cf xlo[J], xhi[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = tid + LN * j;
        if (j == 0 && tid == 0) {
            const cf tdc = lds[W::pad(0)];
            xlo[0] = cf{tdc.x + tdc.y, 0.f};
            xhi[0] = cf{tdc.x - tdc.y, 0.f};
        } else {
            const cf fpk = lds[W::pad(k)];
            const cf q = lds[W::pad(NC - k)];
            const cf fpnk = cf{q.x, -q.y};
            const cf f1k = wf_add(fpk, fpnk);
            const cf f2k = wf_sub(fpk, fpnk);
            const cf tq = wf_cmul(f2k, sw[j]);
            xlo[j] = cf{(f1k.x + tq.x) * 0.5f, (f1k.y + tq.y) * 0.5f};
            xhi[j] = cf{(f1k.x - tq.x) * 0.5f, (tq.y - f1k.y) * 0.5f};
        }
    }
    cf xmid = cf{0.f, 0.f};
    if (tid == 0) {
        const cf fpk = lds[W::pad(NC / 2)];
        const cf fpnk = cf{fpk.x, -fpk.y};
        const cf f1k = wf_add(fpk, fpnk);
        const cf f2k = wf_sub(fpk, fpnk);
        const cf tq = wf_cmul(f2k, swmid);
        xmid = cf{(f1k.x - tq.x) * 0.5f, (tq.y - f1k.y) * 0.5f};
    }
    lds_barrier();
    // DC and Nyquist (zero imaginary part by construction, kiss_fftr.c:97-102: atan2f(+0, r) is +0 or pi by r's sign bit,
    // the magnitude sqrtf(r r + 0 0)) are thread 0's own values; the loop below sees (1, 1) in the DC slot, so that its
    // operand-range test is not tripped by that zero in every frame
    float dc_mag = 0.f, dc_ph = 0.f, ny_mag = 0.f, ny_ph = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = tid + LN * j;
        if (j == 0 && tid == 0) {
            dc_mag = sqrtf(xlo[0].x * xlo[0].x + 0.f * 0.f);
            dc_ph = pv_u2f((uint32_t)((int32_t)pv_f2u(xlo[0].x) >> 31) & pv_f2u(3.1415927410e+00f));
            ny_mag = sqrtf(xhi[0].x * xhi[0].x + 0.f * 0.f);
            ny_ph = pv_u2f((uint32_t)((int32_t)pv_f2u(xhi[0].x) >> 31) & pv_f2u(3.1415927410e+00f));
            lds[0] = cf{1.f, 1.f};
        } else {
            lds[k] = xlo[j];
            lds[NC - k] = xhi[j];
        }
    }
    if (tid == 0) lds[NC / 2] = xmid;
    float *smag = reinterpret_cast<float *>(lds);
    *(lds_u32 *)(uintptr_t)(ATAB + 4u * (uint32_t)lane) = atab_word; // (both waves, the same words)
    lds_barrier();
    // Polar conversion, four runs of four bins per thread.  Iteration q writes its magnitudes over the cartesian values of
    // runs [64 q, 64 q + 64), which iterations 0 and 1 read -- the other wave's threads among them -- and the reads of
    // those two iterations are issued before the one barrier below; the runs of iterations 2 and 3 are never overwritten.
    constexpr int NQ = NC / (4 * LN);
    static_assert(NQ <= 4, "the hazard argument above");
    float4 nx01 = *reinterpret_cast<const float4 *>(lds + 4 * tid), nx23 = *reinterpret_cast<const float4 *>(lds + 4 * tid + 2);
    float4 nn01 = *reinterpret_cast<const float4 *>(lds + 4 * (tid + LN)), nn23 = *reinterpret_cast<const float4 *>(lds + 4 * (tid + LN) + 2);
    lds_barrier();
#pragma nounroll
    for (int q = 0; q < NQ; ++q) {
        const int i4 = 4 * (tid + LN * q);
        const float4 c01 = nx01, c23 = nx23;
        nx01 = nn01, nx23 = nn23;
        if (q + 2 < NQ) {
            nn01 = *reinterpret_cast<const float4 *>(lds + i4 + 8 * LN);
            nn23 = *reinterpret_cast<const float4 *>(lds + i4 + 8 * LN + 2);
        }
        float4 m4, p4;
        const float a0 = c01.x * c01.x + c01.y * c01.y, a1 = c01.z * c01.z + c01.w * c01.w;
        const float a2 = c23.x * c23.x + c23.y * c23.y, a3 = c23.z * c23.z + c23.w * c23.w;
        const float mx = __builtin_fmaxf(__builtin_fmaxf(pv_max3_abs(c01.x, c01.y, c01.z), pv_max3_abs(c01.w, c23.x, c23.y)),
                                         __builtin_fmaxf(__builtin_fabsf(c23.z), __builtin_fabsf(c23.w)));
        const float mn = __builtin_fminf(__builtin_fminf(pv_min3_abs(c01.x, c01.y, c01.z), pv_min3_abs(c01.w, c23.x, c23.y)),
                                         __builtin_fminf(__builtin_fabsf(c23.z), __builtin_fabsf(c23.w)));
        const bool in_range = mx < 0x1p63f && mn >= 0x1p-48f;
        if (__builtin_amdgcn_ballot_w64(!in_range) == 0) {
            m4 = make_float4(pv_sqrt_safe(a0), pv_sqrt_safe(a1), pv_sqrt_safe(a2), pv_sqrt_safe(a3));
            p4.x = pv_atan2f_fd_tab<true>(c01.y, c01.x, atab);
            p4.y = pv_atan2f_fd_tab<true>(c01.w, c01.z, atab);
            p4.z = pv_atan2f_fd_tab<true>(c23.y, c23.x, atab);
            p4.w = pv_atan2f_fd_tab<true>(c23.w, c23.z, atab);
        } else {
            m4 = make_float4(sqrtf(a0), sqrtf(a1), sqrtf(a2), sqrtf(a3));
            p4.x = pv_atan2f_fd_tab<false>(c01.y, c01.x, atab);
            p4.y = pv_atan2f_fd_tab<false>(c01.w, c01.z, atab);
            p4.z = pv_atan2f_fd_tab<false>(c23.y, c23.x, atab);
            p4.w = pv_atan2f_fd_tab<false>(c23.w, c23.z, atab);
        }
        if (q == 0 && tid == 0) p4.x = dc_ph, m4.x = dc_mag;
        *reinterpret_cast<float4 *>(ph + i4) = p4;
        *reinterpret_cast<float4 *>(mag + i4) = m4;
        *reinterpret_cast<float4 *>(smag + i4) = m4;
    }
    if (tid == 0) {
        ph[NC] = ny_ph;
        mag[NC] = ny_mag;
        smag[NC] = ny_mag; // (over cartesian bin NC / 2, which only this thread read, two iterations ago)
    }
    lds_barrier();
    if (!a.find_peaks) return; // (workgroup-uniform)
    // peak list: flags and per-wave counts first, positions once both waves' counts are known (runs in ascending bin
    // order are (q, wave 0), (q, wave 1), (q + 1, wave 0), ...)
    uint16_t *slist = reinterpret_cast<uint16_t *>(smag + NC + 4); // [PKP], 16-byte aligned
    int *scnt = reinterpret_cast<int *>(reinterpret_cast<char *>(lds) + ATAB + 4u * PV_ATAN_BLOB_WORDS); // [2][NQ] behind atan2f's table (lds is the workgroup's LDS base)
    const float inf = __builtin_inff();
    int fl[NQ], pre[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i4 = 4 * (tid + LN * q);
        const float4 m = *reinterpret_cast<const float4 *>(smag + i4);
        float2 lo = *reinterpret_cast<const float2 *>(smag + (q == 0 && tid == 0 ? 0 : i4 - 2));
        float2 hi = *reinterpret_cast<const float2 *>(smag + i4 + 4);
        if (q == 0 && tid == 0) lo = make_float2(inf, inf);
        if (q == NQ - 1 && tid == LN - 1) hi = make_float2(inf, inf);
        const float in01 = __builtin_fmaxf(m.x, m.y), in12 = __builtin_fmaxf(m.y, m.z), in23 = __builtin_fmaxf(m.z, m.w);
        const bool f0 = m.x > __builtin_fmaxf(__builtin_fmaxf(lo.x, lo.y), in12);
        const bool f1 = m.y > __builtin_fmaxf(__builtin_fmaxf(lo.y, m.x), in23);
        const bool f2 = m.z > __builtin_fmaxf(in01, __builtin_fmaxf(m.w, hi.x));
        const bool f3 = m.w > __builtin_fmaxf(in12, __builtin_fmaxf(hi.x, hi.y));
        const bool any = f0 || f1 || f2 || f3, two = f0 && f3;
        const unsigned long long ma = __ballot(any), mb = __ballot(two);
        pre[q] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u)) +
                 (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
        fl[q] = (any ? 4 : 0) | (two ? 8 : 0) | (f0 ? 0 : f1 ? 1 : f2 ? 2 : 3);
        if (lane == 0) scnt[wv * NQ + q] = __popcll(ma) + __popcll(mb);
    }
    lds_barrier();
    int running = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c0 = scnt[q], c1 = scnt[NQ + q];
        const int pos = running + (wv ? c0 : 0) + pre[q];
        const int i4 = 4 * (tid + LN * q);
        if (fl[q] & 4) slist[pos] = (uint16_t)(i4 + (fl[q] & 3));
        if (fl[q] & 8) slist[pos + 1] = (uint16_t)(i4 + 3);
        running += c0 + c1;
    }
    if (tid == 0) {
        a.npk[plane] = running;
        if (running & 1) slist[running] = 0; // the odd tail shares a word with the last peak
    }
    lds_barrier();
    uint32_t *__restrict__ pk32 = reinterpret_cast<uint32_t *>(a.peaks + plane * a.PKP); // PKP % 8 == 0
    const uint32_t *slist32 = reinterpret_cast<const uint32_t *>(slist);
    for (int i = tid; 2 * i < running; i += LN) pk32[i] = slist32[i];
}

// one frame per workgroup of two waves
__global__ __launch_bounds__(128) void pv_analyze_split_kernel(const AnalyzeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    int row, tl;
    if (!block_to_row_slice(a.Tn, a.rows, row, tl)) return; // workgroup-uniform
    analyze_split_role<WF2048S, WF2048S::LDS_CF * sizeof(cf)>(a, row, tl, reinterpret_cast<cf *>(smem_raw));
}

// (the 4096-point variant needs 232 VGPRs, two waves per SIMD; forced to three -- 168 registers, 88 spilled -- it is
// slower: 1.05 vs 0.91 ms per 64 K slices)
template <int NC, int WPB> __global__ __launch_bounds__(64 * WPB) void pv_analyze_wave_kernel(const AnalyzeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    cf *lds = reinterpret_cast<cf *>(smem_raw) + (threadIdx.x >> 6) * WF<NC>::LDS_CF;
    int row, tl;
    if (!block_to_row_slice_w<WPB>(a.Tn, a.rows, row, tl)) return; // wave-uniform
    analyze_wave_role<NC, WPB * WF<NC>::LDS_CF * sizeof(cf)>(a, row, tl, lds);
}

// Kernels may need more than the default 64 KiB of dynamic LDS.  hipFuncSetAttribute applies to the device that is
// current at the call, so the "done" mark is kept per kernel AND device (a bit per device index; atomics because two
// host threads may launch for the first time together -- setting the attribute twice is harmless).
template <typename K> static void allow_big_lds_dev(K kernel, unsigned long long &done_mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024 - 512);
    __atomic_fetch_or(&done_mask, bit, __ATOMIC_RELEASE);
}

void launch_analyze(const AnalyzeArgs &a, hipStream_t st) {
    if (a.tb.nc == 512 || a.tb.nc == 256) { // fft 1024 / 512: four passes (pv_wavefft.h WF<512>, WF<256>)
        constexpr int WPB = 1;
        const int grid = 8 * ((a.rows + 7) / 8) * ((a.Tn + WPB - 1) / WPB);
        if (a.tb.nc == 512)
            hipLaunchKernelGGL((pv_analyze_wave_kernel<512, WPB>), dim3(grid), dim3(64 * WPB),
                               WPB * WF<512>::LDS_CF * sizeof(cf) + 4 * PV_ATAN_BLOB_WORDS, st, a);
        else
            hipLaunchKernelGGL((pv_analyze_wave_kernel<256, WPB>), dim3(grid), dim3(64 * WPB),
                               WPB * WF<256>::LDS_CF * sizeof(cf) + 4 * PV_ATAN_BLOB_WORDS, st, a);
        return;
    }
    if (a.tb.nc == 1024 || a.tb.nc == 2048) {
        if (a.tb.nc == 1024) {
            constexpr int WPB = 1; // one frame per workgroup: nothing couples the waves, and 8.6 KB of LDS each
            const int grid = 8 * ((a.rows + 7) / 8) * ((a.Tn + WPB - 1) / WPB);
            hipLaunchKernelGGL((pv_analyze_wave_kernel<1024, WPB>), dim3(grid), dim3(64 * WPB),
                               WPB * WF<1024>::LDS_CF * sizeof(cf) + 4 * PV_ATAN_BLOB_WORDS, st, a);
        } else {
            if (a.split) { // a frame on two waves (pv_analyze_split_kernel; tb.twl_fwd is WF2048S's table)
                const int grid = 8 * ((a.rows + 7) / 8) * a.Tn;
                hipLaunchKernelGGL(pv_analyze_split_kernel, dim3(grid), dim3(WF2048S::LANES),
                                   WF2048S::LDS_CF * sizeof(cf) + 4 * PV_ATAN_BLOB_WORDS + 64, st, a);
                return;
            }
            constexpr int WPB = 1;
            const int grid = 8 * ((a.rows + 7) / 8) * ((a.Tn + WPB - 1) / WPB);
            static unsigned long long big = 0;
            allow_big_lds_dev(pv_analyze_wave_kernel<2048, WPB>, big);
            hipLaunchKernelGGL((pv_analyze_wave_kernel<2048, WPB>), dim3(grid), dim3(64 * WPB),
                               WPB * WF<2048>::LDS_CF * sizeof(cf) + 4 * PV_ATAN_BLOB_WORDS, st, a);
        }
        return;
    }
    const int grid = 8 * ((a.rows + 7) / 8) * a.Tn;
    const int nt = generic_fft_threads(a.tb.nc);
    const int J = (a.tb.hs + nt - 1) / nt;
    const size_t lds = (size_t)a.tb.nc * sizeof(float2) + sizeof(float) * (a.tb.hs + 4) + sizeof(int) * (J + 1) * 4;
    hipLaunchKernelGGL(pv_analyze_kernel, dim3(grid), dim3(nt), lds, st, a);
}

// --------------------------------------------------------------------------------------------
// phase-locked mode, parallel part: for every peak of every step, the matched previous peak and
// everything about the rotation that does not depend on the recurrence (modifySlicePhaseLocked :640-664).
// The peak lists are shared by the channels of a stream in processing order ch0, ch1, ch0, ...
// (the reference's Impl-member quirk, phasevocoderimpl.h:236-238): the "previous
// peaks" of step (t, c) are those of (t, c-1), or of (t-1, C-1) when c == 0.
// --------------------------------------------------------------------------------------------
constexpr int kMatchThreads = 256; // four waves, one step (slice) each: no workgroup barrier

// nearest set bit to position p in a bitmap of nw 64-bit words (ties -> the lower position); -1 if none
__device__ __forceinline__ int nearest_set_bit(const unsigned long long *bm, int nw, int p) {
    const int w = p >> 6, b = p & 63;
    int up = -1, dn = -1;
    unsigned long long m = bm[w] & (~0ull << b);
    for (int i = w; i < nw; ++i) {
        if (i != w) m = bm[i];
        if (m) {
            up = i * 64 + __ffsll((long long)m) - 1;
            break;
        }
    }
    m = b ? (bm[w] & ((1ull << b) - 1ull)) : 0ull;
    for (int i = w; i >= 0; --i) {
        if (i != w) m = bm[i];
        if (m) {
            dn = i * 64 + 63 - __clzll((long long)m);
            break;
        }
    }
    if (up < 0) return dn;
    if (dn < 0) return up;
    return (up - p) < (p - dn) ? up : dn;
}

// LDS bytes of one wave (= one step) of the match kernel: two bitmaps + prefix counts, then three peak lists
__host__ __device__ inline size_t match_wave_lds(int hs, int PKP) {
    const int nw = (hs + 63) >> 6;
    const size_t bm = (sizeof(unsigned long long) * 2 * nw + sizeof(int) * nw + 15) & ~(size_t)15;
    return bm + 3 * (((size_t)PKP * sizeof(uint16_t) + 15) & ~(size_t)15);
}

__device__ __forceinline__ void match_wave_role(const MatchArgs &a, const int row, const int tl, char *wbase) {
    // bitmaps instead of sorted-list searches: "nearest previous peak" is a find-first-set around p2 in the
    // previous step's peak bitmap; "region of p1" is a prefix popcount over the boundary bitmap of the previous
    // same-channel step (2-4 independent LDS reads instead of two 9-step dependent binary searches).
    // The wave's life is a chain of memory round trips, so there are three of them and no more: (1) the three
    // peak counts and, speculatively and 16 bytes per lane, the three peak lists it may need; (2) the phases at
    // the matched bins of up to six peaks per lane at once; (3) the records out.
    const int nw = (a.hs + 63) >> 6;
    unsigned long long *bprev = reinterpret_cast<unsigned long long *>(wbase);    // [nw] peaks of the previous step
    unsigned long long *bbnd = bprev + nw;                                        // [nw] region boundaries, same channel
    int *pre = reinterpret_cast<int *>(bbnd + nw);                                // [nw] boundaries before each word
    const size_t lpitch = ((size_t)a.PKP * sizeof(uint16_t) + 15) & ~(size_t)15;
    char *lbase = wbase + ((sizeof(unsigned long long) * 2 * nw + sizeof(int) * nw + 15) & ~(size_t)15);
    uint16_t *lcur = reinterpret_cast<uint16_t *>(lbase);                         // [PKP] this step's peaks
    uint16_t *lprev = reinterpret_cast<uint16_t *>(lbase + lpitch);               // [PKP] previous step's peaks
    uint16_t *lsame = reinterpret_cast<uint16_t *>(lbase + 2 * lpitch);           // [PKP] previous same-channel step's
    const int nt = 64, tid = threadIdx.x & 63;
    const int s = row / a.C, c = row - s * a.C;
    const int64_t t = a.t0 + tl;
    const int slot = ring_slot(a.s0, tl, a.TR), pslot = ring_prev(slot, a.TR);
    const int64_t plane = (int64_t)row * a.TR + slot;
    int64_t pplane = -1; // plane of the previous step
    if (c > 0) pplane = (int64_t)(row - 1) * a.TR + slot;
    else if (t > 0) pplane = (int64_t)(s * a.C + a.C - 1) * a.TR + pslot;
    const int64_t splane = t > 0 ? (int64_t)row * a.TR + pslot : -1;
    // round trip 1
    const int ncur = a.npk[plane];
    const int nprev = pplane >= 0 ? a.npk[pplane] : 0;
    const int nsame = splane >= 0 ? a.npk[splane] : 0;
    {
        const uint4 *gcur = reinterpret_cast<const uint4 *>(a.peaks + plane * a.PKP);
        const uint4 *gprev = reinterpret_cast<const uint4 *>(a.peaks + (pplane >= 0 ? pplane : plane) * a.PKP);
        const uint4 *gsame = reinterpret_cast<const uint4 *>(a.peaks + (splane >= 0 ? splane : plane) * a.PKP);
        const int n16 = a.PKP >> 3; // 16-byte pieces per list (PKP % 8 == 0)
        for (int i = tid; i < n16; i += nt) {
            const uint4 v0 = gcur[i], v1 = gprev[i], v2 = gsame[i];
            reinterpret_cast<uint4 *>(lcur)[i] = v0;
            reinterpret_cast<uint4 *>(lprev)[i] = v1;
            reinterpret_cast<uint4 *>(lsame)[i] = v2;
        }
    }
    const bool first = (t == 0 && c == 0);
    const int mode = first ? kModeInit : ((ncur == 0 || nprev == 0) ? kModeProp : kModeLock);
    if (tid == 0) {
        a.modes[plane] = mode;
        // step header in the (always free) last record slot: the sequential kernel fetches it with a vector
        // load together with its peak record, one step ahead
        PeakRec hdr{};
        hdr.p1r1 = (uint32_t)mode | ((uint32_t)ncur << 2);
        a.recs[plane * a.PKP + a.PKP - 1] = hdr;
    }
    if (mode != kModeLock) return;
    for (int i = tid; i < 2 * nw; i += nt) bprev[i] = 0ull; // bprev and bbnd are adjacent
    wave_sync();
    unsigned int *bprev32 = reinterpret_cast<unsigned int *>(bprev), *bbnd32 = reinterpret_cast<unsigned int *>(bbnd);
    for (int i = tid; i < nprev; i += nt) {
        const int b = lprev[i];
        atomicOr(&bprev32[b >> 5], 1u << (b & 31));
    }
    for (int i = tid; i + 1 < nsame; i += nt) {
        const int b = ((int)lsame[i] + (int)lsame[i + 1] + 1) >> 1;
        atomicOr(&bbnd32[b >> 5], 1u << (b & 31));
    }
    wave_sync();
    if (tid < nw) {
        int acc = 0;
        for (int i = 0; i < tid; ++i) acc += __popcll(bbnd[i]);
        pre[tid] = acc;
    }
    wave_sync();
    const float *__restrict__ A2 = a.phase + plane * a.HP;
    const float *__restrict__ A1 = a.phase + (splane >= 0 ? splane : plane) * a.HP;
    const float pinc_f = (float)a.phase_inc[tl], hop_f = (float)a.hop;
    const double Nd = (double)a.N;
    constexpr int kU = 6; // peaks per lane whose phase gathers are in flight together (hs 1024: all of them)
    for (int pb = 0; pb < ncur; pb += nt * kU) {
        int p2v[kU], p1v[kU];
        float a2v[kU], a1v[kU];
        // round trip 2
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int p = pb + tid + nt * u;
            p2v[u] = lcur[p < ncur ? p : ncur - 1];
            // nearest previous peak, ties -> lower bin (== the reference's monotone greedy walk :644-652)
            p1v[u] = nearest_set_bit(bprev, nw, p2v[u]);
            a2v[u] = A2[p2v[u]];
            a1v[u] = A1[p1v[u]]; // prev_phase of this channel == its previous analysis phase
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int p = pb + tid + nt * u;
            if (p >= ncur) continue;
            const int p2 = p2v[u], p1 = p1v[u];
            const float avg_p = (float)((double)(p1 + p2) * 0.5);
            const float pomega = (float)((a.two_pi_hop * (double)(avg_p - 1)) / Nd);
            const float a2 = a2v[u];
            const float a1 = splane >= 0 ? a1v[u] : 0.f;
            const float d1 = a2 - a1 - pomega;
            const float pdelta = (float)((double)pomega + princarg_f(d1));
            // region of p1 in the previous same-channel step = number of its boundaries <= p1
            const int w1 = p1 >> 6, b1 = p1 & 63;
            const unsigned long long le = b1 == 63 ? ~0ull : ((2ull << b1) - 1ull);
            const int r1 = pre[w1] + __popcll(bbnd[w1] & le);
            PeakRec r;
            r.adv = (pdelta * pinc_f) / hop_f;
            r.a2 = a2;
            r.a1 = a1;
            r.p1r1 = (uint32_t)p1 | ((uint32_t)r1 << 16);
            a.recs[plane * a.PKP + p] = r; // round trip 3
        }
    }
}

__global__ __launch_bounds__(kMatchThreads) void pv_match_kernel(const MatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    const int wave = threadIdx.x >> 6;
    const int tl = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (tl >= a.Tn) return; // wave-uniform
    match_wave_role(a, blockIdx.y, tl, smem_raw + (size_t)wave * match_wave_lds(a.hs, a.PKP));
}

void launch_match(const MatchArgs &a, hipStream_t st) {
    const size_t lds = 4 * match_wave_lds(a.hs, a.PKP);
    hipLaunchKernelGGL(pv_match_kernel, dim3((a.Tn + 3) / 4, a.rows), dim3(kMatchThreads), lds, st, a);
}

// --------------------------------------------------------------------------------------------
// phase-locked mode, sequential part: one workgroup per (stream, channel) row walks its slices in
// order.  In a LOCK step only the peaks carry the recurrence:
//   target = princarg(prev_out[p1] + adv),  rot = princarg(target - phase[p2])         (:664-665)
// and prev_out[p1] of a locked previous step is princarg(prev_phase[p1] + rot_prev[region(p1)]) (:688,697),
// evaluated lazily here.  Every other bin's output phase is applied later, in parallel, by the
// synthesis kernel.  INIT / PROP steps (first slice, silence: no peaks) take the per-bin path (:606-636).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ int region_of(const uint16_t *pk, int n, int i) {
    // number of boundaries round((pk[j] + pk[j+1]) / 2) (half away from zero, :676-682) that are <= i
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int bnd = ((int)pk[mid] + (int)pk[mid + 1] + 1) >> 1;
        if (bnd <= i) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// kK: peaks per lane in the fast loop (1: a lane per peak, six waves per row at 2048 points; 3: two waves per row --
// a third of the wave slots and registers beside whatever shares the CU, at the price of a longer step)
template <int kK = 1>
__device__ __forceinline__ void seq_role(const SeqArgs &a, const int row, char *smem_raw) {
    float *srot0 = reinterpret_cast<float *>(smem_raw);          // [PKP]
    float *srot1 = srot0 + a.PKP;                                // [PKP]
    float *spo = srot1 + a.PKP;                                  // [hs] full prev_out (valid when kind == 1)
    uint16_t *spk = reinterpret_cast<uint16_t *>(spo + a.hs);    // [PKP] peaks of the previous same-row step
    const int nt = blockDim.x, tid = threadIdx.x, hs = a.hs;
    const int c = row % a.C;
    int kind = a.st_kind[row];
    float *rprev = srot0, *rcur = srot1;
    if (kind == 2)
        for (int i = tid; i < a.PKP; i += nt) rprev[i] = a.st_rot[(int64_t)row * a.PKP + i];
    if (kind == 1)
        for (int i = tid; i < hs; i += nt) spo[i] = a.st_po[(int64_t)row * hs + i];
    __syncthreads();
    const float hop_f = (float)a.hop;
    const double Nd = (double)a.N;

    // the step's inputs (mode, peak count, this lane's peak record) do not depend on the recurrence: they are
    // fetched one step ahead so the global-load latency overlaps the previous step's princarg chain
    const bool one_pass = a.PKP <= kK * nt; // every peak has its own lane (or one of a lane's kK slots)
    auto plane_of = [&](int tl) { return (int64_t)row * a.TR + ring_slot(a.s0, tl, a.TR); };
    // (one_pass: every peak has its own lane)
    // The header is wave-uniform, but it must come through the vector memory path: a scalar load shares its
    // counter (lgkmcnt) with the LDS reads of the chain and would put its latency back on the critical path.
    // An opaque zero in a VGPR keeps the compiler from scalarising the address.
    int vz = 0;
    asm volatile("" : "+v"(vz));
    // unconditional (index clamped): a load inside an exec-masked branch gets its s_waitcnt at the end of
    // the branch, which would defeat the prefetch
    int rixk[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) rixk[k] = tid + k * nt < a.PKP ? tid + k * nt : a.PKP - 1;
    const int rix = rixk[0];
    // Prefetch ONE step ahead.  Measured alternatives (all slower alone on the GPU): queues 4 and 6 steps deep
    // (+15 %: the chain, not the load latency, bounds a step), always-clamped loads instead of the uniform
    // last-step branch (+13 %).
    auto ld_hdr = [&](int tl) -> uint32_t { return a.recs[plane_of(tl) * a.PKP + a.PKP - 1 + vz].p1r1; };
    auto ld_rec = [&](int tl) -> PeakRec { return a.recs[plane_of(tl) * a.PKP + rix]; };
    auto ld_recs = [&](int tl, PeakRec (&q)[kK]) {
#pragma unroll
        for (int k = 0; k < kK; ++k) q[k] = a.recs[plane_of(tl) * a.PKP + rixk[k]];
    };
    uint32_t h0 = ld_hdr(0);
    PeakRec qk[kK];
    ld_recs(0, qk);
    (void)ld_rec;

    int tl = 0;
    while (tl < a.Tn) {
        // ---- fast inner loop: consecutive phase-locked steps.  It contains exactly two prefetch loads followed
        // by one store per iteration, so the loop-top wait can be the counted vmcnt(1) (the store stays in
        // flight); the slow per-bin path lives outside this loop precisely to keep that count static.
        while (tl < a.Tn && (h0 & 3u) == (uint32_t)kModeLock && one_pass) {
            const int64_t plane = plane_of(tl);
            PeakRec rk[kK];
#pragma unroll
            for (int k = 0; k < kK; ++k) rk[k] = qk[k];
            if (tl + 1 < a.Tn) { // uniform branch
#if !(defined(PV_EXP_SEQ) && (PV_EXP_SEQ & 16))
                h0 = ld_hdr(tl + 1);
                ld_recs(tl + 1, qk);
#endif
            } else {
                h0 = 3u; // sentinel: leaves both loops
            }
            // Branch-free over the lanes: lanes beyond the peak count run on whatever their (clamped) record
            // slot holds and write slots nobody reads.
            // straight-line code (princarg_div has no rare-case branch): any branch inside this loop makes the
            // compiler fall back to vmcnt(0) at the joins, which exposes the store latency under load
            float rtk[kK];
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                const PeakRec r = rk[k];
                const uint32_t r1 = min(r.p1r1 >> 16, (uint32_t)(a.PKP - 1));
                const uint32_t p1 = (r.p1r1 & 0xffffu) & (uint32_t)(hs - 1);
                // (PV_EXP_SEQ: elimination builds for timing only -- tools/build_variant.sh, tools/seq_elim.sh -- bit 0 no
                // princarg_small, 1 no princarg_f, 2 no global store, 3 no barrier, 4 no record prefetch)
#if defined(PV_EXP_SEQ) && (PV_EXP_SEQ & 1)
                const float po_lock = r.a1 + rprev[r1];
#else
                const float po_lock = (float)princarg_small((double)(r.a1 + rprev[r1]));
#endif
                const float po_full = spo[p1];
                const float po = kind == 2 ? po_lock : (kind == 1 ? po_full : 0.f);
#if defined(PV_EXP_SEQ) && (PV_EXP_SEQ & 2)
                const float tgt = po + r.adv;
#else
                const float tgt = (float)princarg_f(po + r.adv);
#endif
#if defined(PV_EXP_SEQ) && (PV_EXP_SEQ & 1)
                rtk[k] = tgt - r.a2;
#else
                rtk[k] = (float)princarg_small((double)(tgt - r.a2));
#endif
            }
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                rcur[rixk[k]] = rtk[k];
#if !(defined(PV_EXP_SEQ) && (PV_EXP_SEQ & 4))
                a.rot[plane * a.PKP + rixk[k]] = rtk[k];
#endif
            }
            kind = 2;
            float *tmp = rprev;
            rprev = rcur;
            rcur = tmp;
            ++tl;
#if defined(PV_EXP_SEQ) && (PV_EXP_SEQ & 8)
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0) only: no barrier (timing experiment)
#else
            __syncthreads();
#endif
        }
        if (tl >= a.Tn) break;
        // ---- general step (first slice, no-peak steps, or more peaks than lanes)
        const int64_t t = a.t0 + tl;
        const int64_t plane = plane_of(tl);
        const int mode = (int)(h0 & 3u), n = (int)(h0 >> 2);
        if (tl + 1 < a.Tn) {
            h0 = ld_hdr(tl + 1);
            ld_recs(tl + 1, qk);
        }
        if (mode == kModeLock) {
            for (int p = tid; p < n; p += nt) {
                const PeakRec r = a.recs[plane * a.PKP + p];
                float po;
                if (kind == 2) po = (float)princarg_small((double)(r.a1 + rprev[r.p1r1 >> 16]));
                else if (kind == 1) po = spo[r.p1r1 & 0xffffu];
                else po = 0.f;
                const float tgt = (float)princarg_f(po + r.adv);
                const float rt = (float)princarg_small((double)(tgt - r.a2));
                rcur[p] = rt;
                a.rot[plane * a.PKP + p] = rt;
            }
            kind = 2;
            float *tmp = rprev;
            rprev = rcur;
            rcur = tmp;
        } else {
            // per-bin path; first materialise prev_out when the previous step of this row was locked
            const float *__restrict__ A = a.phase + plane * a.HP;
            const int64_t splane = t > 0 ? (int64_t)row * a.TR + ring_prev(ring_slot(a.s0, tl, a.TR), a.TR) : -1;
            const float *__restrict__ Ap = splane >= 0 ? a.phase + splane * a.HP : nullptr;
            int nsame = 0;
            if (kind == 2) {
                nsame = a.npk[splane];
                for (int i = tid; i < nsame; i += nt) spk[i] = a.peaks[splane * a.PKP + i];
                __syncthreads();
            }
            const float pinc_f = (float)a.phase_inc[tl];
            float *__restrict__ outp = a.outphase + plane * a.HP;
            for (int i = tid; i < hs; i += nt) {
                const float phi = A[i];
                float outv;
                if (mode == kModeInit) {
                    outv = phi;
                } else {
                    float po;
                    if (kind == 2) po = (float)princarg_small((double)(Ap[i] + rprev[region_of(spk, nsame, i)]));
                    else if (kind == 1) po = spo[i];
                    else po = 0.f;
                    const float pp = Ap ? Ap[i] : 0.f;
                    const float omega = (float)((a.two_pi_hop * (double)i) / Nd);
                    const float d1 = phi - pp - omega;
                    const float delta = (float)((double)omega + princarg_f(d1));
                    const float advance = delta * pinc_f / hop_f;
                    outv = (float)princarg_f(po + advance);
                }
                spo[i] = outv;
                outp[i] = outv;
            }
            kind = 1;
        }
        ++tl;
        __syncthreads();
    }
    if (kind == 2)
        for (int i = tid; i < a.PKP; i += nt) a.st_rot[(int64_t)row * a.PKP + i] = rprev[i];
    if (kind == 1)
        for (int i = tid; i < hs; i += nt) a.st_po[(int64_t)row * hs + i] = spo[i];
    if (tid == 0) a.st_kind[row] = kind;
    (void)c;
}

__host__ __device__ size_t seq_lds_bytes(const SeqArgs &a);
// --------------------------------------------------------------------------------------------
// The same walk with the step records prefetched kSeqDepth steps ahead through an LDS ring (round 3).  Elimination
// builds (profiles/r02/seq_elim.txt) showed that a step of the loop above IS the latency of its one-step-ahead record
// load (0.94 us with it, 0.54 without); register queues four to eight steps deep lost to the compiler's waits and
// spills.  Here a wave fetches its own 64 records of step t + kSeqDepth - 1 with ONE global_load_lds_dwordx4 (LDS-DMA:
// no destination registers, so nothing to spill and nothing the compiler waits for) into slot (t - 1) % kSeqDepth --
// the slot step t - 1 has just finished with -- and reads step t's records from LDS.  The loads, the rotation store
// and their waits are inline assembly so that the vmcnt counts below are exact; the compiler sees no vector memory
// operation in the fast loop.  Arithmetic: the loop above, operation for operation.
//
// Waits.  VM operations retire in issue order (one counter on gfx9).  At the end of iteration t the records of step
// t + 1 must have landed.  Issued behind that load by then: the loads of steps t + 2 .. t + kSeqDepth - 1 (those that
// exist) and one store per iteration since it was issued, so
//     steady state (the last kSeqDepth - 1 iterations all took the fast path):  vmcnt(2 kSeqDepth - 3)
//     conservative (start of the launch, behind a general step):               vmcnt(kSeqDepth - 2), loads only
//     tail (no more loads issued):                                             vmcnt(kSeqDepth - 1) resp. vmcnt(0)
// a count that is too small only waits longer.  The wait sits in front of the step's barrier, the read behind it: the
// step header (last record slot) is read by every wave, and data an LDS-DMA wrote is another wave's to read only
// after that wave's wait and a barrier.
// --------------------------------------------------------------------------------------------
constexpr int kSeqDepthMax = 8;

__device__ __forceinline__ void seq_glds16(const void *gsrc, uint32_t lds_dst) { // lds_dst: wave-uniform byte address
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N> __device__ __forceinline__ void seq_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N < 0 ? 0 : N) : "memory");
}

// (the step's own LDS reads -- header and record -- are issued one step ahead as well, at the top of the step before,
// so that a step's dependent chain is what it was with the record in registers: rotation read, three wraps, rotation
// write, barrier.  Hence the waits below are for step t + 2.)
template <int D> __device__ __forceinline__ void seq_role_ring(const SeqArgs &a, const int row, char *smem_raw) {
    static_assert(D >= 4 && D <= kSeqDepthMax, "ring depth");
    float *srot0 = reinterpret_cast<float *>(smem_raw);          // [PKP]
    float *srot1 = srot0 + a.PKP;                                // [PKP]
    float *spo = srot1 + a.PKP;                                  // [hs] full prev_out (valid when kind == 1)
    uint16_t *spk = reinterpret_cast<uint16_t *>(spo + a.hs);    // [PKP] peaks of the previous same-row step
    const int nt = blockDim.x, tid = threadIdx.x, hs = a.hs;
    PeakRec *ring = reinterpret_cast<PeakRec *>(smem_raw + ((seq_lds_bytes(a) + 15) & ~(size_t)15)); // [D][nt]
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)reinterpret_cast<char *>(ring);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int kind = a.st_kind[row];
    float *rprev = srot0, *rcur = srot1;
    if (kind == 2)
        for (int i = tid; i < a.PKP; i += nt) rprev[i] = a.st_rot[(int64_t)row * a.PKP + i];
    if (kind == 1)
        for (int i = tid; i < hs; i += nt) spo[i] = a.st_po[(int64_t)row * hs + i];
    __syncthreads(); // (the compiler's loads above are complete: it waited for their values)
    const float hop_f = (float)a.hop;
    const double Nd = (double)a.N;
    auto plane_of = [&](int tl) { return (int64_t)row * a.TR + ring_slot(a.s0, tl, a.TR); };
    const int rix = tid < a.PKP ? tid : a.PKP - 1; // (nt >= PKP: every peak has its own lane)
    auto fetch = [&](int step) { // this wave's 64 records of `step` into the step's ring slot
        const PeakRec *src = a.recs + plane_of(step) * a.PKP + rix;
        const uint32_t dst = ring_lds + (uint32_t)(((step % D) * nt + wave * 64) * (int)sizeof(PeakRec));
        seq_glds16(src, __builtin_amdgcn_readfirstlane(dst));
    };
    for (int sft = 0; sft < D - 1 && sft < a.Tn; ++sft) fetch(sft);
    if (a.Tn >= D - 1) seq_wait_barrier<D - 3>(); // steps 0 and 1 have landed (D - 3 younger loads may be in flight)
    else seq_wait_barrier<0>();
    uint32_t h_next = ring[a.PKP - 1].p1r1;
    PeakRec r_next = ring[tid];

    int fast_run = 0; // consecutive iterations, ending with the current one, that issued exactly one (asm) store
    for (int tl = 0; tl < a.Tn; ++tl) {
        if (tl - 1 + D < a.Tn) fetch(tl - 1 + D); // into the slot step tl - 1 used: every wave is past that step's barrier
        const uint32_t h0 = h_next;
        const PeakRec r = r_next;
        if (tl + 1 < a.Tn) { // the next step's header and record (landed: the wait at the end of step tl - 1)
            const PeakRec *slot = ring + ((tl + 1) % D) * nt;
            h_next = slot[a.PKP - 1].p1r1;
            r_next = slot[tid];
        }
        const int mode = (int)(h0 & 3u);
        const int64_t plane = plane_of(tl);
        if (mode == kModeLock) {
            // ---- phase-locked step: branch-free over the lanes (lanes beyond the peak count run on whatever their
            // record slot holds and write slots nobody reads)
            const uint32_t r1 = min(r.p1r1 >> 16, (uint32_t)(a.PKP - 1));
            const uint32_t p1 = (r.p1r1 & 0xffffu) & (uint32_t)(hs - 1);
            const float po_lock = (float)princarg_small((double)(r.a1 + rprev[r1]));
            const float po_full = spo[p1];
            const float po = kind == 2 ? po_lock : (kind == 1 ? po_full : 0.f);
            const float tgt = (float)princarg_f(po + r.adv);
            const float rt = (float)princarg_small((double)(tgt - r.a2));
            rcur[rix] = rt;
            {
                float *dst = a.rot + plane * a.PKP + rix;
                asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(rt) : "memory");
            }
            kind = 2;
            float *tmp = rprev;
            rprev = rcur;
            rcur = tmp;
            ++fast_run;
        } else {
            // ---- per-bin step (first slice, silence): prev_out materialised when the previous step was locked
            fast_run = 0;
            const int64_t t = a.t0 + tl;
            const float *__restrict__ A = a.phase + plane * a.HP;
            const int64_t splane = t > 0 ? (int64_t)row * a.TR + ring_prev(ring_slot(a.s0, tl, a.TR), a.TR) : -1;
            const float *__restrict__ Ap = splane >= 0 ? a.phase + splane * a.HP : nullptr;
            int nsame = 0;
            if (kind == 2) {
                nsame = a.npk[splane];
                for (int i = tid; i < nsame; i += nt) spk[i] = a.peaks[splane * a.PKP + i];
                __syncthreads();
            }
            const float pinc_f = (float)a.phase_inc[tl];
            float *__restrict__ outp = a.outphase + plane * a.HP;
            for (int i = tid; i < hs; i += nt) {
                const float phi = A[i];
                float outv;
                if (mode == kModeInit) {
                    outv = phi;
                } else {
                    float po;
                    if (kind == 2) po = (float)princarg_small((double)(Ap[i] + rprev[region_of(spk, nsame, i)]));
                    else if (kind == 1) po = spo[i];
                    else po = 0.f;
                    const float pp = Ap ? Ap[i] : 0.f;
                    const float omega = (float)((a.two_pi_hop * (double)i) / Nd);
                    const float d1 = phi - pp - omega;
                    const float delta = (float)((double)omega + princarg_f(d1));
                    const float advance = delta * pinc_f / hop_f;
                    outv = (float)princarg_f(po + advance);
                }
                spo[i] = outv;
                outp[i] = outv;
            }
            kind = 1;
        }
        // ---- the records of step tl + 2 have landed; then the step's barrier
        if (tl + 2 >= a.Tn) {
            seq_wait_barrier<0>();
        } else if (fast_run >= D - 2) {
            if (tl - 1 + D < a.Tn) seq_wait_barrier<2 * D - 5>();
            else seq_wait_barrier<D - 2>();
        } else {
            if (tl - 1 + D < a.Tn) seq_wait_barrier<D - 3>();
            else seq_wait_barrier<0>();
        }
    }
    if (kind == 2)
        for (int i = tid; i < a.PKP; i += nt) a.st_rot[(int64_t)row * a.PKP + i] = rprev[i];
    if (kind == 1)
        for (int i = tid; i < hs; i += nt) a.st_po[(int64_t)row * hs + i] = spo[i];
    if (tid == 0) a.st_kind[row] = kind;
}

template <int D> __global__ __launch_bounds__(1024) void pv_seq_ring_kernel(const SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    if (a.high_prio) __builtin_amdgcn_s_setprio(3);
    seq_role_ring<D>(a, blockIdx.x, smem_raw);
}

template <int kK> __global__ __launch_bounds__(1024) void pv_seq_kernel(const SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    // the chain is pure latency and may share the GPU with the overlap-add tiles of the previous chunk (second
    // HIP stream): let its few waves win every issue arbitration
    if (a.high_prio) __builtin_amdgcn_s_setprio(3);
    seq_role<kK>(a, blockIdx.x, smem_raw);
}

// Single-stream engine: match and rotation chain in ONE launch (round 3).  A 480-frame call is two or three steps per
// row: far too little for the two kernels to need separate grids, and every launch costs the host ~7 us of the call's
// ~70.  One workgroup per row: its waves match the row's steps (a step's match depends on analysis results only, of
// this row and its neighbour channel -- nothing another workgroup of this launch writes), then, behind a device-scope
// fence and a barrier, the same workgroup walks the chain.  Same device functions, same results.
template <int D> __global__ __launch_bounds__(1024) void pv_phase_kernel(const MatchArgs m, const SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    for (int tl = wave; tl < m.Tn; tl += nw)
        match_wave_role(m, blockIdx.x, tl, smem_raw + (size_t)wave * match_wave_lds(m.hs, m.PKP));
    __threadfence(); // the records go through global memory: the chain's loads must see them
    __syncthreads();
    if (a.high_prio) __builtin_amdgcn_s_setprio(3);
    seq_role_ring<D>(a, blockIdx.x, smem_raw);
}

int seq_threads(int PKP) {
    int nt = (PKP + 63) & ~63;
    if (nt > 1024) nt = 1024;
    if (nt < 64) nt = 64;
    return nt;
}
__host__ __device__ size_t seq_lds_bytes(const SeqArgs &a) { return sizeof(float) * ((size_t)2 * a.PKP + a.hs) + sizeof(uint16_t) * a.PKP; }

void launch_seq(const SeqArgs &a, hipStream_t st) {
    const size_t lds = seq_lds_bytes(a);
    static unsigned long long big1 = 0, big3 = 0;
    if (a.narrow && a.PKP <= 3 * 1024) {
        // three peaks per lane: a third of the waves (two per row at 2048 points), so that the kernel finds room on
        // a CU beside the fused kernel's workgroup
        int nt = ((a.PKP + 2) / 3 + 63) & ~63;
        if (nt > 1024) nt = 1024;
        allow_big_lds_dev(pv_seq_kernel<3>, big3);
        hipLaunchKernelGGL(pv_seq_kernel<3>, dim3(a.rows), dim3(nt), lds, st, a);
        return;
    }
    const int nt = seq_threads(a.PKP);
    static const bool no_ring = [] { // AUDIOMOD_PV_SEQ_RING=0: round 2's one-step-ahead register prefetch (for A/B runs)
        const char *e = getenv("AUDIOMOD_PV_SEQ_RING");
        return e && atoi(e) == 0;
    }();
    static const int depth = [] { // AUDIOMOD_PV_SEQ_DEPTH=4|8 (tuning knob)
        const char *e = getenv("AUDIOMOD_PV_SEQ_DEPTH");
        return e ? atoi(e) : 0;
    }();
    // Four steps ahead where the kernel shares its CUs with a full batch's other kernels (a deeper ring's LDS then
    // costs more than its latency cover returns: 0.58 vs 0.38 ms per launch beside the resampling kernel); eight for
    // small batches, which are bound by this chain and leave the LDS free (20 streams: 10.6 -> 11.0 G samples/s).
    const int D = depth ? (depth <= 4 ? 4 : 8) : (a.rows <= 96 ? 8 : 4);
    const size_t ring_lds = ((lds + 15) & ~(size_t)15) + (size_t)D * nt * sizeof(PeakRec);
    if (!no_ring && a.PKP <= nt && ring_lds <= 160 * 1024 - 512) {
        static unsigned long long bigr4 = 0, bigr8 = 0;
        if (D == 4) {
            allow_big_lds_dev(pv_seq_ring_kernel<4>, bigr4);
            hipLaunchKernelGGL(pv_seq_ring_kernel<4>, dim3(a.rows), dim3(nt), ring_lds, st, a);
        } else {
            allow_big_lds_dev(pv_seq_ring_kernel<8>, bigr8);
            hipLaunchKernelGGL(pv_seq_ring_kernel<8>, dim3(a.rows), dim3(nt), ring_lds, st, a);
        }
        return;
    }
    allow_big_lds_dev(pv_seq_kernel<1>, big1);
    hipLaunchKernelGGL(pv_seq_kernel<1>, dim3(a.rows), dim3(nt), lds, st, a);
}

// (true when the fused kernel was launched; false: the caller launches the two kernels)
bool launch_phase(const MatchArgs &m, const SeqArgs &a, hipStream_t st) {
    constexpr int D = 8;
    const int nt = seq_threads(a.PKP);
    const size_t seq_l = ((seq_lds_bytes(a) + 15) & ~(size_t)15) + (size_t)D * nt * sizeof(PeakRec);
    const size_t match_l = (size_t)(nt / 64) * match_wave_lds(m.hs, m.PKP);
    const size_t lds = seq_l > match_l ? seq_l : match_l;
    if (a.PKP > nt || lds > 160 * 1024 - 512 || a.narrow) return false;
    static unsigned long long big = 0;
    allow_big_lds_dev(pv_phase_kernel<D>, big);
    hipLaunchKernelGGL(pv_phase_kernel<D>, dim3(a.rows), dim3(nt), lds, st, m, a);
    return true;
}

// --------------------------------------------------------------------------------------------
// coremode 0 (modifySliceSimple :708-753): independent per-bin recurrences; one thread per bin streams
// over the slices of the launch with its state in registers.
// --------------------------------------------------------------------------------------------
constexpr int kPropThreads = 256;

__device__ __forceinline__ void prop_role(const PropArgs &a, const int row, const int i) {
    // Only out = princarg(prev_out + advance) is a recurrence: the advance of a step depends on the analysis phases
    // of that step and the one before, which are data.  So the slices are taken four at a time -- their phases
    // loaded together, their advances computed side by side -- and only the last princarg of each forms the chain.
    const int c = row % a.C;
    const float *__restrict__ ph = a.phase;
    float *__restrict__ op = a.outphase;
    float pp = a.st_pp[(int64_t)row * a.hs + i], po = a.st_po[(int64_t)row * a.hs + i];
    const float omega = (float)((a.two_pi_hop * (double)i) / (double)a.N);
    const float hop_f = (float)a.hop;
    constexpr int kU = 4;
    for (int tl0 = 0; tl0 < a.Tn; tl0 += kU) {
        float phi[kU], adv[kU];
        int64_t plane[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int tl = tl0 + u < a.Tn ? tl0 + u : a.Tn - 1;
            plane[u] = (int64_t)row * a.TR + ring_slot(a.s0, tl, a.TR);
            phi[u] = ph[plane[u] * a.HP + i];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int tl = tl0 + u < a.Tn ? tl0 + u : a.Tn - 1;
            const float d1 = phi[u] - (u ? phi[u - 1] : pp) - omega;
            const float delta = (float)((double)omega + princarg_f(d1));
            adv[u] = delta * (float)a.phase_inc[tl] / hop_f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (tl0 + u >= a.Tn) break;
            // firstentry: only the very first step of the stream (function-static in the reference)
            const float outv = (a.t0 + tl0 + u == 0 && c == 0) ? phi[u] : (float)princarg_f(po + adv[u]);
            pp = phi[u];
            po = outv;
            op[plane[u] * a.HP + i] = outv;
        }
    }
    a.st_pp[(int64_t)row * a.hs + i] = pp;
    a.st_po[(int64_t)row * a.hs + i] = po;
}

__global__ __launch_bounds__(kPropThreads) void pv_prop_kernel(const PropArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.hs) prop_role(a, blockIdx.y, i);
}

void launch_prop(const PropArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(pv_prop_kernel, dim3((a.hs + kPropThreads - 1) / kPropThreads, a.rows), dim3(kPropThreads), 0,
                       st, a);
}

// --------------------------------------------------------------------------------------------
// synthesis (+ per-bin application of the phase modification)
// --------------------------------------------------------------------------------------------
// modifySliceVocoder (phasevocoderprocess.cc:755-776): carrier magnitude of bin k times the mean modulator
// magnitude of its band (band_len bins, float running sum from 0, divided by band_len*2); DC and Nyquist zeroed.
__device__ __forceinline__ float vocoder_mag(const float *__restrict__ mod, const float *__restrict__ cmag, int k,
                                             int hs, int band_len) {
    if (k == 0 || k == hs) return 0.f;
    float mg = cmag[k];
    if (band_len > 0 && k < 512 * band_len) {
        const int j0 = (k / band_len) * band_len;
        float mean = 0.f;
        for (int i = 0; i < band_len; ++i) mean += mod[j0 + i];
        mean /= (float)(band_len * 2);
        mg *= mean;
    }
    return mg;
}

__global__ __launch_bounds__(kFftThreads) void pv_synth_kernel(const SynthArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    const DevTables &tb = a.tb;
    const int N = tb.N, hs = tb.hs, nc = tb.nc, nt = blockDim.x;
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);        // [nc]
    float2 *X = buf + nc;                                      // [nc + 1]
    float *sph = reinterpret_cast<float *>(X + nc + 1);        // [hs + 1] output phase of every bin
    float *srot = sph + hs + 4;                                // [PKP]
    uint16_t *spk = reinterpret_cast<uint16_t *>(srot + a.PKP); // [PKP]
    int row, tl;
    if (!block_to_row_slice(a.Tn, a.rows, row, tl)) return;
    const int64_t t = a.t0 + tl;
    const int64_t plane = (int64_t)row * a.TR + ring_slot(a.s0, tl, a.TR);
    const float *__restrict__ mag = a.mag + plane * tb.HP;
    const float *__restrict__ A = a.phase + plane * tb.HP;
    const double Nd = (double)N;

    // 1. output phase of every bin
    const int cslot = ring_slot(a.s0, tl, a.TR);
    if (a.voc_band_len >= 0) {
        const float *__restrict__ cp = a.cphase + (int64_t)cslot * tb.HP;
        for (int k = threadIdx.x; k <= hs; k += nt) sph[k] = cp[k];
    } else if (a.robotic) {
        for (int k = threadIdx.x; k <= hs; k += nt) sph[k] = 0.f;
    } else if (a.passthru) {
        for (int k = threadIdx.x; k <= hs; k += nt) sph[k] = A[k];
    } else if (a.whisper) {
        const float *__restrict__ wp = a.whisper + ((int64_t)tl * a.C + row % a.C) * tb.HP;
        for (int k = threadIdx.x; k <= hs; k += nt) sph[k] = wp[k];
    } else if (a.coremode == 2) {
        const float pinc_f = (float)a.phase_inc[tl], hop_f = (float)a.hop;
        for (int k = threadIdx.x; k <= hs; k += nt) sph[k] = k < hs ? A[k] * pinc_f / hop_f : A[k];
    } else {
        const int mode = a.coremode == 1 ? a.modes[plane] : kModeProp;
        if (mode == kModeLock) {
            const int n = a.npk[plane];
            for (int i = threadIdx.x; i < n; i += nt) {
                spk[i] = a.peaks[plane * a.PKP + i];
                srot[i] = a.rot[plane * a.PKP + i];
            }
            __syncthreads();
            for (int k = threadIdx.x; k <= hs; k += nt) {
                const float phi = A[k];
                sph[k] = k < hs ? (float)princarg_small((double)(phi + srot[region_of(spk, n, k)])) : phi;
            }
        } else {
            const float *__restrict__ op = a.outphase + plane * tb.HP;
            for (int k = threadIdx.x; k <= hs; k += nt) sph[k] = k < hs ? op[k] : A[k];
        }
    }
    __syncthreads();

    // 2. spectrum: freqCompSlice gather (:869-916, both branches are pure gathers from the pre-call arrays),
    //    gains, polar -> cartesian (FFT.cc:2711-2718)
    for (int k = threadIdx.x; k <= hs; k += nt) {
        float mg, p;
        if (a.do_freq_comp) {
            if (a.freq_comp > 1.0f) {
                const int src = __float2int_rn((float)k * a.freq_comp);
                if (src > hs) {
                    mg = 0.f;
                    p = 0.f;
                } else {
                    mg = mag[src];
                    p = sph[src] + (float)((a.two_pi_hop * (double)(k - src)) / Nd);
                }
            } else if (k < hs) {
                const int src = __float2int_rn((float)k * a.freq_comp);
                mg = mag[src];
                p = sph[src] + (float)((a.two_pi_hop * (double)(k - src)) / Nd);
            } else {
                mg = mag[k];
                p = sph[k];
            }
            mg *= a.fixed_gain;
        } else if (a.voc_band_len >= 0) {
            mg = vocoder_mag(mag, a.cmag + (int64_t)cslot * tb.HP, k, hs, a.voc_band_len);
            p = sph[k];
        } else {
            mg = mag[k];
            p = sph[k];
        }
        mg *= a.inv_n;
        X[k] = make_float2(mg * cosf(p), mg * sinf(p));
    }
    __syncthreads();

    // 3. kiss_fftri pre-pass (kiss_fftr.c:134-157), scattered straight into butterfly order
    for (int k = threadIdx.x; k <= nc / 2; k += nt) {
        if (k == 0) {
            buf[tb.iperm[0]] = make_float2(X[0].x + X[nc].x, X[0].x - X[nc].x);
        } else {
            const float2 fk = X[k];
            const float2 q = X[nc - k];
            const float2 fnkc = make_float2(q.x, -q.y);
            const float2 fek = cadd(fk, fnkc);
            const float2 tq = csub(fk, fnkc);
            const float2 fok = cmul(tq, tb.st_inv[k]);
            const float2 u = cadd(fek, fok);
            float2 v = csub(fek, fok);
            v.y = v.y * -1.f;
            if (k != nc - k) buf[tb.iperm[k]] = u;
            buf[tb.iperm[nc - k]] = v;
        }
    }
    __syncthreads();
    fft_stages<true>(buf, tb, tb.tw_inv);

    // 4. ifftshift + synthesis window (phasevocoderimpl.h:183-198)
    const float *fb = reinterpret_cast<const float *>(buf);
    const int fslot = (int)(t & (int64_t)(a.FR - 1));
    float *__restrict__ out = a.frames + ((int64_t)row * a.FR + fslot) * N;
    const float *__restrict__ w = tb.window;
    for (int i = threadIdx.x; i < N; i += nt) out[i] = fb[(i + hs) & (N - 1)] * w[i];
}

// --------------------------------------------------------------------------------------------
// synthesis, wave-per-frame variant.  The wave-private LDS region is reused four times:
// [output phases + rot/peak lists] -> [spectrum X] -> [butterfly-ordered input] -> [time-domain frame].
// --------------------------------------------------------------------------------------------
// kSink: 0 = the windowed frame goes to the HBM frame ring (a.frames); 1 = the un-windowed, un-shifted time-domain
// frame stays in the wave's LDS region (element e at lds[W::pad(e)] = samples 2e, 2e + 1) for the fused
// overlap-add that follows in the same kernel (pv_synth_chain_kernel)
// kFast: PV_ARITH_FAST (include/audiomod_pv.h) -- behind the phase propagation the output is continuous in everything
// computed here, so the arithmetic is free within the 1e-4 RMS contract: the region rotation is added without the
// wrap (sine and cosine do not care), sine / cosine come from the hardware's v_sin_f32 / v_cos_f32 (argument in turns),
// complex products use fma, and the first FFT pass multiplies by literal twiddles (pv_wavefft.h).  ~30 % fewer vector
// instructions per slice; measured against the oracle: see tests/test_gpu_parity.py (fast arithmetic) and bench.py.
template <int NC, int kPlainCore = -1, int kSink = 0, bool kFast = false>
__device__ __forceinline__ void synth_wave_role(const SynthArgs &a_in, const int row, const int tl, cf *lds,
                                                const int lane_in = -1) {
    // kPlainCore >= 0: the plain pitch shift / stretch in that core mode (no frequency compression, vocoder,
    // robotic, whisper or pass-through source): the mode switches fold away, the kernel is half the code
    SynthArgs a = a_in;
    if (kPlainCore >= 0) {
        a.do_freq_comp = 0, a.voc_band_len = -1, a.robotic = 0, a.passthru = 0, a.whisper = nullptr;
        a.coremode = kPlainCore;
    }
    // kPlainCore == 3: the formant / gender modes in the phase-locked core mode (frequency compression on, every
    // other switch off), for the fused kernel: with twelve waves per workgroup it has the registers (round 1's fold
    // spilled at 128)
    if (kPlainCore == 3) a.do_freq_comp = 1, a.coremode = 1;
    using W = WF<NC>;
    constexpr int N = 2 * NC, hs = NC, R = W::R;
    // (a caller that loops over slices passes a lane id that is opaque per iteration, so that the lane-dependent
    // addresses and constants below are not hoisted out of its loop and kept in registers across iterations)
    const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
    const DevTables &tb = a.tb;
    const int64_t t = a.t0 + tl;
    const int64_t plane = (int64_t)row * a.TR + ring_slot(a.s0, tl, a.TR);
    const float *__restrict__ mag = a.mag + plane * tb.HP;
    const float *__restrict__ A = a.phase + plane * tb.HP;
    const cf *__restrict__ tw = reinterpret_cast<const cf *>(tb.tw_inv);
    const cf *__restrict__ stw = reinterpret_cast<const cf *>(tb.st_inv);
    const cf2 *__restrict__ twl = reinterpret_cast<const cf2 *>(tb.twl_inv);
    const double Nd = (double)N;
    float *sph = reinterpret_cast<float *>(lds);                    // [hs + 1]
    float *srot = sph + hs + 4;                                     // [PKP]
    uint16_t *spk = reinterpret_cast<uint16_t *>(srot + a.PKP);     // [PKP]
    constexpr int JB = NC / 64; // bins per lane (plus the Nyquist bin on lane 0)

    // 1. every global value this frame needs, requested before anything is computed: the per-bin phases of
    //    whichever source the mode selects, the magnitudes, and (phase-locked steps) the peak list and its
    //    rotations, read speculatively up to the list's capacity so that they do not wait for the peak count.
    const int cslot = ring_slot(a.s0, tl, a.TR);
    constexpr int QP = (NC / 3 + 9 + 63) / 64; // >= PKP / 64 (pv_engine.cc: pkmax = hs / 3 + 2)
    const bool plain = !a.do_freq_comp && a.voc_band_len < 0; // magnitudes are used bin by bin, unmoved
    const float *__restrict__ psrc = A;
    int mode = kModeProp, n = 0;
    bool zero_phase = false;
    if (a.voc_band_len >= 0) {
        psrc = a.cphase + (int64_t)cslot * tb.HP;
    } else if (a.robotic) {
        zero_phase = true;
    } else if (a.passthru) {
    } else if (a.whisper) {
        psrc = a.whisper + ((int64_t)tl * a.C + row % a.C) * tb.HP;
    } else if (a.coremode == 2) {
    } else {
        mode = a.coremode == 1 ? a.modes[plane] : kModeProp;
        if (mode == kModeLock) n = a.npk[plane];
        else psrc = a.outphase + plane * tb.HP;
    }
    // A lane owns runs of four consecutive bins, k = 4 (lane + 64 q) + c: one 16-byte load per run and plane
    // (the texture path is issue-bound), and the spectrum goes to LDS as two 16-byte writes per run.
    constexpr int QB = NC / 256; // runs per lane
    constexpr int QH = (QB + 1) / 2; // magnitude runs fetched up front (the rest while those are being used)
    float4 base[QB], mreg[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) base[q] = *reinterpret_cast<const float4 *>(psrc + 4 * (lane + 64 * q));
    // the Nyquist bin keeps its analysis phase in every mode but the carrier / whisper / robotic ones (:695-699)
    float pnyq = (a.voc_band_len >= 0 || a.whisper) ? psrc[hs] : A[hs];
    float mnyq = 0.f;
    auto load_mags = [&]() { // issued once the peak list has left its registers (128-VGPR budget)
        if (plain) { // the first half now, the second half while the first is being used (see below)
#pragma unroll
            for (int q = 0; q < QH; ++q) mreg[q] = *reinterpret_cast<const float4 *>(mag + 4 * (lane + 64 * q));
            mnyq = mag[hs];
        } else if (a.do_freq_comp) { // freqCompSlice gathers across bins: the whole row, staged in LDS below
#pragma unroll
            for (int q = 0; q < QB; ++q) mreg[q] = *reinterpret_cast<const float4 *>(mag + 4 * (lane + 64 * q));
            mnyq = mag[hs];
        }
    };
    uint16_t pkr[QP];
    float rotr[QP];
    if (mode == kModeLock) {
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const int i = lane + 64 * q;
            const int ic = i < a.PKP ? i : 0;
            pkr[q] = a.peaks[plane * a.PKP + ic];
            rotr[q] = a.rot[plane * a.PKP + ic];
        }
    }

    // 2. output phase of every bin, in registers (and in sph for the modes that gather across bins)
    if (mode != kModeLock) load_mags();
    if (zero_phase) {
#pragma unroll
        for (int q = 0; q < QB; ++q) base[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        pnyq = 0.f;
    } else if (a.coremode == 2 && !a.passthru && !a.whisper && a.voc_band_len < 0) {
        const float pinc_f = (float)a.phase_inc[tl], hop_f = (float)a.hop;
#pragma unroll
        for (int q = 0; q < QB; ++q)
            base[q] = make_float4(base[q].x * pinc_f / hop_f, base[q].y * pinc_f / hop_f, base[q].z * pinc_f / hop_f,
                                  base[q].w * pinc_f / hop_f);
    } else if (mode == kModeLock) {
        // region(k) = number of region boundaries <= k.  Boundaries go into a bitmap (one 64-bit word per
        // 64 bins), so the lookup is a prefix count + one masked popcount per bin.
        unsigned int *bits32 = reinterpret_cast<unsigned int *>(spk + a.PKP); // [2 * JB]
        int *pre = reinterpret_cast<int *>(bits32 + 2 * JB);                  // [JB]
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const int i = lane + 64 * q;
            if (i < n) {
                spk[i] = pkr[q];
                srot[i] = rotr[q];
            }
        }
        load_mags();
        if (lane < 2 * JB) bits32[lane] = 0u;
        wave_sync();
        for (int i = lane; i + 1 < n; i += 64) {
            const int b = ((int)spk[i] + (int)spk[i + 1] + 1) >> 1; // round(x.5) away from zero (:676-682)
            atomicOr(&bits32[b >> 5], 1u << (b & 31));
        }
        wave_sync();
        const unsigned long long *bits = reinterpret_cast<const unsigned long long *>(bits32);
        if (lane < JB) {
            int acc = 0;
            for (int i = 0; i < lane; ++i) acc += __popcll(bits[i]);
            pre[lane] = acc;
        }
        wave_sync();
        const int b0 = 4 * (lane & 15); // bit of the run's first bin in its 64-bin word
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int wd = (lane >> 4) + 4 * q; // == (4 (lane + 64 q)) >> 6
            const unsigned long long word = bits[wd];
            const int p0 = pre[wd];
            auto rotated = [&](float ph, int c) -> float {
                const int reg = p0 + __popcll(word & ((2ull << (b0 + c)) - 1ull));
                if (kFast) return ph + srot[reg]; // |.| <= 2 pi: the wrap only matters to a comparison, not to sin / cos
                return (float)princarg_small((double)(ph + srot[reg]));
            };
            base[q] = make_float4(rotated(base[q].x, 0), rotated(base[q].y, 1), rotated(base[q].z, 2),
                                  rotated(base[q].w, 3));
        }
    }
    float *smg = sph + hs + 4; // [hs + 1] magnitudes for the gather (over the rotation / peak lists, done with)
    if (!plain) {
        wave_sync();
#pragma unroll
        for (int q = 0; q < QB; ++q) *reinterpret_cast<float4 *>(sph + 4 * (lane + 64 * q)) = base[q];
        if (lane == 0) sph[hs] = pnyq;
        if (a.do_freq_comp) {
#pragma unroll
            for (int q = 0; q < QB; ++q) *reinterpret_cast<float4 *>(smg + 4 * (lane + 64 * q)) = mreg[q];
            if (lane == 0) smg[hs] = mnyq;
        }
        wave_sync();
    }

    // spectrum in registers: freqCompSlice gather (:869-916), gains, polar -> cartesian (FFT.cc:2711-2718)
    cf xs[QB][4];
    cf xnyq = cf{0.f, 0.f};
    auto to_cartesian = [&](float mg, float p) -> cf {
        mg *= a.inv_n;
        float sn, cs;
        if (kFast) { // phases unwrapped by freqCompSlice reach hundreds of radians: two-term reduction, then turns
            const float n = __builtin_rintf(p * 0.15915494309189535f);
            float r = __builtin_fmaf(n, -6.2831854820251465f, p);
            r = __builtin_fmaf(n, 1.7484555e-07f, r);
            const float rev = r * 0.15915494309189535f;
            sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
        } else {
            sincosf(p, &sn, &cs);
        }
        return cf{mg * cs, mg * sn};
    };
    auto spectrum_bin = [&](int k) -> cf {
        float mg, p;
        if (a.do_freq_comp) {
            if (a.freq_comp > 1.0f) {
                const int src = __float2int_rn((float)k * a.freq_comp);
                if (src > hs) {
                    mg = 0.f;
                    p = 0.f;
                } else {
                    mg = smg[src];
                    p = sph[src] + (float)((a.two_pi_hop * (double)(k - src)) / Nd);
                }
            } else if (k < hs) {
                const int src = __float2int_rn((float)k * a.freq_comp);
                mg = smg[src];
                p = sph[src] + (float)((a.two_pi_hop * (double)(k - src)) / Nd);
            } else {
                mg = smg[k];
                p = sph[k];
            }
            mg *= a.fixed_gain;
        } else {
            mg = vocoder_mag(mag, a.cmag + (int64_t)cslot * tb.HP, k, hs, a.voc_band_len);
            p = sph[k];
        }
        return to_cartesian(mg, p);
    };
    if (plain) {
        // The phases of the plain modes are wrapped (or a small multiple of a wrapped phase, core mode 2): the short
        // sine / cosine of pv_sincos.h serves them; one wave-uniform test of the run's sixteen phases keeps the
        // device library's for anything beyond its argument bound (and for NaN, which fails the comparison).
        float pmax = __builtin_fabsf(pnyq);
#pragma unroll
        for (int q = 0; q < QB; ++q)
            pmax = __builtin_fmaxf(__builtin_fmaxf(pmax, pv_max3_abs(base[q].x, base[q].y, base[q].z)), __builtin_fabsf(base[q].w));
        auto to_cartesian_small = [&](float mg, float p) -> cf {
            mg *= a.inv_n;
            float sn, cs;
            if (kFast) { // |p| is a few turns at most: v_sin_f32 / v_cos_f32 take turns, valid to +-256
                const float rev = p * 0.15915494309189535f;
                sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
            } else {
                pv_sincos_small(p, sn, cs);
            }
            return cf{mg * cs, mg * sn};
        };
        if (kFast || __builtin_amdgcn_ballot_w64(!(pmax <= PV_SINCOS_MAX_ARG)) == 0) {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                if (q + QH < QB) mreg[q + QH] = *reinterpret_cast<const float4 *>(mag + 4 * (lane + 64 * (q + QH)));
                xs[q][0] = to_cartesian_small(mreg[q].x, base[q].x);
                xs[q][1] = to_cartesian_small(mreg[q].y, base[q].y);
                xs[q][2] = to_cartesian_small(mreg[q].z, base[q].z);
                xs[q][3] = to_cartesian_small(mreg[q].w, base[q].w);
            }
            if (lane == 0) xnyq = to_cartesian_small(mnyq, pnyq);
        } else {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                if (q + QH < QB) mreg[q + QH] = *reinterpret_cast<const float4 *>(mag + 4 * (lane + 64 * (q + QH)));
                xs[q][0] = to_cartesian(mreg[q].x, base[q].x);
                xs[q][1] = to_cartesian(mreg[q].y, base[q].y);
                xs[q][2] = to_cartesian(mreg[q].z, base[q].z);
                xs[q][3] = to_cartesian(mreg[q].w, base[q].w);
            }
            if (lane == 0) xnyq = to_cartesian(mnyq, pnyq);
        }
    } else {
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) xs[q][c] = spectrum_bin(4 * (lane + 64 * q) + c);
        if (lane == 0) xnyq = spectrum_bin(hs);
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        float4 *dst = reinterpret_cast<float4 *>(lds + 4 * (lane + 64 * q));
        dst[0] = make_float4(xs[q][0].x, xs[q][0].y, xs[q][1].x, xs[q][1].y);
        dst[1] = make_float4(xs[q][2].x, xs[q][2].y, xs[q][3].x, xs[q][3].y);
    }
    if (lane == 0) lds[hs] = xnyq;
    wave_sync();

    // 3. kiss_fftri pre-pass (kiss_fftr.c:134-157): pairs (k, NC-k) into registers, then scattered into
    //    butterfly order.  k = lane + 64 j; NC - k = 64 * (JB - j - (lane != 0)) + ((64 - lane) & 63).
    constexpr int J = NC / 128;
    constexpr bool kRoomy = NC <= 1024; // registers to spare for the earlier of two possible issue points
    cf sw[J];
    if (kRoomy) {
#pragma unroll
        for (int j = 0; j < J; ++j) sw[j] = stw[lane + 64 * j];
    }
    const cf swmid = stw[NC / 2];
    WfTw<W> T0, T1, T2;
    WfTwRaw<W, 1> raw1;
    if (kRoomy) {
        if (!(kFast && wf_pass_all_const<W, 0>())) wf_load_pass_tw<W, 0>(T0, lane, tw); // (fast: literal twiddles)
        wf_fetch_pass_tw<W, 1>(raw1, lane, twl);
    }
    cf pa[J], pb[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = lane + 64 * j;
        pa[j] = lds[k];
        pb[j] = lds[NC - k];
    }
    const cf pmid = lds[NC / 2];
    wave_sync();
    const int e_lane = wf_e_of_src<W>(lane);             // low six source bits
    const int e_lane2 = wf_e_of_src<W>((64 - lane) & 63);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (j == 0 && lane == 0) {
            lds[W::pad(wf_e_of_src<W>(0))] = cf{pa[0].x + pb[0].x, pa[0].x - pb[0].x};
        } else {
            const cf fk = pa[j];
            const cf fnkc = cf{pb[j].x, -pb[j].y};
            const cf fek = wf_add(fk, fnkc);
            const cf tq = wf_sub(fk, fnkc);
            const cf fok = kFast ? wf_cmul_fma(tq, kRoomy ? sw[j] : stw[lane + 64 * j])
                                 : wf_cmul(tq, kRoomy ? sw[j] : stw[lane + 64 * j]);
            const cf u = wf_add(fek, fok);
            cf vv = wf_sub(fek, fok);
            vv.y = vv.y * -1.f;
            const int e1 = e_lane | wf_e_of_src<W>(64 * j);
            const int hi2 = lane != 0 ? wf_e_of_src<W>(64 * (JB - j - 1)) : wf_e_of_src<W>((64 * (JB - j)) & (NC - 1));
            const int e2 = e_lane2 | hi2;
            lds[W::pad(e1)] = u;
            lds[W::pad(e2)] = vv;
        }
    }
    if (lane == 0) { // k == NC/2: only the second assignment survives
        const cf fk = pmid;
        const cf fnkc = cf{pmid.x, -pmid.y};
        const cf fek = wf_add(fk, fnkc);
        const cf tq = wf_sub(fk, fnkc);
        const cf fok = wf_cmul(tq, swmid);
        cf vv = wf_sub(fek, fok);
        vv.y = vv.y * -1.f;
        lds[W::pad(wf_e_of_src<W>(NC / 2))] = vv;
    }
    wave_sync();

    // 4. inverse complex FFT.  With registers to spare (N = 2048: 125 VGPRs, four waves per SIMD) the twiddles
    //    are fetched a pass ahead and the synthesis window before the frame's first store (see the analysis
    //    kernel); at N = 4096 the 32 values per lane leave no room for that and every load sits at its use.
    cf v[R];
    {
        const int lp = wf_lane_part<W>(0, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = lds[W::pad(lp | wf_reg_part<W>(0, r))];
    }
    const int fslot = (int)(t & (int64_t)(a.FR - 1));
    float *__restrict__ out = a.frames + ((int64_t)row * a.FR + fslot) * N;
    const float *__restrict__ w = tb.window;
    if constexpr (kRoomy) {
        const int lp0 = wf_lane_part<W>(0, lane), lp1 = wf_lane_part<W>(1, lane), lp2 = wf_lane_part<W>(2, lane);
        wave_sync();
        if (kFast) wf_apply_pass_stages_fast<W, 0, true>(v, T0);
        else wf_apply_pass_stages<W, 0, true>(v, T0);
        wf_store<W, 0>(lds, v, lp0);
        wave_sync();
        wf_load<W, 1>(lds, v, lp1);
        wf_unpack_pass_tw<W, 1>(T1, raw1);
        if (kFast) wf_apply_pass_stages_fast<W, 1, true>(v, T1);
        else wf_apply_pass_stages<W, 1, true>(v, T1);
        WfTwRaw<W, 2> raw2;
        wf_fetch_pass_tw<W, 2>(raw2, lane, twl);
        wf_store<W, 1>(lds, v, lp1);
        wave_sync();
        wf_load<W, 2>(lds, v, lp2);
        wf_unpack_pass_tw<W, 2>(T2, raw2);
        if (kFast) wf_apply_pass_stages_fast<W, 2, true>(v, T2);
        else wf_apply_pass_stages<W, 2, true>(v, T2);
        constexpr int PL = W::NPASS - 1; // the last pass
        const int lpl = wf_lane_part<W>(PL, lane);
        if constexpr (W::NPASS == 4) { // (512 points: the last radix-4 stage is a pass of its own)
            WfTwRaw<W, 3> raw3;
            wf_fetch_pass_tw<W, 3>(raw3, lane, twl);
            wf_store<W, 2>(lds, v, lp2);
            wave_sync();
            wf_load<W, 3>(lds, v, lpl);
            wf_unpack_pass_tw<W, 3>(T1, raw3);
            if (kFast) wf_apply_pass_stages_fast<W, 3, true>(v, T1);
            else wf_apply_pass_stages<W, 3, true>(v, T1);
        }
        if constexpr (kSink == 1) {
            wf_store<W, PL>(lds, v, lpl);
            wave_sync();
            return;
        }
        float4 ww[NC / 128];
#pragma unroll
        for (int j = 0; j < NC / 128; ++j) ww[j] = *reinterpret_cast<const float4 *>(w + 4 * (lane + 64 * j));
        wf_store<W, PL>(lds, v, lpl);
        wave_sync();
        // 5. ifftshift + synthesis window (phasevocoderimpl.h:183-198): four consecutive samples per lane and
        //    store (the stores are issue-bound: half as many 16-byte ones beat twice as many 8-byte ones)
#pragma unroll
        for (int j = 0; j < NC / 128; ++j) {
            const int i = 4 * (lane + 64 * j);        // output sample index
            const int e = ((i + hs) & (N - 1)) >> 1;  // even: elements e, e+1 hold samples i+hs .. i+hs+3
            const cf z0 = lds[W::pad(e)], z1 = lds[W::pad(e) + 1]; // same group of 16, so adjacent after padding
            *reinterpret_cast<float4 *>(out + i) =
                make_float4(z0.x * ww[j].x, z0.y * ww[j].y, z1.x * ww[j].z, z1.y * ww[j].w);
        }
    } else {
        // no room to fetch a pass ahead, but the lane-major tables still turn each pass's twiddle gathers into a
        // few contiguous 16-byte loads
        wave_sync();
        if (kFast && wf_pass_all_const<W, 0>()) { // every twiddle of the pass is a literal: nothing to fetch
            wf_apply_pass_stages_fast<W, 0, true>(v, T0);
            wf_store<W, 0>(lds, v, wf_lane_part<W>(0, lane));
        } else {
            wf_fft_pass<W, 0, true>(v, lane, lds, tw);
        }
        wave_sync();
        {
            WfTwRaw<W, 1> r1;
            wf_fetch_pass_tw<W, 1>(r1, lane, twl);
            wf_unpack_pass_tw<W, 1>(T1, r1);
            if (kFast) {
                wf_load<W, 1>(lds, v, wf_lane_part<W>(1, lane));
                wf_apply_pass_stages_fast<W, 1, true>(v, T1);
                wf_store<W, 1>(lds, v, wf_lane_part<W>(1, lane));
            } else {
                wf_fft_pass_tw<W, 1, true>(v, lane, lds, T1);
            }
        }
        wave_sync();
        {
            WfTwRaw<W, 2> r2;
            wf_fetch_pass_tw<W, 2>(r2, lane, twl);
            wf_unpack_pass_tw<W, 2>(T2, r2);
            if (kFast) {
                wf_load<W, 2>(lds, v, wf_lane_part<W>(2, lane));
                wf_apply_pass_stages_fast<W, 2, true>(v, T2);
                wf_store<W, 2>(lds, v, wf_lane_part<W>(2, lane));
            } else {
                wf_fft_pass_tw<W, 2, true>(v, lane, lds, T2);
            }
        }
        if constexpr (kSink == 1) {
            wave_sync();
            return;
        }
        // (v is dead now: the whole window fits in its registers, fetched before the frame's first store, and the
        // frame leaves 16 bytes per lane like the other variant's)
        float4 ww[NC / 128];
#pragma unroll
        for (int j = 0; j < NC / 128; ++j) ww[j] = *reinterpret_cast<const float4 *>(w + 4 * (lane + 64 * j));
        wave_sync();
#pragma unroll
        for (int j = 0; j < NC / 128; ++j) {
            const int i = 4 * (lane + 64 * j);
            const int e = ((i + hs) & (N - 1)) >> 1;
            const cf z0 = lds[W::pad(e)], z1 = lds[W::pad(e) + 1];
            *reinterpret_cast<float4 *>(out + i) =
                make_float4(z0.x * ww[j].x, z0.y * ww[j].y, z1.x * ww[j].z, z1.y * ww[j].w);
        }
    }
}

// (N = 2048 is asked to fit four waves per SIMD, 128 VGPRs; the compiler gets there without spilling)
template <int NC, int WPB, int kPlainCore = -1>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(NC == 1024 ? 4 : 1))) void
pv_synth_wave_kernel(const SynthArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    cf *lds = reinterpret_cast<cf *>(smem_raw) + (threadIdx.x >> 6) * WF<NC>::LDS_CF;
    int row, tl;
    if (!block_to_row_slice_w<WPB>(a.Tn, a.rows, row, tl)) return; // wave-uniform
    synth_wave_role<NC, kPlainCore>(a, row, tl, lds);
}

// AUDIOMOD_PV_SYNTH_GENERIC=1: every mode through the all-modes kernel (tests compare it with the specialisation)
static bool synth_generic_only() {
    static const bool on = [] {
        const char *e = getenv("AUDIOMOD_PV_SYNTH_GENERIC");
        return e && atoi(e) != 0;
    }();
    return on;
}

void launch_synth(const SynthArgs &a, hipStream_t st) {
    if (a.tb.nc == 512 || a.tb.nc == 256) { // fft 1024 / 512 (round 3): the phase-locked plain specialisation, else the all-modes kernel
        constexpr int WPB = 1;
        const int grid = 8 * ((a.rows + 7) / 8) * ((a.Tn + WPB - 1) / WPB);
        const bool plain = !a.do_freq_comp && a.voc_band_len < 0 && !a.robotic && !a.passthru && !a.whisper &&
                           !synth_generic_only();
        if (a.tb.nc == 512) {
            const size_t lds = WPB * WF<512>::LDS_CF * sizeof(cf);
            if (plain && a.coremode == 1) hipLaunchKernelGGL((pv_synth_wave_kernel<512, WPB, 1>), dim3(grid), dim3(64 * WPB), lds, st, a);
            else hipLaunchKernelGGL((pv_synth_wave_kernel<512, WPB>), dim3(grid), dim3(64 * WPB), lds, st, a);
        } else {
            const size_t lds = WPB * WF<256>::LDS_CF * sizeof(cf);
            if (plain && a.coremode == 1) hipLaunchKernelGGL((pv_synth_wave_kernel<256, WPB, 1>), dim3(grid), dim3(64 * WPB), lds, st, a);
            else hipLaunchKernelGGL((pv_synth_wave_kernel<256, WPB>), dim3(grid), dim3(64 * WPB), lds, st, a);
        }
        return;
    }
    if (a.tb.nc == 1024 || a.tb.nc == 2048) {
        if (a.tb.nc == 1024) {
            constexpr int WPB = 1;
            const int grid = 8 * ((a.rows + 7) / 8) * ((a.Tn + WPB - 1) / WPB);
            const bool plain = !a.do_freq_comp && a.voc_band_len < 0 && !a.robotic && !a.passthru && !a.whisper &&
                               !synth_generic_only();
            if (plain && a.coremode >= 0 && a.coremode <= 2) {
                const size_t lds = WPB * WF<1024>::LDS_CF * sizeof(cf);
                if (a.coremode == 1) hipLaunchKernelGGL((pv_synth_wave_kernel<1024, WPB, 1>), dim3(grid), dim3(64 * WPB), lds, st, a);
                else if (a.coremode == 0) hipLaunchKernelGGL((pv_synth_wave_kernel<1024, WPB, 0>), dim3(grid), dim3(64 * WPB), lds, st, a);
                else hipLaunchKernelGGL((pv_synth_wave_kernel<1024, WPB, 2>), dim3(grid), dim3(64 * WPB), lds, st, a);
                return;
            }
            static unsigned long long big1 = 0;
            allow_big_lds_dev(pv_synth_wave_kernel<1024, WPB>, big1);
            hipLaunchKernelGGL((pv_synth_wave_kernel<1024, WPB>), dim3(grid), dim3(64 * WPB),
                               WPB * WF<1024>::LDS_CF * sizeof(cf), st, a);
        } else {
            constexpr int WPB = 1;
            const int grid = 8 * ((a.rows + 7) / 8) * ((a.Tn + WPB - 1) / WPB);
            const bool plain = !a.do_freq_comp && a.voc_band_len < 0 && !a.robotic && !a.passthru && !a.whisper &&
                               !synth_generic_only();
            if (plain && a.coremode >= 0 && a.coremode <= 2) {
                const size_t lds = WPB * WF<2048>::LDS_CF * sizeof(cf);
                if (a.coremode == 1) hipLaunchKernelGGL((pv_synth_wave_kernel<2048, WPB, 1>), dim3(grid), dim3(64 * WPB), lds, st, a);
                else if (a.coremode == 0) hipLaunchKernelGGL((pv_synth_wave_kernel<2048, WPB, 0>), dim3(grid), dim3(64 * WPB), lds, st, a);
                else hipLaunchKernelGGL((pv_synth_wave_kernel<2048, WPB, 2>), dim3(grid), dim3(64 * WPB), lds, st, a);
                return;
            }
            static unsigned long long big2 = 0;
            allow_big_lds_dev(pv_synth_wave_kernel<2048, WPB>, big2);
            hipLaunchKernelGGL((pv_synth_wave_kernel<2048, WPB>), dim3(grid), dim3(64 * WPB),
                               WPB * WF<2048>::LDS_CF * sizeof(cf), st, a);
        }
        return;
    }
    const int grid = 8 * ((a.rows + 7) / 8) * a.Tn;
    const size_t lds = (size_t)(2 * a.tb.nc + 1) * sizeof(float2) + sizeof(float) * (a.tb.hs + 4 + a.PKP) +
                       sizeof(uint16_t) * a.PKP;
    static unsigned long long big = 0;
    allow_big_lds_dev(pv_synth_kernel, big);
    hipLaunchKernelGGL(pv_synth_kernel, dim3(grid), dim3(generic_fft_threads(a.tb.nc)), lds, st, a);
}

// --------------------------------------------------------------------------------------------
// Cepstral formant shift of a slice's magnitudes (extension mode; the reference's formantShiftSlice,
// phasevocoderprocess.cc:925-999, with D_KISSFFT::inverseCepstral FFT.cc:2723-2733 and ::forward :2606-2610 --
// unreachable upstream, restated in oracle/pv_oracle.c formant_shift and pinned there on the real function):
//   cep = kiss_fftri(log(mag + 1e-6));  keep the first 60 quefrencies (ends halved), times 1/N;
//   envelope = exp(Re kiss_fftr(cep));  mag = mag / envelope * envelope[lrint(k * env_comp)]
// One wave per (row, slice), both transforms through the wave-FFT core in the wave's LDS region.
// --------------------------------------------------------------------------------------------
template <int NC>
__device__ __forceinline__ void cepstral_wave_role(const CepstralArgs &a, const int row, const int tl, cf *lds) {
    using W = WF<NC>;
    constexpr int hs = NC, R = W::R, J = NC / 128, QB = NC / 256, kCut = 60;
    const int lane = threadIdx.x & 63;
    const DevTables &tb = a.tb;
    const int64_t plane = (int64_t)row * a.TR + ring_slot(a.s0, tl, a.TR);
    float *__restrict__ mag = a.mag + plane * tb.HP;
    const cf *__restrict__ twi = reinterpret_cast<const cf *>(tb.tw_inv);
    const cf *__restrict__ twf = reinterpret_cast<const cf *>(tb.tw_fwd);
    const cf *__restrict__ sti = reinterpret_cast<const cf *>(tb.st_inv);
    const cf *__restrict__ stf = reinterpret_cast<const cf *>(tb.st_fwd);

    // magnitudes: runs of four consecutive bins per lane (kept in registers for the end), and the Nyquist bin
    float4 m4[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) m4[q] = *reinterpret_cast<const float4 *>(mag + 4 * (lane + 64 * q));
    const float mny = mag[hs];
    // X[k] = (logf(mag[k] + 1e-6), 0) in natural order
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        float4 *dst = reinterpret_cast<float4 *>(lds + 4 * (lane + 64 * q));
        dst[0] = make_float4(logf(m4[q].x + 0.000001f), 0.f, logf(m4[q].y + 0.000001f), 0.f);
        dst[1] = make_float4(logf(m4[q].z + 0.000001f), 0.f, logf(m4[q].w + 0.000001f), 0.f);
    }
    const cf xnyq = cf{logf(mny + 0.000001f), 0.f};
    wave_sync();

    // kiss_fftri pre-pass (kiss_fftr.c:134-157), as in the synthesis kernel
    cf v[R];
    {
        cf pa[J], pb[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int k = lane + 64 * j;
            pa[j] = lds[k];
            pb[j] = lds[(NC - k) & (NC - 1)];
        }
        if (lane == 0) pb[0] = xnyq;
        const cf pmid = lds[NC / 2];
        wave_sync();
        const int e_lane = wf_e_of_src<W>(lane);
        const int e_lane2 = wf_e_of_src<W>((64 - lane) & 63);
        constexpr int JB = NC / 64;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (j == 0 && lane == 0) {
                lds[W::pad(wf_e_of_src<W>(0))] = cf{pa[0].x + pb[0].x, pa[0].x - pb[0].x};
            } else {
                const cf fk = pa[j];
                const cf fnkc = cf{pb[j].x, -pb[j].y};
                const cf fek = wf_add(fk, fnkc);
                const cf tq = wf_sub(fk, fnkc);
                const cf fok = wf_cmul(tq, sti[lane + 64 * j]);
                const cf u = wf_add(fek, fok);
                cf vv = wf_sub(fek, fok);
                vv.y = vv.y * -1.f;
                const int e1 = e_lane | wf_e_of_src<W>(64 * j);
                const int hi2 = lane != 0 ? wf_e_of_src<W>(64 * (JB - j - 1)) : wf_e_of_src<W>((64 * (JB - j)) & (NC - 1));
                const int e2 = e_lane2 | hi2;
                lds[W::pad(e1)] = u;
                lds[W::pad(e2)] = vv;
            }
        }
        if (lane == 0) {
            const cf fk = pmid;
            const cf fnkc = cf{pmid.x, -pmid.y};
            const cf fek = wf_add(fk, fnkc);
            const cf tq = wf_sub(fk, fnkc);
            const cf fok = wf_cmul(tq, sti[NC / 2]);
            cf vv = wf_sub(fek, fok);
            vv.y = vv.y * -1.f;
            lds[W::pad(wf_e_of_src<W>(NC / 2))] = vv;
        }
        wave_sync();
        const int lp = wf_lane_part<W>(0, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = lds[W::pad(lp | wf_reg_part<W>(0, r))];
    }
    wave_sync();
    wf_fft_pass<W, 0, true>(v, lane, lds, twi);
    wave_sync();
    wf_fft_pass<W, 1, true>(v, lane, lds, twi);
    wave_sync();
    wf_fft_pass<W, 2, true>(v, lane, lds, twi);
    wave_sync();

    // lifter: element e of the transform holds cep[2e], cep[2e+1]; only the first 60 survive
    cf z = cf{0.f, 0.f};
    if (lane < kCut / 2) {
        z = lds[W::pad(lane)];
        if (lane == 0) z.x = z.x / 2;
        if (lane == kCut / 2 - 1) z.y = z.y / 2;
        z.x = z.x * a.inv_n;
        z.y = z.y * a.inv_n;
    }
    wave_sync();
    cf *zl = lds + W::LDS_CF - 32; // scratch at the end of the region (consumed before the next pass writes there)
    if (lane < 32) zl[lane] = z;
    wave_sync();
    // forward transform of the zero-extended sequence: pass-0 layout straight from the 32 survivors
    {
        const int lp = wf_lane_part<W>(0, lane);
        const int lsrc = wf_src_of<W>(lp);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int src = lsrc | wf_src_of_const<W>(wf_reg_part<W>(0, r));
            const cf t = zl[src & 31];
            v[r] = src < 32 ? t : cf{0.f, 0.f};
        }
    }
    wave_sync();
    wf_fft_pass<W, 0, false>(v, lane, lds, twf);
    wave_sync();
    wf_fft_pass<W, 1, false>(v, lane, lds, twf);
    wave_sync();
    wf_fft_pass<W, 2, false>(v, lane, lds, twf);
    wave_sync();

    // real parts of the real-FFT split (kiss_fftr.c:91-120), exponentiated: the spectral envelope
    float elo[J], ehi[J], emid = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = lane + 64 * j;
        if (j == 0 && lane == 0) {
            const cf tdc = lds[W::pad(0)];
            elo[0] = expf(tdc.x + tdc.y);
            ehi[0] = expf(tdc.x - tdc.y);
        } else {
            const cf fpk = lds[W::pad(k)];
            const cf q = lds[W::pad(NC - k)];
            const cf fpnk = cf{q.x, -q.y};
            const cf f1k = wf_add(fpk, fpnk);
            const cf f2k = wf_sub(fpk, fpnk);
            const cf tq = wf_cmul(f2k, stf[k]);
            elo[j] = expf((f1k.x + tq.x) * 0.5f);
            ehi[j] = expf((f1k.x - tq.x) * 0.5f);
        }
    }
    if (lane == 0) {
        const cf fpk = lds[W::pad(NC / 2)];
        const cf fpnk = cf{fpk.x, -fpk.y};
        const cf f1k = wf_add(fpk, fpnk);
        const cf f2k = wf_sub(fpk, fpnk);
        const cf tq = wf_cmul(f2k, stf[NC / 2]);
        emid = expf((f1k.x - tq.x) * 0.5f);
    }
    wave_sync();
    float *senv = reinterpret_cast<float *>(lds); // [hs + 1]
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = lane + 64 * j;
        if (j == 0 && lane == 0) {
            senv[0] = elo[0];
            senv[NC] = ehi[0];
        } else {
            senv[k] = elo[j];
            senv[NC - k] = ehi[j];
        }
    }
    if (lane == 0) senv[NC / 2] = emid;
    wave_sync();

    // whiten by the envelope, re-colour by the envelope read at lrint(k * env_comp)
    auto shifted = [&](int k) -> float {
        if (a.env_comp > 1.0f) {
            const int src = __float2int_rn((float)k * a.env_comp);
            return src > hs ? 0.f : senv[src];
        }
        if (k == hs) return senv[hs]; // the downward loop of the reference never touches the Nyquist bin
        return senv[__float2int_rn((float)k * a.env_comp)];
    };
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int k0 = 4 * (lane + 64 * q);
        const float4 e4 = *reinterpret_cast<const float4 *>(senv + k0);
        float4 o;
        o.x = (m4[q].x / e4.x) * shifted(k0);
        o.y = (m4[q].y / e4.y) * shifted(k0 + 1);
        o.z = (m4[q].z / e4.z) * shifted(k0 + 2);
        o.w = (m4[q].w / e4.w) * shifted(k0 + 3);
        *reinterpret_cast<float4 *>(mag + k0) = o;
    }
    if (lane == 0) mag[hs] = (mny / senv[hs]) * shifted(hs);
}

template <int NC> __global__ __launch_bounds__(64) void pv_cepstral_wave_kernel(const CepstralArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    int row, tl;
    if (!block_to_row_slice_w<1>(a.Tn, a.rows, row, tl)) return; // wave-uniform
    cepstral_wave_role<NC>(a, row, tl, reinterpret_cast<cf *>(smem_raw));
}

// Any FFT size (round 2): one workgroup per (row, slice), both transforms through the generic LDS butterfly stages --
// the same steps as the wave version above, the same arithmetic as the reference's formantShiftSlice
// (phasevocoderprocess.cc:925-999), which is size-agnostic.
__global__ __launch_bounds__(kFftThreads) void pv_cepstral_kernel(const CepstralArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    const DevTables &tb = a.tb;
    const int hs = tb.hs, nc = tb.nc, nt = blockDim.x;
    constexpr int kCut = 60;
    float2 *buf = reinterpret_cast<float2 *>(smem_raw); // [nc]      butterfly array
    float2 *X = buf + nc;                               // [nc + 1]  log-magnitude spectrum
    float *senv = reinterpret_cast<float *>(X + nc + 1); // [hs + 1] envelope
    float *scep = senv + hs + 4;                        // [64]      the surviving quefrencies
    int row, tl;
    if (!block_to_row_slice(a.Tn, a.rows, row, tl)) return;
    const int64_t plane = (int64_t)row * a.TR + ring_slot(a.s0, tl, a.TR);
    float *__restrict__ mag = a.mag + plane * tb.HP;
    for (int k = threadIdx.x; k <= hs; k += nt) X[k] = make_float2(logf(mag[k] + 0.000001f), 0.f);
    __syncthreads();
    // kiss_fftri pre-pass (kiss_fftr.c:134-157) scattered into butterfly order, inverse stages: buf = cep pairs
    for (int k = threadIdx.x; k <= nc / 2; k += nt) {
        if (k == 0) {
            buf[tb.iperm[0]] = make_float2(X[0].x + X[nc].x, X[0].x - X[nc].x);
        } else {
            const float2 fk = X[k];
            const float2 q = X[nc - k];
            const float2 fnkc = make_float2(q.x, -q.y);
            const float2 fek = cadd(fk, fnkc);
            const float2 tq = csub(fk, fnkc);
            const float2 fok = cmul(tq, tb.st_inv[k]);
            const float2 u = cadd(fek, fok);
            float2 v = csub(fek, fok);
            v.y = v.y * -1.f;
            if (k != nc - k) buf[tb.iperm[k]] = u;
            buf[tb.iperm[nc - k]] = v;
        }
    }
    __syncthreads();
    fft_stages<true>(buf, tb, tb.tw_inv);
    // lifter: the first 60 quefrencies survive, the two ends halved, all times 1 / N (:952-960)
    if (threadIdx.x < 64) {
        const float *cep = reinterpret_cast<const float *>(buf);
        float c = threadIdx.x < kCut ? cep[threadIdx.x] : 0.f;
        if (threadIdx.x == 0 || threadIdx.x == kCut - 1) c = c / 2;
        scep[threadIdx.x] = threadIdx.x < kCut ? c * a.inv_n : 0.f;
    }
    __syncthreads();
    // forward real transform of the zero-extended sequence (kiss_fftr.c:67-121): pairs in butterfly order
    for (int j = threadIdx.x; j < nc; j += nt) {
        const int src = tb.perm[j];
        buf[j] = src < 32 ? make_float2(scep[2 * src], scep[2 * src + 1]) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    fft_stages<false>(buf, tb, tb.tw_fwd);
    for (int k = threadIdx.x; k <= nc / 2; k += nt) {
        if (k == 0) {
            const float2 tdc = buf[0];
            senv[0] = expf(tdc.x + tdc.y);
            senv[nc] = expf(tdc.x - tdc.y);
        } else {
            const float2 fpk = buf[k];
            const float2 q = buf[nc - k];
            const float2 fpnk = make_float2(q.x, -q.y);
            const float2 f1k = cadd(fpk, fpnk);
            const float2 f2k = csub(fpk, fpnk);
            const float2 tq = cmul(f2k, tb.st_fwd[k]);
            if (k != nc - k) senv[k] = expf((f1k.x + tq.x) * 0.5f);
            senv[nc - k] = expf((f1k.x - tq.x) * 0.5f);
        }
    }
    __syncthreads();
    // whiten by the envelope, re-colour by the envelope read at lrint(k * env_comp) (:962-998)
    for (int k = threadIdx.x; k <= hs; k += nt) {
        float sh;
        if (a.env_comp > 1.0f) {
            const int src = __float2int_rn((float)k * a.env_comp);
            sh = src > hs ? 0.f : senv[src];
        } else if (k == hs) {
            sh = senv[hs]; // the downward loop of the reference never touches the Nyquist bin
        } else {
            sh = senv[__float2int_rn((float)k * a.env_comp)];
        }
        mag[k] = (mag[k] / senv[k]) * sh;
    }
}

void launch_cepstral(const CepstralArgs &a, hipStream_t st) {
    const int grid = 8 * ((a.rows + 7) / 8) * a.Tn;
    if (a.tb.nc != 1024 && a.tb.nc != 2048) {
        const size_t lds = (size_t)(2 * a.tb.nc + 1) * sizeof(float2) + sizeof(float) * (a.tb.hs + 4 + 64);
        static unsigned long long big = 0;
        allow_big_lds_dev(pv_cepstral_kernel, big);
        hipLaunchKernelGGL(pv_cepstral_kernel, dim3(grid), dim3(generic_fft_threads(a.tb.nc)), lds, st, a);
        return;
    }
    if (a.tb.nc == 1024) {
        hipLaunchKernelGGL((pv_cepstral_wave_kernel<1024>), dim3(grid), dim3(64), WF<1024>::LDS_CF * sizeof(cf), st, a);
    } else {
        hipLaunchKernelGGL((pv_cepstral_wave_kernel<2048>), dim3(grid), dim3(64), WF<2048>::LDS_CF * sizeof(cf), st, a);
    }
}

// --------------------------------------------------------------------------------------------
// overlap-add + normalise + resample
// --------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// NR rows (stream-channels) per workgroup.  All rows share one schedule, so a tile's descriptor, its frame
// offsets, its output table and -- in the resampler -- the coefficient row of an output are the same for every row:
// with NR = 2 a workgroup fetches them once for two rows, each resampler thread reads a coefficient once for two
// outputs (the loop is bound by LDS reads), and the dependent chain descriptor -> offsets -> frame gather ->
// resample, which at full occupancy is what the kernel waits on, is paid once for twice the work.
// kRes: -1 = resampling mode read from the arguments; 0 = none, 1 = direct sinc table, 2 = cubic-interpolated table
// (the argument's flags become compile-time constants)
template <int NR, int kRes = -1>
__device__ __forceinline__ void ola_role(const OlaArgs &a_in, const int tile_i, const int row0, char *smem_raw) {
    OlaArgs a = a_in;
    if (kRes >= 0) a.resample = kRes != 0, a.interp = kRes == 2;
    // interpolated mode: coefficient table expanded per sub-sample offset, tab4[off][j] = the four taps
    // sinc[4 + (j+1)*ov - off + {-2,-1,0,1}] of resampler_basic_interpolate_single (resample.c:494-535) as one
    // aligned float4; rows are padded to NF+1 slots so the (at most ov) distinct rows a wave reads in one
    // ds_read_b128 land on different banks.  direct mode: the sinc table as it is.
    float4 *tab4 = reinterpret_cast<float4 *>(smem_raw);
    float *stab = reinterpret_cast<float *>(smem_raw);
    float *ola = reinterpret_cast<float *>(smem_raw + a.tab_bytes);   // [NR][lds_floats]
    int *sP = reinterpret_cast<int *>(ola + NR * a.lds_floats);       // [kMaxTileFrames] P_t - n_lo
    const int nt = blockDim.x, tid = threadIdx.x;
    const OlaTile tile = a.tiles[tile_i];
    const int N = a.N, NF = a.filt_len;

    // position of this thread's output in the tile, its sub-sample offset and interpolation fraction: planned
    // on the host (pv_engine.cc build_tiles), one 8-byte entry per output, fetched ahead of the gather
    const float *__restrict__ wacc = a.wacc + (int64_t)tile_i * a.wacc_pitch;
    uint2 oe = make_uint2(0u, 0u);
    if (a.resample && tid < tile.kcnt) oe = reinterpret_cast<const uint2 *>(wacc + a.otab_off)[tid];
    if (tid < tile.t_cnt) sP[tid] = (int)(a.P[tile.p_off + tid] - tile.n_lo);
    auto copy_table = [&](int first, int stride) {
        if (!a.resample) return;
        if (a.interp) {
            const int cnt = a.oversample * (NF + 1);
            for (int i = first; i < cnt; i += stride) tab4[i] = a.tab4[i];
        } else {
            for (int i = first; i < a.sinc_len; i += stride) stab[i] = a.sinc[i];
        }
    };
    // y[n] = (sum_t frame_t[n - P_t]) / (delta[n] + sum_t gain*w[n - P_t]), ascending t, starting from 0.0f
    // (== outputAccumulator / windowAccumulator at the moment writeSlice divides them).  The denominator is
    // data-independent: the host planner evaluates it once per tile with the same float arithmetic
    // (pv_engine.cc build_tiles) and every (stream, channel) row reuses it.  Frames that do not cover n
    // contribute an exact +0.0f (adding +0 never changes a sum that started at +0), so the loads are
    // unconditional on a clamped address: no divergent branch, and the compiler can batch them.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool quads = NR == 2 && nt == 256 && tile.n_cnt <= 4 * 126 && (N & 3) == 0;
    copy_table(tid, nt);
    __syncthreads();
    if (quads) {
        // The gather is bound by the number of load instructions and by their latency: a lane owns four consecutive
        // tile samples and fetches 16 bytes per frame.  A frame starts anywhere, so its samples for those four sit
        // in two neighbouring aligned 16-byte pieces: the lane loads the lower one and takes the upper one from the
        // next lane (DPP wave shift; lane 63 of a wave only serves as its lane 62's neighbour, so a wave covers
        // 63 quads and a pair of waves 126).  Waves 0-1 gather the first row, waves 2-3 the second.
        const int r = wave >> 1;
        const bool row_ok = row0 + r < a.rows;
        const float *__restrict__ fr = a.frames + (int64_t)(row_ok ? row0 + r : row0) * a.FR * N;
        const int u = 63 * (wave & 1) + lane;
        // quads past the tile's last sample are never used (one more serves as the last one's DPP neighbour): they
        // fetch what that neighbour fetches instead of frame data beyond the tile -- the 126 quads of a wave pair
        // cover 504 samples, a tile needs ~400, and the difference was a fifth of the kernel's HBM reads
        const int qmax = (tile.n_cnt + 3) >> 2;
        const int ua = u < qmax ? u : qmax;
        const float4 *__restrict__ fr4 = reinterpret_cast<const float4 *>(fr);
        const int nq = N >> 2;
        const float4 w4 = reinterpret_cast<const float4 *>(wacc)[4 * u < a.lds_floats ? u : 0];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        constexpr int kF = 4; // frames whose loads are in flight together
        for (int j0 = 0; j0 < tile.t_cnt; j0 += kF) {
            float4 Av[kF];
            int spv[kF];
#pragma unroll
            for (int q = 0; q < kF; ++q) {
                const int j = j0 + q < tile.t_cnt ? j0 + q : tile.t_cnt - 1;
                spv[q] = sP[j];
                const int f0 = 4 * ua - spv[q]; // frame sample under the quad's first tile sample
                const int rr = (-spv[q]) & 3;   // where in its aligned piece that sample sits (wave-uniform)
                const int qf = (f0 - rr) >> 2;
                const int qc = qf < 0 ? 0 : (qf >= nq ? nq - 1 : qf);
                const int slot = (tile.t_first + j) & (a.FR - 1);
                Av[q] = fr4[(int64_t)slot * nq + qc];
            }
#pragma unroll
            for (int q = 0; q < kF; ++q) {
                if (j0 + q >= tile.t_cnt) break; // wave-uniform
                const float4 A = Av[q];
                const int f0 = 4 * u - spv[q];
                const int rr = (-spv[q]) & 3;
                float4 B;
                B.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(A.x), 0x130, 0xf, 0xf, false));
                B.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(A.y), 0x130, 0xf, 0xf, false));
                B.z = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(A.z), 0x130, 0xf, 0xf, false));
                float v0, v1, v2, v3;
                if (rr == 0) {
                    v0 = A.x, v1 = A.y, v2 = A.z, v3 = A.w;
                } else if (rr == 1) {
                    v0 = A.y, v1 = A.z, v2 = A.w, v3 = B.x;
                } else if (rr == 2) {
                    v0 = A.z, v1 = A.w, v2 = B.x, v3 = B.y;
                } else {
                    v0 = A.w, v1 = B.x, v2 = B.y, v3 = B.z;
                }
                a0 += (f0 >= 0 && f0 < N) ? v0 : 0.f;
                a1 += (f0 + 1 >= 0 && f0 + 1 < N) ? v1 : 0.f;
                a2 += (f0 + 2 >= 0 && f0 + 2 < N) ? v2 : 0.f;
                a3 += (f0 + 3 >= 0 && f0 + 3 < N) ? v3 : 0.f;
            }
        }
        if (lane < 63 && 4 * u < a.lds_floats) {
            const int64_t n0 = tile.n_lo + 4 * u;
            float4 y;
            y.x = n0 >= 0 ? a0 / w4.x : 0.f;
            y.y = n0 + 1 >= 0 ? a1 / w4.y : 0.f;
            y.z = n0 + 2 >= 0 ? a2 / w4.z : 0.f;
            y.w = n0 + 3 >= 0 ? a3 / w4.w : 0.f;
            reinterpret_cast<float4 *>(ola + r * a.lds_floats)[u] = y;
        }
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (row0 + r >= a.rows) break; // workgroup-uniform
            const float *__restrict__ fr = a.frames + (int64_t)(row0 + r) * a.FR * N;
            for (int i = tid; i < tile.n_cnt; i += nt) {
                float acc = 0.f;
#pragma unroll 4
                for (int j = 0; j < tile.t_cnt; ++j) {
                    const int off = i - sP[j]; // n - P_t
                    const bool in = off >= 0 && off < N;
                    const int slot = (tile.t_first + j) & (a.FR - 1);
                    const float fv = fr[(int64_t)slot * N + (in ? off : 0)];
                    acc += in ? fv : 0.f;
                }
                ola[r * a.lds_floats + i] = tile.n_lo + i >= 0 ? acc / wacc[i] : 0.f;
            }
        }
    }
    __syncthreads();

    if (tid >= tile.kcnt) return;
    const int64_t k = tile.k0 + tid;
    const int nr = a.rows - row0 < NR ? a.rows - row0 : NR; // rows this workgroup really has (>= 1)
    float *__restrict__ out = a.out + (int64_t)row0 * a.out_stride_row + (k - a.k_base);
    if (!a.resample) {
        for (int r = 0; r < nr; ++r) out[(int64_t)r * a.out_stride_row] = ola[r * a.lds_floats + tid];
        return;
    }
    const float *x = ola + (int)(oe.x & 0xffffu); // tap j = 0
    if (a.interp) {
        const int offset = (int)(oe.x >> 16);
        const float frac = __uint_as_float(oe.y);
        const float4 *__restrict__ T = tab4 + offset * (NF + 1);
        v2f a01[NR], a23[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) a01[r] = v2f{0.f, 0.f}, a23[r] = v2f{0.f, 0.f};
#pragma unroll 4
        for (int j = 0; j < NF; ++j) { // NF is a multiple of 4 (resample.c:687)
            const float4 c = T[j];
            const v2f c01 = {c.x, c.y}, c23 = {c.z, c.w};
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float xv = x[r * a.lds_floats + j]; // a missing second row reads stale LDS: never stored
                const v2f xx = {xv, xv};
                a01[r] += xx * c01; // -ffp-contract=off: separate multiply and add, per element, like the reference
                a23[r] += xx * c23;
            }
        }
        // cubic_coef (resample.c:339-351)
        const float c0 = -0.16667f * frac + 0.16667f * frac * frac * frac;
        const float c1 = frac + 0.5f * frac * frac - 0.5f * frac * frac * frac;
        const float c3 = -0.33333f * frac + 0.5f * frac * frac - 0.16667f * frac * frac * frac;
        const float c2 = (float)(1. - c0 - c1 - c3);
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (r < nr)
                out[(int64_t)r * a.out_stride_row] = (c0 * a01[r].x) + (c1 * a01[r].y) + (c2 * a23[r].x) + (c3 * a23[r].y);
    } else {
        const float *t = stab + (oe.x >> 16) * (uint32_t)NF;
        for (int r = 0; r < nr; ++r) {
            float sum = 0.f;
            for (int j = 0; j < NF; ++j) sum += x[r * a.lds_floats + j] * t[j];
            out[(int64_t)r * a.out_stride_row] = sum;
        }
    }
}

constexpr int kOlaRows = 2; // rows per workgroup of the batch / streaming overlap-add kernel

template <int kRes> __global__ __launch_bounds__(kTileOut) void pv_ola_kernel(const OlaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    ola_role<kOlaRows, kRes>(a, blockIdx.x, blockIdx.y * kOlaRows, smem_raw);
}

size_t ola_lds_bytes(const OlaArgs &a, int rows_per_group) {
    return (size_t)a.tab_bytes + sizeof(float) * (size_t)a.lds_floats * rows_per_group + sizeof(int) * kMaxTileFrames;
}

void launch_ola(const OlaArgs &a, hipStream_t st) {
    const size_t lds = ola_lds_bytes(a, kOlaRows);
    const dim3 grid(a.ntiles, (a.rows + kOlaRows - 1) / kOlaRows);
    static unsigned long long big0 = 0, big1 = 0, big2 = 0;
    if (!a.resample) {
        allow_big_lds_dev(pv_ola_kernel<0>, big0);
        hipLaunchKernelGGL(pv_ola_kernel<0>, grid, dim3(kTileOut), lds, st, a);
    } else if (!a.interp) {
        allow_big_lds_dev(pv_ola_kernel<1>, big1);
        hipLaunchKernelGGL(pv_ola_kernel<1>, grid, dim3(kTileOut), lds, st, a);
    } else {
        allow_big_lds_dev(pv_ola_kernel<2>, big2);
        hipLaunchKernelGGL(pv_ola_kernel<2>, grid, dim3(kTileOut), lds, st, a);
    }
}

// --------------------------------------------------------------------------------------------
// Fused synthesis + overlap-add + resample (pv_kernels.h ChainArgs): the reference's streaming state kept in LDS.
//
//   synthesiseSlice (:1057,1073):  outputAccumulator[0..N) += frame         -> acc ring, frame t at P_t
//   writeSlice (:1157-1194):       acc[0..s) /= wacc[0..s); resample or ring-write; shift both by s
//                                  -> finalise [P_t, P_t + s): divide by the host-planned window sum (the
//                                     denominator is data-independent), zero the slots; the normalised samples
//                                     are the output, or go to a small HBM ring for pv_resample_kernel
//
// One workgroup per row; wave w owns slices w, w + W, ...: it synthesises its frame into its LDS region (or takes
// the windowed frame from the HBM frame ring), waits for its turn, adds the frame, finalises and passes the turn on.  Adds happen strictly in slice order, so every accumulator sample sees the
// reference's sequence of float additions; a dropped slice (adv == 0) simply leaves its frame piled where it is.
// There is no workgroup barrier inside the loop: waves drift apart and hide each other's latencies, the turn
// counter only serialises the few hundred cycles of the add.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_ror1(float v) { // lane i <- lane i - 1, lane 0 <- lane 63
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13C, 0xf, 0xf, false));
}

// ChainArgs::diag is ALWAYS ZERO in the product library: nothing reads AUDIOMOD_PV_CHAIN_DIAG there (pv_engine.cc sets
// the field only under -DPV_DIAG, for tools/chain_diag.sh's elimination runs).  The tests on it stay in the code
// because they are opaque to the compiler: round 3 compiled them out and the kernel ran 20 % SLOWER (0.73 vs 0.60 ms
// per 128 K slices, same source otherwise, profiles/r03/fused_fence_ab.txt) -- without these three conditional
// regions the scheduler moves work across the synthesis / turn / finalise boundaries and pays for it in scalar-register
// reloads.  They are code-motion fences that happen to double as measurement switches in diagnostic builds.
#define PV_CHAIN_DIAG(c, bit) (((c).diag & (bit)) != 0)

struct ChainLds {
    float *acc;   // [AR]
    int *turn;    // next slice whose frame may be added
};
__device__ __forceinline__ ChainLds chain_carve(const ChainArgs &c, char *base) {
    ChainLds l;
    l.acc = reinterpret_cast<float *>(base);
    l.turn = reinterpret_cast<int *>(l.acc + c.AR);
    return l;
}
__host__ __device__ inline size_t chain_shared_bytes(const ChainArgs &c) {
    return sizeof(float) * (size_t)c.AR + 16;
}

// ring images and coefficient table in, before the first slice (whole workgroup)
// (first run of a row: the carried accumulator; later runs start from zeros and rebuild what reaches into their range
// by re-adding the frames before it, ChainSlice flag bit 1)
__device__ __forceinline__ void chain_prologue(const ChainArgs &c, const ChainLds &l, int row, bool first_run = true) {
    const int nt = blockDim.x, tid = threadIdx.x;
    const float4 *sa = reinterpret_cast<const float4 *>(c.st_acc + ((int64_t)(c.acc_sel & 1) * c.rows + row) * c.AR);
    const bool carried = first_run && !(c.acc_sel & 2); // (a stream's first launch starts from the empty accumulator)
    for (int i = tid; i < c.AR / 4; i += nt)
        reinterpret_cast<float4 *>(l.acc)[i] = carried ? sa[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) *l.turn = 0;
    __syncthreads();
}
__device__ __forceinline__ void chain_epilogue(const ChainArgs &c, const ChainLds &l, int row, bool last_run = true) {
    __syncthreads(); // every wave has passed its last turn
    if (!last_run) return; // only the run that ends the launch holds the row's true accumulator
    const int nt = blockDim.x, tid = threadIdx.x;
#if defined(PV_DIAG)
    const int wr_half = (c.acc_sel & 4) ? 0 : ((c.acc_sel & 1) ^ 1); // (AUDIOMOD_PV_DEBUG_STALE_ACC: round 2's single buffer)
#else
    const int wr_half = (c.acc_sel & 1) ^ 1;
#endif
    float4 *sa = reinterpret_cast<float4 *>(c.st_acc + ((int64_t)wr_half * c.rows + row) * c.AR);
    for (int i = tid; i < c.AR / 4; i += nt) sa[i] = reinterpret_cast<const float4 *>(l.acc)[i];
}

// What a wave fetches for its slice before it waits for its turn: the first 256 denominators (the rest, for hops
// above 256 samples, are read in the loop).  The output-table entries are fetched after the turn has been passed
// on: inside the turn they would only hold registers.
struct ChainPrefetch {
    float wd[4];
};
__device__ __forceinline__ void chain_prefetch(const ChainArgs &c, const ChainSlice &sl, int row, int lane,
                                               ChainPrefetch &pf) {
    // (the denominators are laid out by ring quads: entry 0 belongs to sample P_t - r)
    const float *__restrict__ wden = ((row % c.C) > 0 ? c.wden_hi : c.wden) + (sl.acc_pos & 3);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int i = lane + 64 * m;
        pf.wd[m] = wden[sl.wden_off + (i < sl.adv ? i : 0)];
    }
}

__device__ __forceinline__ void chain_wait_turn(int *turn, int tl) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != tl)
        __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void chain_pass_turn(int *turn, int tl, int lane) {
    // LDS only: the wave's outstanding global loads and stores are none of the next wave's business
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(turn, tl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Add NP pieces of a frame (piece j = this lane's four windowed samples 4 (lane + 64 (j0 + j)) .. + 3, zeros past the
// frame's end) into the accumulator ring.  The ring is updated in aligned quads: a frame that starts r = R_ samples
// into a quad is shifted by r through the neighbouring lane (DPP rotate; lane 0 takes lane 63 of the previous
// piece, carried in prevR between calls).  Positions outside the frame receive +0.0f, which leaves any sum that
// started from +0.0f as it is.  `last`: also the quad that holds the frame's final r samples.
template <int R_, int NP>
__device__ __forceinline__ void chain_add_pieces(float *acc, const int AQ, const int a, const int NQ, const int lane,
                                                 const int j0, const float4 (&A)[NP], float4 &prevR, const bool last) {
    float4 *acc4 = reinterpret_cast<float4 *>(acc);
    const int qend = NQ + (R_ ? 1 : 0); // quads the frame touches
    // The quads of a frame are distinct, so every read may be issued before the first write: the add is inside the
    // turn, where a chain of read -> add -> write round trips per quad is time every other wave of the row waits.
    constexpr int NS = NP + (R_ ? 1 : 0);
    float4 V[NS];
    auto quad_of = [&](int j) -> int { // ring quad of this lane's piece j, or -1 when it lies past the frame
        const int u = lane + 64 * (j0 + j);
        int qq = a + u;
        if (qq >= AQ) qq -= AQ;
        return (u < qend && (j < NP || last)) ? qq : -1;
    };
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int qq = quad_of(j);
        V[j] = acc4[qq >= 0 ? qq : 0];
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float4 Aj = j < NP ? A[j < NP ? j : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 S = Aj;
        if (R_ != 0) {
            const float4 Rr = make_float4(dpp_ror1(Aj.x), dpp_ror1(Aj.y), dpp_ror1(Aj.z), dpp_ror1(Aj.w));
            const float4 P = lane == 0 ? prevR : Rr;
            if (j < NP) prevR = Rr;
            if (R_ == 1) S = make_float4(P.w, Aj.x, Aj.y, Aj.z);
            else if (R_ == 2) S = make_float4(P.z, P.w, Aj.x, Aj.y);
            else S = make_float4(P.y, P.z, P.w, Aj.x);
        }
        V[j].x += S.x, V[j].y += S.y, V[j].z += S.z, V[j].w += S.w;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int qq = quad_of(j);
        if (qq >= 0) acc4[qq] = V[j];
    }
}
template <int NP>
__device__ __forceinline__ void chain_add_dispatch(float *acc, int AR, int acc_pos, int NQ, int lane, int j0,
                                                   const float4 (&A)[NP], float4 &prevR, bool last) {
    const int r = acc_pos & 3, AQ = AR >> 2, a = acc_pos >> 2; // r is wave-uniform
    if (r == 0) chain_add_pieces<0, NP>(acc, AQ, a, NQ, lane, j0, A, prevR, last);
    else if (r == 1) chain_add_pieces<1, NP>(acc, AQ, a, NQ, lane, j0, A, prevR, last);
    else if (r == 2) chain_add_pieces<2, NP>(acc, AQ, a, NQ, lane, j0, A, prevR, last);
    else chain_add_pieces<3, NP>(acc, AQ, a, NQ, lane, j0, A, prevR, last);
}

template <int kRes> // 0 = the finalised samples are the output, 1 = they go to the stream ring of the resampler
__device__ __forceinline__ void chain_finish_slice(const ChainArgs &c_in, const ChainLds &l, const ChainSlice &sl,
                                                   const ChainPrefetch &pf, const int row, const int tl, const int lane) {
    ChainArgs c = c_in;
    c.resample = kRes != 0;
    // ---- finalise [P_t, P_t + adv): acc / window sum -> stream ring (or straight out when nothing resamples)
    const float *__restrict__ wden = ((row % c.C) > 0 ? c.wden_hi : c.wden) + (sl.acc_pos & 3);
    float *__restrict__ out = c.out + (int64_t)row * c.out_stride_row + sl.k_off;
    float *__restrict__ stream = c.stream + (int64_t)row * ((int64_t)c.smask + 1);
    auto acc_index = [&](int i) {
        int ai = sl.acc_pos + i;
        return ai >= c.AR ? ai - c.AR : ai;
    };
    const bool quiet = (sl.flags & 2) != 0; // a warm-up slice of a later run: its samples belong to the run before
    auto emit = [&](int i, float y) {
        if (quiet) return;
        if (c.resample) {
            stream[(uint32_t)(sl.str_pos + i) & (uint32_t)c.smask] = y;
        } else if (i < sl.kcnt) {
            out[i] = y;
        }
    };
    if (!PV_CHAIN_DIAG(c, 2)) {
        // (all reads, then the divisions, then the writes: this is still inside the turn)
        float v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int i = lane + 64 * m;
            v[m] = i < sl.adv ? l.acc[acc_index(i)] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int i = lane + 64 * m;
            if (i < sl.adv) {
                l.acc[acc_index(i)] = 0.f;
                emit(i, v[m] / pf.wd[m]);
            }
        }
        for (int i = 256 + lane; i < sl.adv; i += 64) {
            const int ai = acc_index(i);
            const float y = l.acc[ai] / wden[sl.wden_off + i];
            l.acc[ai] = 0.f;
            emit(i, y);
        }
    }
    // ---- the next frame may be added now
    chain_pass_turn(l.turn, tl, lane);
}

// The wave-FFT kernel's version of "wait, add, finalise, pass the turn".  Everything that does not need the
// accumulator happens before the wait (the frame shifted to the ring's quad alignment, the denominators fetched),
// and inside the turn a lane reads its quads, adds, and writes them back once -- zeros where a sample belongs to
// [P_t, P_t + adv), which this frame completes (writeSlice's divide + shift, :1157-1194): those sums stay in the
// lane's registers and are normalised and appended to the stream after the turn has been passed on.  The second
// counter publishes, in slice order, that a slice's stream samples are written (the next slice's wave resamples the
// outputs this one deferred).  wden is laid out by ring quads: entry 0 belongs to sample P_t - r.
// (kFast: the denominators arrive as reciprocals -- the host inverts them, pv_engine.cc -- and normalising is a multiply)
template <int R_, int NP, int kRes, bool kFast = false>
__device__ __forceinline__ void chain_slice_tail(const ChainArgs &c_in, const ChainLds &l, const ChainSlice &sl,
                                                 const float4 (&A)[NP], const bool skip, const int row, const int tl,
                                                 const int lane) {
    ChainArgs c = c_in;
    c.resample = kRes != 0;
    constexpr int NS = NP + (R_ ? 1 : 0);
    const int AQ = c.AR >> 2, a = sl.acc_pos >> 2;
    float4 *acc4 = reinterpret_cast<float4 *>(l.acc);
    // ---- before the turn
    float4 S[NS];
    {
        float4 prevR = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const float4 Aj = j < NP ? A[j < NP ? j : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
            S[j] = Aj;
            if (R_ != 0) {
                const float4 Rr = make_float4(dpp_ror1(Aj.x), dpp_ror1(Aj.y), dpp_ror1(Aj.z), dpp_ror1(Aj.w));
                const float4 P = lane == 0 ? prevR : Rr;
                prevR = Rr;
                if (R_ == 1) S[j] = make_float4(P.w, Aj.x, Aj.y, Aj.z);
                else if (R_ == 2) S[j] = make_float4(P.z, P.w, Aj.x, Aj.y);
                else S[j] = make_float4(P.y, P.z, P.w, Aj.x);
            }
        }
    }
    const float *__restrict__ wden = ((row % c.C) > 0 ? c.wden_hi : c.wden) + sl.wden_off;
    const int fin_quads = (sl.adv + R_ + 3) >> 2; // quads that hold samples of the region being finalised
    float4 wd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = lane + 64 * j;
        wd[j] = *reinterpret_cast<const float4 *>(wden + 4 * (u < fin_quads ? u : 0));
    }
    int q[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int qq = a + lane + 64 * j;
        q[j] = qq >= AQ ? qq - AQ : qq;
    }
    const bool tail_lane = lane == 0; // the quad behind the last full piece belongs to one lane
    // What goes back to the ring: the sum, or zero where the sample is inside [0, adv) relative to P_t --
    // as lane masks in scalar registers (no vector registers held across the wait), made before the turn for the
    // two pieces that reach into the region for hops up to 512 - R samples; later pieces test in place.
    constexpr int NM = NS < 2 ? NS : 2;
    unsigned long long zero_at[NM][4];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int s0 = 4 * (lane + 64 * j) - R_;
        zero_at[j][0] = __builtin_amdgcn_ballot_w64(s0 >= 0 && s0 < sl.adv);
        zero_at[j][1] = __builtin_amdgcn_ballot_w64(s0 + 1 >= 0 && s0 + 1 < sl.adv);
        zero_at[j][2] = __builtin_amdgcn_ballot_w64(s0 + 2 >= 0 && s0 + 2 < sl.adv);
        zero_at[j][3] = __builtin_amdgcn_ballot_w64(s0 + 3 < sl.adv);
    }
    auto masked = [](float v, unsigned long long m) -> float { // lanes set in m get +0
        float o;
        asm("v_cndmask_b32_e64 %0, %1, 0, %2 ; pvmask" : "=v"(o) : "v"(v), "s"(m));
        return o;
    };
    auto write_back = [&](int j, const float4 v) -> float4 {
        const int s0 = 4 * (lane + 64 * j) - R_;
        float4 wb = v;
        if (j < NM) {
            wb.x = masked(v.x, zero_at[j < NM ? j : 0][0]);
            wb.y = masked(v.y, zero_at[j < NM ? j : 0][1]);
            wb.z = masked(v.z, zero_at[j < NM ? j : 0][2]);
            wb.w = masked(v.w, zero_at[j < NM ? j : 0][3]);
        } else if (256 * j - R_ < sl.adv) { // wave-uniform: this piece reaches into the region (hops above ~500)
            wb.x = (s0 >= 0 && s0 < sl.adv) ? 0.f : v.x;
            wb.y = (s0 + 1 >= 0 && s0 + 1 < sl.adv) ? 0.f : v.y;
            wb.z = (s0 + 2 >= 0 && s0 + 2 < sl.adv) ? 0.f : v.z;
            wb.w = (s0 + 3 < sl.adv) ? 0.f : v.w;
        }
        return wb;
    };
    // The masks are pinned in their scalar registers HERE, in front of the wait.  Not for speed (measured: pinning the
    // shifted frame as well, so that the compiler cannot sink its selects into the turn, costs 0.320 -> 0.349 ms per
    // launch in spills; addresses and masks only: 0.320; raised priority alone: 0.315) but for correctness: on gfx950 a
    // VALU instruction that reads an SGPR written by a VALU instruction (the v_cmp behind a ballot) needs two wait states
    // in between, the compiler's hazard recogniser inserts them for its own instructions only, and masked() is inline
    // assembly -- left free, the compiler may put the compare right in front of it (seen as a rare wrong write-back:
    // test_fused_overlap_add_is_bit_identical_to_the_tile_path failed once in four full runs).  With the compares
    // forced to this side of the wait loop, the loop's own instructions are the distance.
    // The s_nop inside the pin makes the distance part of the instruction stream rather than a property of what the
    // compiler happens to schedule behind it: every compare that produced a mask is in front of this statement (the
    // masks are its operands), every masked() behind it.
#pragma unroll
    for (int j = 0; j < NM; ++j)
        asm volatile("s_nop 1 ; pvpin" : "+s"(zero_at[j][0]), "+s"(zero_at[j][1]), "+s"(zero_at[j][2]), "+s"(zero_at[j][3]));
    // ---- the turn
    if (!PV_CHAIN_DIAG(c, 4)) chain_wait_turn(l.turn, tl);
    __builtin_amdgcn_s_setprio(3); // the turn's instructions ahead of the three other waves of this SIMD
    float4 V[NS];
    if (!skip) {
#pragma unroll
        for (int j = 0; j < NS; ++j) V[j] = acc4[q[j]];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            V[j].x += S[j].x, V[j].y += S[j].y, V[j].z += S[j].z, V[j].w += S[j].w;
            if (j < NP || tail_lane) acc4[q[j]] = write_back(j, V[j]);
        }
    } else {
        // a frame this channel does not add (CONSTANT-mode overrun): adv is 0 then, nothing is finalised either and
        // V is never looked at (left undefined rather than zero-filled: 36 moves the common path would execute too)
#pragma unroll
        for (int j = 0; j < NS; ++j) asm volatile("" : "=v"(V[j].x), "=v"(V[j].y), "=v"(V[j].z), "=v"(V[j].w));
    }
    chain_pass_turn(l.turn, tl, lane);
    __builtin_amdgcn_s_setprio(0);
    // ---- after the turn: normalise the finalised samples and append them to the stream (or emit them)
    float *__restrict__ out = c.out + (int64_t)row * c.out_stride_row + sl.k_off;
    float *__restrict__ stream = c.stream + (int64_t)row * ((int64_t)c.smask + 1);
    const bool quiet = (sl.flags & 2) != 0; // a warm-up slice of a later run: its samples belong to the run before
    auto emit = [&](int sidx, float y) { // sidx: sample index relative to P_t
        if (quiet || sidx < 0 || sidx >= sl.adv) return;
        if (c.resample) {
            stream[(uint32_t)(sl.str_pos + sidx) & (uint32_t)c.smask] = y;
        } else if (sidx < sl.kcnt) {
            out[sidx] = y;
        }
    };
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        if (256 * j - R_ < sl.adv) { // wave-uniform
            const int u = lane + 64 * j;
            const float4 d = j < 2 ? wd[j < 2 ? j : 0]
                                   : *reinterpret_cast<const float4 *>(wden + 4 * (u < fin_quads ? u : 0));
            const int s0 = 4 * u - R_;
            if (j < NP || tail_lane) {
                emit(s0, kFast ? V[j].x * d.x : V[j].x / d.x);
                emit(s0 + 1, kFast ? V[j].y * d.y : V[j].y / d.y);
                emit(s0 + 2, kFast ? V[j].z * d.z : V[j].z / d.z);
                emit(s0 + 3, kFast ? V[j].w * d.w : V[j].w / d.w);
            }
        }
    }
}

// (the all-modes variant -- frequency compression, vocoder, ... -- needs ~150 VGPRs where the plain ones fit 128:
// compiled for twelve waves per workgroup, three per SIMD, instead of spilling)
// Threads per workgroup each variant is compiled for: sixteen waves (128 registers) for the specialisations, twelve for
// the all-modes kernel and the exact formant / gender one (~150 registers); the free-form formant / gender kernel fits
// 128 (round 3).  4096-point frames: eight waves (LDS).  Core::init sizes the launch by the same rule.
constexpr int chain_kernel_max_threads(int NC, int kPlainCore, bool kFast) {
    return NC <= 1024 ? ((kPlainCore < 0 || (kPlainCore == 3 && !kFast)) ? 768 : 1024) : 512;
}
template <int NC, int kPlainCore, int kRes, bool kFast = false>
__global__ __launch_bounds__(chain_kernel_max_threads(NC, kPlainCore, kFast)) void pv_synth_chain_kernel(
    const SynthArgs s, const ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    using W = WF<NC>;
    constexpr int N = 2 * NC, hs = NC, NQ = N / 4, NP = NQ / 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = blockIdx.x, run = blockIdx.y;
    // a row's slices of the launch are split into gridDim.y runs, one workgroup each (entry i of a run's list is its
    // i-th slice in order, warm-up slices first; ChainSlice::tl says which slice of the launch it is)
    const int i_begin = c.run_off[run], i_count = c.run_off[run + 1] - i_begin;
    cf *wlds = reinterpret_cast<cf *>(smem_raw) + wave * W::LDS_CF;
    const ChainLds l = chain_carve(c, smem_raw + (size_t)c.waves * W::LDS_CF * sizeof(cf));
    chain_prologue(c, l, row, run == 0);
    const float *__restrict__ w = s.tb.window;
    const bool upper = (row % c.C) > 0;
    const int lane0 = lane;
    for (int i = wave; i < i_count; i += c.waves) {
        // the lane id, made opaque once per iteration: everything derived from it is recomputed per slice instead of
        // being hoisted out of the loop and held in registers (which spilled ~100 VGPRs)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const ChainSlice sl = c.slices[i_begin + i];
        const int tl = sl.tl;
        const bool skip = (sl.flags & 1) && upper; // wave-uniform
        float4 A[NP];
        if (!skip) {
            if (!PV_CHAIN_DIAG(c, 8)) synth_wave_role<NC, kPlainCore, 1, kFast>(s, row, tl, wlds, lane);
            // ifftshift + synthesis window (phasevocoderimpl.h:183-198): four consecutive samples per lane and piece
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int i = 4 * (lane + 64 * j);
                const int e = ((i + hs) & (N - 1)) >> 1;
                const cf z0 = wlds[W::pad(e)], z1 = wlds[W::pad(e) + 1];
                const float4 ww = *reinterpret_cast<const float4 *>(w + i);
                A[j] = make_float4(z0.x * ww.x, z0.y * ww.y, z1.x * ww.z, z1.y * ww.w);
            }
        }
        const int r = sl.acc_pos & 3; // wave-uniform
        if (r == 0) chain_slice_tail<0, NP, kRes, kFast>(c, l, sl, A, skip, row, i, lane);
        else if (r == 1) chain_slice_tail<1, NP, kRes, kFast>(c, l, sl, A, skip, row, i, lane);
        else if (r == 2) chain_slice_tail<2, NP, kRes, kFast>(c, l, sl, A, skip, row, i, lane);
        else chain_slice_tail<3, NP, kRes, kFast>(c, l, sl, A, skip, row, i, lane);
    }
    chain_epilogue(c, l, row, run == (int)gridDim.y - 1);
}

// Any FFT size: the synthesis kernel has written the windowed frames to the HBM frame ring; the chain takes them
// from there, up to eight pieces (2048 samples) at a time.
template <int kRes> __global__ __launch_bounds__(1024) void pv_frames_chain_kernel(const ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = blockIdx.x, run = blockIdx.y;
    const int i_begin = c.run_off[run], i_count = c.run_off[run + 1] - i_begin;
    const ChainLds l = chain_carve(c, smem_raw);
    chain_prologue(c, l, row, run == 0);
    const int NQ = c.N >> 2;
    const int groups = (NQ + 511) >> 9;
    const bool upper = (row % c.C) > 0;
    for (int i = wave; i < i_count; i += c.waves) {
        const ChainSlice sl = c.slices[i_begin + i];
        const int tl = sl.tl;
        const bool skip = (sl.flags & 1) && upper;
        ChainPrefetch pf;
        chain_prefetch(c, sl, row, lane, pf);
        const int fslot = (int)((c.t0 + tl) & (int64_t)(c.FR - 1));
        const float4 *__restrict__ fr =
            reinterpret_cast<const float4 *>(c.frames + ((int64_t)row * c.FR + fslot) * c.N);
        auto load_group = [&](int g, float4 (&A)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u = lane + 64 * (8 * g + j);
                A[j] = (u < NQ && !skip) ? fr[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        float4 A[8];
        load_group(0, A);
        chain_wait_turn(l.turn, i);
        if (!skip) {
            float4 prevR = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g = 0; g < groups; ++g) {
                if (g > 0) load_group(g, A);
                chain_add_dispatch<8>(l.acc, c.AR, sl.acc_pos, NQ, lane, 8 * g, A, prevR, g == groups - 1);
            }
            wave_sync();
        }
        chain_finish_slice<kRes>(c, l, sl, pf, row, i, lane);
    }
    chain_epilogue(c, l, row, run == (int)gridDim.y - 1);
}

size_t chain_lds_bytes(const ChainArgs &a, int nc_wave) {
    const size_t per_wave = nc_wave == 256    ? WF<256>::LDS_CF * sizeof(cf)
                            : nc_wave == 512  ? WF<512>::LDS_CF * sizeof(cf)
                            : nc_wave == 1024 ? WF<1024>::LDS_CF * sizeof(cf)
                            : nc_wave == 2048 ? WF<2048>::LDS_CF * sizeof(cf) : 0;
    return (size_t)a.waves * per_wave + chain_shared_bytes(a);
}

template <int NC, int kPlainCore> static void launch_synth_chain_res(const SynthArgs &s, const ChainArgs &c, hipStream_t st) {
    const size_t lds = chain_lds_bytes(c, NC);
    const bool use_fast = kPlainCore >= 0 && c.fast;
    const int kMaxThreads = use_fast ? chain_kernel_max_threads(NC, kPlainCore, true) : chain_kernel_max_threads(NC, kPlainCore, false);
    if (64 * c.waves > kMaxThreads) { // (never: Core::init sizes chain_waves by the same rule; a launch beyond the bounds faults)
        fprintf(stderr, "audiomod_pv: fused kernel launch of %d waves exceeds its bounds (%d threads): not launched\n", c.waves, kMaxThreads);
        return;
    }
    const dim3 grid(c.rows, c.runs), block(64 * c.waves);
    static unsigned long long m0 = 0, m1 = 0, f0 = 0, f1 = 0;
    if constexpr (kPlainCore >= 0) { // the specialisations have a free-form (PV_ARITH_FAST) twin
        if (c.fast) {
            if (!c.resample) {
                allow_big_lds_dev(pv_synth_chain_kernel<NC, kPlainCore, 0, true>, f0);
                hipLaunchKernelGGL((pv_synth_chain_kernel<NC, kPlainCore, 0, true>), grid, block, lds, st, s, c);
            } else {
                allow_big_lds_dev(pv_synth_chain_kernel<NC, kPlainCore, 1, true>, f1);
                hipLaunchKernelGGL((pv_synth_chain_kernel<NC, kPlainCore, 1, true>), grid, block, lds, st, s, c);
            }
            return;
        }
    }
    if (!c.resample) {
        allow_big_lds_dev(pv_synth_chain_kernel<NC, kPlainCore, 0>, m0);
        hipLaunchKernelGGL((pv_synth_chain_kernel<NC, kPlainCore, 0>), grid, block, lds, st, s, c);
    } else {
        allow_big_lds_dev(pv_synth_chain_kernel<NC, kPlainCore, 1>, m1);
        hipLaunchKernelGGL((pv_synth_chain_kernel<NC, kPlainCore, 1>), grid, block, lds, st, s, c);
    }
}
// does launch_synth_chain pick a free-form kernel for this configuration?  (the engine then uploads the window-sum
// denominators as reciprocals)
bool synth_chain_has_fast(const SynthArgs &s) {
    const bool plain = !s.do_freq_comp && s.voc_band_len < 0 && !s.robotic && !s.passthru && !s.whisper &&
                       !synth_generic_only() && s.coremode >= 0 && s.coremode <= 2;
    const bool fc_locked = s.do_freq_comp && s.voc_band_len < 0 && !s.robotic && !s.passthru && !s.whisper &&
                           !synth_generic_only() && s.coremode == 1;
    return (s.tb.nc == 1024 && (plain || fc_locked)) || (s.tb.nc == 2048 && plain) ||
           ((s.tb.nc == 512 || s.tb.nc == 256) && plain);
}

void launch_synth_chain(const SynthArgs &s, const ChainArgs &c, hipStream_t st) {
    const bool plain = !s.do_freq_comp && s.voc_band_len < 0 && !s.robotic && !s.passthru && !s.whisper &&
                       !synth_generic_only() && s.coremode >= 0 && s.coremode <= 2;
    const bool fc_locked = s.do_freq_comp && s.voc_band_len < 0 && !s.robotic && !s.passthru && !s.whisper &&
                           !synth_generic_only() && s.coremode == 1;
    if (s.tb.nc == 256) { // fft 512
        if (plain && s.coremode == 1) launch_synth_chain_res<256, 1>(s, c, st);
        else if (plain && s.coremode == 0) launch_synth_chain_res<256, 0>(s, c, st);
        else if (plain) launch_synth_chain_res<256, 2>(s, c, st);
        else launch_synth_chain_res<256, -1>(s, c, st);
    } else if (s.tb.nc == 512) { // fft 1024: the plain specialisations (sixteen waves, as the engine sizes the launch), else all modes
        if (plain && s.coremode == 1) launch_synth_chain_res<512, 1>(s, c, st);
        else if (plain && s.coremode == 0) launch_synth_chain_res<512, 0>(s, c, st);
        else if (plain) launch_synth_chain_res<512, 2>(s, c, st);
        else launch_synth_chain_res<512, -1>(s, c, st);
    } else if (s.tb.nc == 1024) {
        if (fc_locked) launch_synth_chain_res<1024, 3>(s, c, st);
        else if (plain && s.coremode == 1) launch_synth_chain_res<1024, 1>(s, c, st);
        else if (plain && s.coremode == 0) launch_synth_chain_res<1024, 0>(s, c, st);
        else if (plain) launch_synth_chain_res<1024, 2>(s, c, st);
        else launch_synth_chain_res<1024, -1>(s, c, st);
    } else {
        if (plain && s.coremode == 1) launch_synth_chain_res<2048, 1>(s, c, st);
        else if (plain && s.coremode == 0) launch_synth_chain_res<2048, 0>(s, c, st);
        else if (plain) launch_synth_chain_res<2048, 2>(s, c, st);
        else launch_synth_chain_res<2048, -1>(s, c, st);
    }
}

void launch_frames_chain(const ChainArgs &c, hipStream_t st) {
    const size_t lds = chain_lds_bytes(c, 0);
    const dim3 grid(c.rows, c.runs), block(64 * c.waves);
    static unsigned long long m0 = 0, m1 = 0;
    if (!c.resample) {
        allow_big_lds_dev(pv_frames_chain_kernel<0>, m0);
        hipLaunchKernelGGL(pv_frames_chain_kernel<0>, grid, block, lds, st, c);
    } else {
        allow_big_lds_dev(pv_frames_chain_kernel<1>, m1);
        hipLaunchKernelGGL(pv_frames_chain_kernel<1>, grid, block, lds, st, c);
    }
}

// --------------------------------------------------------------------------------------------
// Speex resampling of the normalised overlap-add stream (the fused path's second kernel): a tile of 256 outputs of
// two rows; the samples its windows cover are one contiguous run of each row's stream ring, copied to LDS, then every
// thread runs resampler_basic_interpolate_single / resampler_basic_direct_single (speex/resample.c:462-560, 353-401)
// for the same output of both rows, so that a tap's coefficients are read once for two outputs (all rows share one
// schedule).  Separate multiply and add (-ffp-contract=off), tap order and accumulators as in the reference.
// --------------------------------------------------------------------------------------------
constexpr int kResRows = 4; // rows per workgroup of the resampling kernel: a tap's coefficients are read once for all

template <int kRes> // 1 = direct sinc table, 2 = cubic-interpolated table
__global__ __launch_bounds__(kTileOut) void pv_resample_kernel(const ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    constexpr int NR = kResRows;
    float4 *tab4 = reinterpret_cast<float4 *>(smem_raw);
    float *stab = reinterpret_cast<float *>(smem_raw);
    float *xs = reinterpret_cast<float *>(smem_raw + a.tab_bytes); // [NR][lds_floats]
    const int nt = blockDim.x, tid = threadIdx.x;
    const ResTile tile = a.tiles[blockIdx.x];
    const int row0 = blockIdx.y * NR, NF = a.filt_len;
    const int nr = a.rows - row0 < NR ? a.rows - row0 : NR;
    uint2 oe = make_uint2(0u, 0u);
    if (tid < tile.kcnt) oe = a.otab[(int64_t)blockIdx.x * kTileOut + tid];
    // the tile's stream samples of every row, all loads in flight before the first LDS write (a tile needs at most
    // two samples per thread and row at the ratios that resample: n_cnt <= 256 * num/den + filt_len)
    float xv[NR][2];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float *__restrict__ st = a.stream + (int64_t)(r < nr ? row0 + r : row0) * ((int64_t)a.smask + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * kTileOut;
            const int64_t n = tile.n_lo + i; // the stream is zero before its first sample (skip_zeros, :1225)
            xv[r][h] = (i < tile.n_cnt && n >= 0) ? st[(uint32_t)n & (uint32_t)a.smask] : 0.f;
        }
    }
    if (kRes == 2) {
        const int cnt = a.oversample * (NF + 1);
        for (int i = tid; i < cnt; i += nt) tab4[i] = a.tab4[i];
    } else {
        for (int i = tid; i < a.sinc_len; i += nt) stab[i] = a.sinc[i];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * kTileOut;
            if (i < tile.n_cnt) xs[r * a.lds_floats + i] = xv[r][h];
        }
    }
    // (longer tiles -- strong down-sampling, more than two samples per thread -- take the rest in a loop)
    for (int i = tid + 2 * kTileOut; i < tile.n_cnt; i += nt)
        for (int r = 0; r < nr; ++r) {
            const int64_t n = tile.n_lo + i;
            const float *__restrict__ st = a.stream + (int64_t)(row0 + r) * ((int64_t)a.smask + 1);
            xs[r * a.lds_floats + i] = n >= 0 ? st[(uint32_t)n & (uint32_t)a.smask] : 0.f;
        }
    __syncthreads();
    if (tid >= tile.kcnt) return;
    float *__restrict__ out = a.out + (int64_t)row0 * a.out_stride_row + (tile.k0 - a.k_base) + tid;
    const float *x = xs + (int)(oe.x & 0xffffu); // tap j = 0
    if (kRes == 2) {
        const float frac = __uint_as_float(oe.y);
        const float4 *__restrict__ T = tab4 + (int)(oe.x >> 16) * (NF + 1);
        v2f a01[NR], a23[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) a01[r] = v2f{0.f, 0.f}, a23[r] = v2f{0.f, 0.f};
#pragma unroll 4
        for (int j = 0; j < NF; ++j) { // NF is a multiple of 4 (resample.c:687)
            const float4 c = T[j];
            const v2f c01 = {c.x, c.y}, c23 = {c.z, c.w};
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float xr = x[r * a.lds_floats + j]; // a missing row holds zeros: computed, never stored
                const v2f xx = {xr, xr};
                a01[r] += xx * c01;
                a23[r] += xx * c23;
            }
        }
        // cubic_coef (resample.c:339-351)
        const float c0 = -0.16667f * frac + 0.16667f * frac * frac * frac;
        const float c1 = frac + 0.5f * frac * frac - 0.5f * frac * frac * frac;
        const float c3 = -0.33333f * frac + 0.5f * frac * frac - 0.16667f * frac * frac * frac;
        const float c2 = (float)(1. - c0 - c1 - c3);
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (r < nr)
                out[(int64_t)r * a.out_stride_row] = (c0 * a01[r].x) + (c1 * a01[r].y) + (c2 * a23[r].x) + (c3 * a23[r].y);
    } else {
        const float *t = stab + (oe.x >> 16) * (uint32_t)NF;
        for (int r = 0; r < nr; ++r) {
            float sum = 0.f;
            for (int j = 0; j < NF; ++j) sum += x[r * a.lds_floats + j] * t[j];
            out[(int64_t)r * a.out_stride_row] = sum;
        }
    }
}

// --------------------------------------------------------------------------------------------
// The same resampling with the arithmetic the 1e-4 RMS contract leaves free (ResArgs::fast; the batch engine's
// default, audiomod_pv.h pv_set_arithmetic).  The output of the reference's interpolating resampler is
//     sum_j x[j] * (c0 T0[j] + c1 T1[j] + c2 T2[j] + c3 T3[j])          (resample.c:494-543, regrouped)
// and the bracket -- the cubic-interpolated filter tap -- depends on the output index only, not on the row: every
// row of a batch follows one schedule.  So a thread interpolates each tap ONCE (one multiply + three fma) and applies
// it to the same output of NR rows with one fma per row: (4 + NR) / NR fused operations per row and tap where the
// reference's operation order needs four multiplies and four adds (~4.5x fewer vector instructions at NR = 8).  The
// result differs from the reference's by rounding only (~1e-7 relative: one accumulator instead of four, fma).
// --------------------------------------------------------------------------------------------
constexpr int kResFastRows = 16; // (8 where the batch has fewer rows or the tile does not fit 64 KB of LDS)

template <int kRes, int NR> // 1 = direct sinc table, 2 = cubic-interpolated table
__global__ __launch_bounds__(kTileOut) void pv_resample_fast_kernel(const ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    float4 *tab4 = reinterpret_cast<float4 *>(smem_raw);
    float *stab = reinterpret_cast<float *>(smem_raw);
    float *xs = reinterpret_cast<float *>(smem_raw + a.tab_bytes); // [NR][lds_floats]
    const int nt = blockDim.x, tid = threadIdx.x;
    const ResTile tile = a.tiles[blockIdx.x];
    const int row0 = blockIdx.y * NR, NF = a.filt_len;
    const int nr = a.rows - row0 < NR ? a.rows - row0 : NR;
    uint2 oe = make_uint2(0u, 0u);
    if (tid < tile.kcnt) oe = a.otab[(int64_t)blockIdx.x * kTileOut + tid];
    // the tile's stream samples of every row (a missing row re-reads the group's first: computed, never stored)
    for (int i = tid; i < tile.n_cnt; i += nt) {
        const int64_t n = tile.n_lo + i; // the stream is zero before its first sample (skip_zeros, :1225)
        float xv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float *__restrict__ st = a.stream + (int64_t)(r < nr ? row0 + r : row0) * ((int64_t)a.smask + 1);
            xv[r] = n >= 0 ? st[(uint32_t)n & (uint32_t)a.smask] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) xs[r * a.lds_floats + i] = xv[r];
    }
    if (kRes == 2) {
        const int cnt = a.oversample * (NF + 1);
        for (int i = tid; i < cnt; i += nt) tab4[i] = a.tab4[i];
    } else {
        for (int i = tid; i < a.sinc_len; i += nt) stab[i] = a.sinc[i];
    }
    __syncthreads();
    if (tid >= tile.kcnt) return;
    float *__restrict__ out = a.out + (int64_t)row0 * a.out_stride_row + (tile.k0 - a.k_base) + tid;
    const float *x = xs + (int)(oe.x & 0xffffu); // tap j = 0
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.f;
    if (kRes == 2) {
        const float frac = __uint_as_float(oe.y);
        // cubic_coef (resample.c:339-351)
        const float c0 = -0.16667f * frac + 0.16667f * frac * frac * frac;
        const float c1 = frac + 0.5f * frac * frac - 0.5f * frac * frac * frac;
        const float c3 = -0.33333f * frac + 0.5f * frac * frac - 0.16667f * frac * frac * frac;
        const float c2 = (float)(1. - c0 - c1 - c3);
        const float4 *__restrict__ T = tab4 + (int)(oe.x >> 16) * (NF + 1);
#pragma unroll 4
        for (int j = 0; j < NF; ++j) { // NF is a multiple of 4 (resample.c:687)
            // (PV_EXP_RES: elimination builds for timing only -- bit 0 no sample reads, bit 1 no coefficient reads)
#if defined(PV_EXP_RES) && (PV_EXP_RES & 2)
            const float4 c = make_float4(frac, c0, c1, (float)j);
#else
            const float4 c = T[j];
#endif
            const float h = __builtin_fmaf(c3, c.w, __builtin_fmaf(c2, c.z, __builtin_fmaf(c1, c.y, c0 * c.x)));
#pragma unroll
            for (int r = 0; r < NR; ++r) {
#if defined(PV_EXP_RES) && (PV_EXP_RES & 1)
                acc[r] = __builtin_fmaf(frac + (float)r, h, acc[r]);
#else
                acc[r] = __builtin_fmaf(x[r * a.lds_floats + j], h, acc[r]);
#endif
            }
        }
    } else {
        const float *t = stab + (oe.x >> 16) * (uint32_t)NF;
#pragma unroll 4
        for (int j = 0; j < NF; ++j) {
            const float h = t[j];
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = __builtin_fmaf(x[r * a.lds_floats + j], h, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
        if (r < nr) out[(int64_t)r * a.out_stride_row] = acc[r];
}

// (Measured and left out, round 3: NO = 4 consecutive outputs per thread walking the samples of their common span once
// -- the filter rows padded with zeros so that taps outside 0 .. NF - 1 contribute nothing -- four times fewer sample
// reads, bit-identical output.  Alone it is faster, 0.255 -> 0.202 ms per launch, its four 16-byte coefficient reads per
// sample step then being the LDS cost; beside the rotation chain, where the kernel really runs, its 64 KB of LDS per
// workgroup and low occupancy make it slower, 0.30 -> 0.34-0.38 ms: profiles/r03/resample_elim.txt.)
template <int NR> static void launch_resample_fast_rows(const ResArgs &a, hipStream_t st) {
    const size_t lds = (size_t)a.tab_bytes + sizeof(float) * (size_t)a.lds_floats * NR;
    const dim3 grid(a.ntiles, (a.rows + NR - 1) / NR);
    static unsigned long long m1 = 0, m2 = 0;
    if (a.interp) {
        allow_big_lds_dev(pv_resample_fast_kernel<2, NR>, m2);
        hipLaunchKernelGGL((pv_resample_fast_kernel<2, NR>), grid, dim3(kTileOut), lds, st, a);
    } else {
        allow_big_lds_dev(pv_resample_fast_kernel<1, NR>, m1);
        hipLaunchKernelGGL((pv_resample_fast_kernel<1, NR>), grid, dim3(kTileOut), lds, st, a);
    }
}
static void launch_resample_fast(const ResArgs &a, hipStream_t st) {
    static const int rows_env = [] { // AUDIOMOD_PV_RES_ROWS=8|16 (tuning knob)
        const char *e = getenv("AUDIOMOD_PV_RES_ROWS");
        return e ? atoi(e) : 0;
    }();
    const int want = rows_env ? rows_env : kResFastRows; // (8 -> 16 rows per thread: 0.265 -> 0.255 ms per launch alone, cfg4 -7 st +3 %)
    if (want >= 16 && a.rows >= 16 && (size_t)a.tab_bytes + sizeof(float) * (size_t)a.lds_floats * 16 <= 64 * 1024)
        launch_resample_fast_rows<16>(a, st);
    else
        launch_resample_fast_rows<8>(a, st);
}

void launch_resample(const ResArgs &a, hipStream_t st) {
    if (a.ntiles <= 0) return; // (a group of dropped slices completes no output)
    if (a.fast) return launch_resample_fast(a, st);
    const size_t lds = (size_t)a.tab_bytes + sizeof(float) * (size_t)a.lds_floats * kResRows;
    const dim3 grid(a.ntiles, (a.rows + kResRows - 1) / kResRows);
    static unsigned long long m1 = 0, m2 = 0;
    if (a.interp) {
        allow_big_lds_dev(pv_resample_kernel<2>, m2);
        hipLaunchKernelGGL(pv_resample_kernel<2>, grid, dim3(kTileOut), lds, st, a);
    } else {
        allow_big_lds_dev(pv_resample_kernel<1>, m1);
        hipLaunchKernelGGL(pv_resample_kernel<1>, grid, dim3(kTileOut), lds, st, a);
    }
}

// --------------------------------------------------------------------------------------------
// The drop-in streaming path in ONE launch (opt-in, AUDIOMOD_PV_STREAM_LAUNCHES=single).  A 480-frame call of a
// stereo stream triggers two or three slices per channel: far too little work to fill the chip, and with one
// launch per stage the host spends ~7 us on each of five launches.  One workgroup of eight waves runs the stages
// back to back instead -- the same device functions as the batch kernels, frames / steps spread over the waves,
// rows and tiles looped -- with a workgroup barrier (and a fence: the stages hand data over through global
// memory) between stages.  Measured on MI355X it LOSES (93 vs 70 us per call, 513 vs 242 us at 4800-frame
// calls): the separate launches are asynchronous and already overlap the kernels, while a single workgroup pays
// every stage's latency in sequence on one CU.  Kept as a tested alternative, off by default.
// --------------------------------------------------------------------------------------------
constexpr int kStreamThreads = 512;

__device__ __forceinline__ void stage_handoff() {
    __threadfence();                  // later stages read what other waves of this workgroup wrote to global memory
    __builtin_amdgcn_s_dcache_inv();  // ... some of it through the scalar cache (wave-uniform counts and modes)
    __syncthreads();
}

template <int NC> __global__ __launch_bounds__(kStreamThreads) void pv_stream_kernel(const StreamArgs s) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PV_POISON_LDS(reinterpret_cast<char *>(smem_raw));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = kStreamThreads / 64;
    const int Tn = s.aa.Tn, rows = s.aa.rows, work = rows * Tn;
    cf *wlds = reinterpret_cast<cf *>(smem_raw) + wave * WF<NC>::LDS_CF;
    for (int i = wave; i < work; i += nw)
        analyze_wave_role<NC, (kStreamThreads / 64) * WF<NC>::LDS_CF * sizeof(cf)>(s.aa, i / Tn, i % Tn, wlds);
    stage_handoff();
    if (s.coremode == 1) {
        // (slice-major order is not needed: a step only reads peak lists, all written by now)
        for (int i = wave; i < work; i += nw)
            match_wave_role(s.ma, i / Tn, i % Tn, smem_raw + (size_t)wave * match_wave_lds(s.ma.hs, s.ma.PKP));
        stage_handoff();
        for (int row = 0; row < rows; ++row) {
            seq_role<1>(s.qa, row, smem_raw);
            stage_handoff();
        }
    } else if (s.coremode == 0) {
        for (int row = 0; row < rows; ++row)
            for (int i = threadIdx.x; i < s.pa.hs; i += kStreamThreads) prop_role(s.pa, row, i);
        stage_handoff();
    }
    if (s.cepstral) {
        for (int i = wave; i < work; i += nw) cepstral_wave_role<NC>(s.ca, i / Tn, i % Tn, wlds);
        stage_handoff();
    }
    for (int i = wave; i < work; i += nw) synth_wave_role<NC>(s.sa, i / Tn, i % Tn, wlds);
    stage_handoff();
    for (int tile = 0; tile < s.oa.ntiles; ++tile)
        for (int row = 0; row < rows; ++row) {
            ola_role<1>(s.oa, tile, row, smem_raw);
            __syncthreads();
        }
}

static size_t stream_lds_bytes(const StreamArgs &s) {
    const size_t per_wave = (s.aa.tb.nc == 1024 ? WF<1024>::LDS_CF : WF<2048>::LDS_CF) * sizeof(cf);
    size_t lds = (kStreamThreads / 64) * per_wave + 4 * PV_ATAN_BLOB_WORDS; // + atan2f's interval table
    const size_t m = (kStreamThreads / 64) * match_wave_lds(s.ma.hs, s.ma.PKP);
    if (s.coremode == 1 && m > lds) lds = m;
    if (s.coremode == 1 && seq_lds_bytes(s.qa) > lds) lds = seq_lds_bytes(s.qa);
    if (ola_lds_bytes(s.oa, 1) > lds) lds = ola_lds_bytes(s.oa, 1);
    return lds;
}

bool stream_kernel_supported(const StreamArgs &s) {
    return (s.aa.tb.nc == 1024 || s.aa.tb.nc == 2048) && stream_lds_bytes(s) <= 160 * 1024 - 512;
}

// The wave-per-frame analysis code addresses atan2f's table by a compile-time LDS address, which is right only while
// the kernels that contain it have no static LDS (their dynamic LDS then starts at 0).  The engine asks once per
// device, at creation.
bool lds_starts_at_zero() {
    const void *ks[] = {reinterpret_cast<const void *>(pv_analyze_wave_kernel<256, 1>),
                        reinterpret_cast<const void *>(pv_analyze_wave_kernel<512, 1>),
                        reinterpret_cast<const void *>(pv_analyze_wave_kernel<1024, 1>),
                        reinterpret_cast<const void *>(pv_analyze_wave_kernel<2048, 1>),
                        reinterpret_cast<const void *>(pv_stream_kernel<1024>),
                        reinterpret_cast<const void *>(pv_stream_kernel<2048>)};
    for (const void *k : ks) {
        hipFuncAttributes fa{};
        if (hipFuncGetAttributes(&fa, k) != hipSuccess || fa.sharedSizeBytes != 0) return false;
    }
    return true;
}

void launch_stream(const StreamArgs &s, hipStream_t st) {
    const size_t lds = stream_lds_bytes(s);
    if (s.aa.tb.nc == 1024) {
        static unsigned long long big1 = 0;
        allow_big_lds_dev(pv_stream_kernel<1024>, big1);
        hipLaunchKernelGGL((pv_stream_kernel<1024>), dim3(1), dim3(kStreamThreads), lds, st, s);
    } else {
        static unsigned long long big2 = 0;
        allow_big_lds_dev(pv_stream_kernel<2048>, big2);
        hipLaunchKernelGGL((pv_stream_kernel<2048>), dim3(1), dim3(kStreamThreads), lds, st, s);
    }
}

} // namespace pv
