// pv_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the phase-vocoder engine.
//
// Four kernels, one per HBM-visible stage of SURVEY.md section 8(d):
//   pv_analyze_kernel : window -> fftshift -> real FFT (LDS butterflies) -> (mag, phase)
//                       replaces analyzeSlice + FFT::forwardPolar + kiss_fftr
//                       (reference phasevocoderprocess.cc:492-503, FFT.cc:2617-2631, kiss_fftr.c:67-121)
//   pv_phase_kernel   : per-stream sequential phase propagation (simple / phase-locked / int-ratio)
//                       replaces modifySlice{Simple,PhaseLocked,IntRatio} (phasevocoderprocess.cc:558-753)
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
// OLA and resampler MACs are bit-identical to the x86 reference; only atan2f/sinf/cosf differ
// (device libm, a few ulp).  princarg stays in double with a true IEEE divide, as the reference
// (common/system/sys.h:84,91).
#include "pv_kernels.h"

#include <hip/hip_runtime.h>

namespace pv {

#define PV_PI 3.14159265358979323846

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    float2 m;
    m.x = a.x * b.x - a.y * b.y;
    m.y = a.x * b.y + a.y * b.x;
    return m;
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

__device__ __forceinline__ double princarg(double a) {
    const double x = a + PV_PI;
    const double y = -2.0 * PV_PI;
    return (x - (y * floor(x / y))) + PV_PI;
}

// XCD-aware block -> (row, slice) map: blocks b and b+8 share an XCD (and its L2), so each XCD walks
// whole rows (one stream-channel) slice after slice and re-reads the overlapping input from its own L2.
__device__ __forceinline__ bool block_to_row_slice(int Tn, int rows, int &row, int &tl) {
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    row = xcd + 8 * (q / Tn);
    tl = q % Tn;
    return row < rows;
}

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
// analysis
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFftThreads) void pv_analyze_kernel(const AnalyzeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);
    int row, tl;
    if (!block_to_row_slice(a.Tn, a.rows, row, tl)) return;
    const DevTables &tb = a.tb;
    const int N = tb.N, hs = tb.hs, nc = tb.nc, nt = blockDim.x;
    const int64_t a0 = (a.t0 + tl) * (int64_t)a.hop;
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
    float *__restrict__ mag = a.mag + ((int64_t)row * a.Tc + tl) * tb.HP;
    float *__restrict__ ph = a.phase + ((int64_t)row * a.Tc + tl) * tb.HP;
    for (int k = threadIdx.x; k <= nc / 2; k += nt) {
        if (k == 0) {
            const float2 tdc = buf[0];
            const float r0 = tdc.x + tdc.y, rn = tdc.x - tdc.y;
            mag[0] = sqrtf(r0 * r0 + 0.f * 0.f);
            ph[0] = atan2f(0.f, r0);
            mag[nc] = sqrtf(rn * rn + 0.f * 0.f);
            ph[nc] = atan2f(0.f, rn);
        } else {
            const float2 fpk = buf[k];
            const float2 q = buf[nc - k];
            const float2 fpnk = make_float2(q.x, -q.y);
            const float2 f1k = cadd(fpk, fpnk);
            const float2 f2k = csub(fpk, fpnk);
            const float2 t = cmul(f2k, tb.st_fwd[k]);
            const float xr = (f1k.x + t.x) * 0.5f, xi = (f1k.y + t.y) * 0.5f;
            const float yr = (f1k.x - t.x) * 0.5f, yi = (t.y - f1k.y) * 0.5f;
            if (k != nc - k) {
                mag[k] = sqrtf(xr * xr + xi * xi);
                ph[k] = atan2f(xi, xr);
            }
            mag[nc - k] = sqrtf(yr * yr + yi * yi);
            ph[nc - k] = atan2f(yi, yr);
        }
    }
}

// kernels may need more than the default 64 KiB of dynamic LDS at the largest FFT sizes
template <typename K> static void allow_big_lds(K kernel, bool &done) {
    if (done) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024 - 512);
    done = true;
}

void launch_analyze(const AnalyzeArgs &a, hipStream_t st) {
    const int grid = 8 * ((a.rows + 7) / 8) * a.Tn;
    const size_t lds = (size_t)a.tb.nc * sizeof(float2);
    hipLaunchKernelGGL(pv_analyze_kernel, dim3(grid), dim3(kFftThreads), lds, st, a);
}

// --------------------------------------------------------------------------------------------
// phase propagation: one workgroup per stream, sequential over (slice, channel) in the reference's
// processing order ch0, ch1, ch0, ... (phasevocoderprocess.cc:281-284); parallel over bins.
// The peak lists are per STREAM, not per channel: that is the reference's Impl-member quirk
// (phasevocoderimpl.h:236-238).
// --------------------------------------------------------------------------------------------
int phase_threads(int hs) { return hs < 1024 ? hs : 1024; }

size_t phase_lds_bytes(int hs, int C, int pkmax) {
    // smag[hs] sph[hs] prev_phase[C][hs] prev_out[C][hs] pk[2][pkmax] rot[pkmax] bnd[pkmax] wcnt[64] misc[4]
    return sizeof(float) * ((size_t)2 * hs + (size_t)2 * C * hs + (size_t)4 * pkmax + 64 + 4);
}

__global__ __launch_bounds__(1024) void pv_phase_kernel(const PhaseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int hs = a.hs, C = a.C, nt = blockDim.x, tid = threadIdx.x;
    const int bpt = hs / nt; // bins per thread (1 or more)
    float *smag = reinterpret_cast<float *>(smem_raw);
    float *sph = smag + hs;
    float *sprev_phase = sph + hs;           // [C][hs]
    float *sprev_out = sprev_phase + C * hs; // [C][hs]
    int *pk0 = reinterpret_cast<int *>(sprev_out + C * hs);
    int *pk1 = pk0 + a.pkmax;
    float *rot = reinterpret_cast<float *>(pk1 + a.pkmax);
    int *bnd = reinterpret_cast<int *>(rot + a.pkmax);
    int *wcnt = bnd + a.pkmax; // [64]
    int *misc = wcnt + 64;     // [0] = npeak of the current step

    const int s = blockIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = (nt + 63) >> 6;

    // load persistent state
    for (int i = tid; i < C * hs; i += nt) {
        sprev_phase[i] = a.st_prev_phase[(int64_t)s * C * hs + i];
        sprev_out[i] = a.st_prev_out[(int64_t)s * C * hs + i];
    }
    int nprev = a.st_npeaks[s];
    int *pk_prev = pk0, *pk_cur = pk1;
    for (int i = tid; i < nprev; i += nt) pk_prev[i] = a.st_peaks[(int64_t)s * a.pkmax + i];
    __syncthreads();

    const float hop_f = (float)a.hop;
    const double Nd = (double)a.N;

    for (int tl = 0; tl < a.Tn; ++tl) {
        const float pinc_f = (float)a.phase_inc[tl];
        for (int c = 0; c < C; ++c) {
            const int64_t base = (((int64_t)s * C + c) * a.Tc + tl) * a.HP;
            float *__restrict__ gph = a.phase + base;
            const float *__restrict__ gmag = a.mag + base;
            float *pp = sprev_phase + c * hs;
            float *po = sprev_out + c * hs;
            const bool first = (a.t0 + tl == 0) && (c == 0);

            if (a.coremode == 2) {
                // modifySliceIntRatio (:567-570): no state, not even firstentry
                for (int j = 0; j < bpt; ++j) {
                    const int i = tid + j * nt;
                    gph[i] = gph[i] * pinc_f / hop_f;
                }
                continue;
            }

            for (int j = 0; j < bpt; ++j) {
                const int i = tid + j * nt;
                sph[i] = gph[i];
                if (a.coremode == 1) smag[i] = gmag[i];
            }
            __syncthreads();

            int npeak = 0;
            if (a.coremode == 1) {
                // (i) peak picking, ordered compaction with wave ballots
                unsigned long long bal[4];
                for (int j = 0; j < bpt; ++j) {
                    const int b = tid + j * nt;
                    bool isp = false;
                    if (b >= 2 && b + 2 < hs) {
                        const float mb = smag[b];
                        isp = mb > smag[b - 1] && mb > smag[b - 2] && mb > smag[b + 1] && mb > smag[b + 2];
                    }
                    bal[j & 3] = __ballot(isp);
                    if (lane == 0) wcnt[j * nwaves + wave] = __popcll(bal[j & 3]);
                }
                __syncthreads();
                int total = 0;
                for (int q = 0; q < bpt * nwaves; ++q) total += wcnt[q];
                npeak = total;
                for (int j = 0; j < bpt; ++j) {
                    int off = 0;
                    for (int q = 0; q < j * nwaves + wave; ++q) off += wcnt[q];
                    const unsigned long long bm = bal[j & 3];
                    if ((bm >> lane) & 1ull) {
                        const int idx = off + __popcll(bm & ((1ull << lane) - 1ull));
                        pk_cur[idx] = tid + j * nt;
                    }
                }
                __syncthreads();
            }

            if (first) {
                // init branch (:606-616 / :718-728): output phase = input phase, state = input phase
                for (int j = 0; j < bpt; ++j) {
                    const int i = tid + j * nt;
                    const float tp = sph[i];
                    pp[i] = tp;
                    po[i] = tp;
                }
            } else if (a.coremode != 1 || npeak == 0 || nprev == 0) {
                // per-bin propagation (:620-636 / :732-748)
                for (int j = 0; j < bpt; ++j) {
                    const int i = tid + j * nt;
                    const float phi = sph[i];
                    const float omega = (float)((a.two_pi_hop * (double)i) / Nd);
                    const float d1 = phi - pp[i] - omega;
                    const float delta = (float)((double)omega + princarg((double)d1));
                    const float advance = delta * pinc_f / hop_f;
                    const float outp = (float)princarg((double)(po[i] + advance));
                    pp[i] = phi;
                    po[i] = outp;
                    gph[i] = outp;
                }
            } else {
                // (iv) phase locking (:640-699)
                for (int p = tid; p < npeak; p += nt) {
                    const int p2 = pk_cur[p];
                    // nearest previous peak, ties -> lower index (== the reference's monotone greedy walk)
                    int lo = 0, hi = nprev;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (pk_prev[mid] < p2) lo = mid + 1;
                        else hi = mid;
                    }
                    int sel;
                    if (lo == 0) sel = 0;
                    else if (lo == nprev) sel = nprev - 1;
                    else sel = (pk_prev[lo] - p2) < (p2 - pk_prev[lo - 1]) ? lo : lo - 1;
                    const int p1 = pk_prev[sel];
                    const float avg_p = (float)((double)(p1 + p2) * 0.5);
                    const float pomega = (float)((a.two_pi_hop * (double)(avg_p - 1)) / Nd);
                    const float phi2 = sph[p2];
                    const float d1 = phi2 - pp[p1] - pomega;
                    const float pdelta = (float)((double)pomega + princarg((double)d1));
                    const float tgt = (float)princarg((double)(po[p1] + (pdelta * pinc_f) / hop_f));
                    rot[p] = (float)princarg((double)(tgt - phi2));
                    if (p + 1 < npeak) bnd[p] = (p2 + pk_cur[p + 1] + 1) >> 1; // round(x.5) away from zero
                }
                __syncthreads();
                for (int j = 0; j < bpt; ++j) {
                    const int i = tid + j * nt;
                    // region = number of boundaries <= i
                    int lo = 0, hi = npeak - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (bnd[mid] <= i) lo = mid + 1;
                        else hi = mid;
                    }
                    const float phi = sph[i];
                    const float locked = (float)princarg((double)(phi + rot[lo]));
                    // pp/po of OTHER bins were read by the peak loop above (before the barrier) -- safe to update
                    pp[i] = phi;
                    po[i] = locked;
                    gph[i] = locked;
                }
            }
            if (a.coremode == 1) {
                int *t = pk_prev;
                pk_prev = pk_cur;
                pk_cur = t;
                nprev = npeak;
            }
            __syncthreads();
        }
    }

    // store persistent state
    for (int i = tid; i < C * hs; i += nt) {
        a.st_prev_phase[(int64_t)s * C * hs + i] = sprev_phase[i];
        a.st_prev_out[(int64_t)s * C * hs + i] = sprev_out[i];
    }
    for (int i = tid; i < nprev; i += nt) a.st_peaks[(int64_t)s * a.pkmax + i] = pk_prev[i];
    if (tid == 0) a.st_npeaks[s] = nprev;
    (void)misc;
}

void launch_phase(const PhaseArgs &a, int nstreams, hipStream_t st) {
    const int nt = phase_threads(a.hs);
    const size_t lds = phase_lds_bytes(a.hs, a.C, a.pkmax);
    static bool big = false;
    allow_big_lds(pv_phase_kernel, big);
    hipLaunchKernelGGL(pv_phase_kernel, dim3(nstreams), dim3(nt), lds, st, a);
}

// --------------------------------------------------------------------------------------------
// synthesis
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFftThreads) void pv_synth_kernel(const SynthArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const DevTables &tb = a.tb;
    const int N = tb.N, hs = tb.hs, nc = tb.nc, nt = blockDim.x;
    float2 *buf = reinterpret_cast<float2 *>(smem_raw); // [nc]
    float2 *X = buf + nc;                               // [nc + 1]
    int row, tl;
    if (!block_to_row_slice(a.Tn, a.rows, row, tl)) return;
    const float *__restrict__ mag = a.mag + ((int64_t)row * a.Tc + tl) * tb.HP;
    const float *__restrict__ ph = a.phase + ((int64_t)row * a.Tc + tl) * tb.HP;
    const double Nd = (double)N;

    for (int k = threadIdx.x; k <= hs; k += nt) {
        float mg, p;
        if (a.do_freq_comp) {
            // freqCompSlice (:869-916): both branches are pure gathers from the pre-call arrays
            if (a.freq_comp > 1.0f) {
                const int src = __float2int_rn((float)k * a.freq_comp);
                if (src > hs) {
                    mg = 0.f;
                    p = 0.f;
                } else {
                    mg = mag[src];
                    p = ph[src] + (float)((a.two_pi_hop * (double)(k - src)) / Nd);
                }
            } else if (k < hs) {
                const int src = __float2int_rn((float)k * a.freq_comp);
                mg = mag[src];
                p = ph[src] + (float)((a.two_pi_hop * (double)(k - src)) / Nd);
            } else {
                mg = mag[k];
                p = ph[k];
            }
            mg *= a.fixed_gain;
        } else {
            mg = mag[k];
            p = ph[k];
        }
        if (a.robotic) p = 0.f;
        mg *= a.inv_n;
        X[k] = make_float2(mg * cosf(p), mg * sinf(p));
    }
    __syncthreads();

    // kiss_fftri pre-pass (kiss_fftr.c:134-157), scattered straight into butterfly order
    for (int k = threadIdx.x; k <= nc / 2; k += nt) {
        if (k == 0) {
            buf[tb.iperm[0]] = make_float2(X[0].x + X[nc].x, X[0].x - X[nc].x);
        } else {
            const float2 fk = X[k];
            const float2 q = X[nc - k];
            const float2 fnkc = make_float2(q.x, -q.y);
            const float2 fek = cadd(fk, fnkc);
            const float2 t = csub(fk, fnkc);
            const float2 fok = cmul(t, tb.st_inv[k]);
            const float2 u = cadd(fek, fok);
            float2 v = csub(fek, fok);
            v.y = v.y * -1.f;
            if (k != nc - k) buf[tb.iperm[k]] = u;
            buf[tb.iperm[nc - k]] = v;
        }
    }
    __syncthreads();
    fft_stages<true>(buf, tb, tb.tw_inv);

    // ifftshift + synthesis window (phasevocoderimpl.h:183-198)
    const float *fb = reinterpret_cast<const float *>(buf);
    const int slot = (int)((a.t0 + tl) & (int64_t)(a.FR - 1));
    float *__restrict__ out = a.frames + ((int64_t)row * a.FR + slot) * N;
    const float *__restrict__ w = tb.window;
    for (int i = threadIdx.x; i < N; i += nt) out[i] = fb[(i + hs) & (N - 1)] * w[i];
}

void launch_synth(const SynthArgs &a, hipStream_t st) {
    const int grid = 8 * ((a.rows + 7) / 8) * a.Tn;
    const size_t lds = (size_t)(2 * a.tb.nc + 1) * sizeof(float2);
    static bool big = false;
    allow_big_lds(pv_synth_kernel, big);
    hipLaunchKernelGGL(pv_synth_kernel, dim3(grid), dim3(kFftThreads), lds, st, a);
}

// --------------------------------------------------------------------------------------------
// overlap-add + normalise + resample
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTileOut) void pv_ola_kernel(const OlaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *ola = reinterpret_cast<float *>(smem_raw);           // [lds_floats]
    float *stab = ola + a.lds_floats;                           // [sinc_len]
    int *sP = reinterpret_cast<int *>(stab + a.sinc_len);       // [kMaxTileFrames] P_t - n_lo
    const int tile_i = blockIdx.x, row = blockIdx.y, nt = blockDim.x, tid = threadIdx.x;
    const OlaTile tile = a.tiles[tile_i];
    const int N = a.N;

    if (tid < tile.t_cnt) sP[tid] = (int)(a.P[tile.p_off + tid] - tile.n_lo);
    if (a.resample)
        for (int i = tid; i < a.sinc_len; i += nt) stab[i] = a.sinc[i];
    __syncthreads();

    // y[n] = (sum_t frame_t[n - P_t]) / (delta[n] + sum_t gain*w[n - P_t]), ascending t, starting from 0.0f
    // (== outputAccumulator / windowAccumulator at the moment writeSlice divides them)
    const float *__restrict__ fr = a.frames + (int64_t)row * a.FR * N;
    for (int i = tid; i < tile.n_cnt; i += nt) {
        const int64_t n = tile.n_lo + i;
        float y = 0.f;
        if (n >= 0) {
            float acc = 0.f;
            float wacc = n == 0 ? 1.f : 0.f;
            for (int j = 0; j < tile.t_cnt; ++j) {
                const int off = i - sP[j]; // n - P_t
                if (off >= 0 && off < N) {
                    const int slot = (tile.t_first + j) & (a.FR - 1);
                    acc += fr[(int64_t)slot * N + off];
                    wacc += a.window[off] * a.win_gain;
                }
            }
            y = acc / wacc;
        }
        ola[i] = y;
    }
    __syncthreads();

    if (tid >= tile.kcnt) return;
    const int64_t k = tile.k0 + tid;
    float *__restrict__ out = a.out + (int64_t)row * a.out_stride_row + (k - a.k_base);
    if (!a.resample) {
        *out = ola[tid];
        return;
    }
    // position of output k in the OLA stream: last_sample = filt_len/2 + floor(k*num/den),
    // samp_frac_num = (k*num) mod den  (closed form of resample.c:548-554 from skip_zeros :1225)
    const unsigned long long tot = (unsigned long long)k * a.num;
    const int64_t pos = (int64_t)(a.filt_len / 2) + (int64_t)(tot / a.den);
    const uint32_t frac_num = (uint32_t)(tot % a.den);
    const int i0 = (int)(pos - a.filt_len + 1 - tile.n_lo); // LDS index of tap j = 0
    const int NF = a.filt_len;
    if (a.interp) {
        const uint32_t ov = (uint32_t)a.oversample;
        const int offset = (int)(frac_num * ov / a.den);
        const float frac = ((float)((frac_num * ov) % a.den)) / a.den;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float *t = stab + 4 + (int)ov - offset; // tap j reads t[j*ov - 2 .. j*ov + 1]
        for (int j = 0; j < NF; ++j) {
            const float x = ola[i0 + j];
            const float *tj = t + j * (int)ov;
            a0 += x * tj[-2];
            a1 += x * tj[-1];
            a2 += x * tj[0];
            a3 += x * tj[1];
        }
        // cubic_coef (resample.c:339-351)
        const float c0 = -0.16667f * frac + 0.16667f * frac * frac * frac;
        const float c1 = frac + 0.5f * frac * frac - 0.5f * frac * frac * frac;
        const float c3 = -0.33333f * frac + 0.5f * frac * frac - 0.16667f * frac * frac * frac;
        const float c2 = (float)(1. - c0 - c1 - c3);
        *out = (c0 * a0) + (c1 * a1) + (c2 * a2) + (c3 * a3);
    } else {
        float sum = 0.f;
        const float *t = stab + frac_num * (uint32_t)NF;
        for (int j = 0; j < NF; ++j) sum += ola[i0 + j] * t[j];
        *out = sum;
    }
}

void launch_ola(const OlaArgs &a, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)a.lds_floats + a.sinc_len) + sizeof(int) * kMaxTileFrames;
    hipLaunchKernelGGL(pv_ola_kernel, dim3(a.ntiles, a.rows), dim3(kTileOut), lds, st, a);
}

} // namespace pv
