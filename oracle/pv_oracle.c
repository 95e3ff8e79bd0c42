/* oracle/pv_oracle.c -- TEST INFRASTRUCTURE ONLY (see pv_oracle.h).
 *
 * CPU restatement of the reference phase-vocoder path.  Written from scratch from the
 * behaviour documented in SURVEY.md section 8(a); every function cites the reference
 * file:line whose arithmetic it restates (paths relative to /root/reference/).
 *
 * The goal is BIT-identity with the compiled reference on x86-64 (gcc, no FMA, SSE
 * scalar float, glibc libm), so expression shapes (which operand is float, which is
 * double, evaluation order) are part of the specification.  Build with
 * -ffp-contract=off and without -ffast-math (oracle/Makefile).
 */
#include "pv_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------------- */
static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "pv_oracle: out of memory\n"); abort(); }
    return p;
}

/* src/common/system/sys.h:84,91 : princarg(a) = mod(a + pi, -2pi) + pi with
 * mod(x, y) = x - y*floor(x/y), all in double. */
double pvo_princarg(double a) {
    double x = a + M_PI;
    double y = -2.0 * M_PI;
    return (x - (y * floor(x / y))) + M_PI;
}

/* ------------------------------------------------------------------------------------------
 * Periodic Hann window  (src/common/dsp/windowfunc.h:129-131,159-169; area :152-156)
 * ---------------------------------------------------------------------------------------- */
void pvo_hann(int n, float *w, float *area) {
    const float a0 = 0.50f, a1 = 0.50f, a2 = 0.0f, a3 = 0.0f;
    for (int i = 0; i < n; ++i) {
        float m = 1.0f;
        double e = (a0 - a1 * cos(2 * M_PI * i / n) + a2 * cos(4 * M_PI * i / n) - a3 * cos(6 * M_PI * i / n));
        m = (float)(m * e);
        w[i] = m;
    }
    float acc = 0;
    for (int i = 0; i < n; ++i) acc += w[i];
    acc /= n;
    if (area) *area = acc;
}

/* ------------------------------------------------------------------------------------------
 * Mixed-radix (4s then 2s) decimation-in-time complex FFT with kissfft's operation order.
 * Restates src/common/kissfft/kiss_fft.c: kf_factor :292-316, twiddles :341-347,
 * kf_work :250-288 (as an explicit permutation + level-by-level butterflies, which is
 * arithmetic-identical because butterflies of one level are independent), kf_bfly4 :59-103,
 * kf_bfly2 :36-57.  Only power-of-two sizes are needed by the phase vocoder.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float r, i; } cpx;

typedef struct {
    int n, inverse;
    int nlev;
    int radix[32], m[32], fstride[32];
    int *perm;  /* out index -> in index */
    cpx *tw;    /* n twiddles */
} cfft;

static void cfft_perm(const cfft *p, int lev, int out_base, int in_base, int *perm) {
    int r = p->radix[lev], m = p->m[lev], fs = p->fstride[lev];
    if (m == 1) {
        for (int q = 0; q < r; ++q) perm[out_base + q] = in_base + q * fs;
    } else {
        for (int q = 0; q < r; ++q) cfft_perm(p, lev + 1, out_base + q * m, in_base + q * fs, perm);
    }
}

static cfft *cfft_new(int n, int inverse) {
    cfft *p = (cfft *)xcalloc(1, sizeof(cfft));
    p->n = n;
    p->inverse = inverse;
    p->tw = (cpx *)xcalloc(n, sizeof(cpx));
    for (int i = 0; i < n; ++i) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / n;
        if (inverse) phase *= -1;
        p->tw[i].r = (float)cos(phase);
        p->tw[i].i = (float)sin(phase);
    }
    /* factorisation: 4s first, then 2s (then odd primes; not needed for powers of two) */
    int rem = n, r = 4, lev = 0, fs = 1;
    double floor_sqrt = floor(sqrt((double)n));
    do {
        while (rem % r) {
            switch (r) {
            case 4: r = 2; break;
            case 2: r = 3; break;
            default: r += 2; break;
            }
            if (r > floor_sqrt) r = rem;
        }
        rem /= r;
        if (r != 2 && r != 4) { fprintf(stderr, "pv_oracle: FFT size %d not a power of two\n", n); abort(); }
        p->radix[lev] = r;
        p->m[lev] = rem;
        p->fstride[lev] = fs;
        fs *= r;
        ++lev;
    } while (rem > 1);
    p->nlev = lev;
    p->perm = (int *)xcalloc(n, sizeof(int));
    cfft_perm(p, 0, 0, 0, p->perm);
    return p;
}

static void cfft_free(cfft *p) {
    if (!p) return;
    free(p->perm);
    free(p->tw);
    free(p);
}

static inline cpx cmul(cpx a, cpx b) {
    cpx m;
    m.r = a.r * b.r - a.i * b.i;
    m.i = a.r * b.i + a.i * b.r;
    return m;
}
static inline cpx cadd(cpx a, cpx b) { cpx m; m.r = a.r + b.r; m.i = a.i + b.i; return m; }
static inline cpx csub(cpx a, cpx b) { cpx m; m.r = a.r - b.r; m.i = a.i - b.i; return m; }

static void bfly2(const cfft *p, cpx *F, int fstride, int m) {
    for (int k = 0; k < m; ++k) {
        cpx t = cmul(F[k + m], p->tw[k * fstride]);
        F[k + m] = csub(F[k], t);
        F[k] = cadd(F[k], t);
    }
}

static void bfly4(const cfft *p, cpx *F, int fstride, int m) {
    for (int k = 0; k < m; ++k) {
        cpx s0 = cmul(F[k + m], p->tw[k * fstride]);
        cpx s1 = cmul(F[k + 2 * m], p->tw[2 * k * fstride]);
        cpx s2 = cmul(F[k + 3 * m], p->tw[3 * k * fstride]);
        cpx s5 = csub(F[k], s1);
        cpx f0 = cadd(F[k], s1);
        cpx s3 = cadd(s0, s2);
        cpx s4 = csub(s0, s2);
        F[k + 2 * m] = csub(f0, s3);
        F[k] = cadd(f0, s3);
        if (p->inverse) {
            F[k + m].r = s5.r - s4.i;
            F[k + m].i = s5.i + s4.r;
            F[k + 3 * m].r = s5.r + s4.i;
            F[k + 3 * m].i = s5.i - s4.r;
        } else {
            F[k + m].r = s5.r + s4.i;
            F[k + m].i = s5.i - s4.r;
            F[k + 3 * m].r = s5.r - s4.i;
            F[k + 3 * m].i = s5.i + s4.r;
        }
    }
}

/* out and in must not alias */
static void cfft_run(const cfft *p, const cpx *in, cpx *out) {
    for (int i = 0; i < p->n; ++i) out[i] = in[p->perm[i]];
    for (int lev = p->nlev - 1; lev >= 0; --lev) {
        int r = p->radix[lev], m = p->m[lev], fs = p->fstride[lev];
        int blk = r * m;
        for (int base = 0; base < p->n; base += blk) {
            if (r == 4) bfly4(p, out + base, fs, m);
            else bfly2(p, out + base, fs, m);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Real FFT via half-size complex FFT  (src/common/kissfft/kiss_fftr.c: alloc :27-64,
 * kiss_fftr :67-121, kiss_fftri :123-159) + polar wrappers (src/common/dsp/FFT.cc:2617-2631,
 * :2711-2721).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int n, nc;
    cfft *fwd, *inv;
    cpx *st_fwd, *st_inv; /* "super twiddles" */
    cpx *tmp, *packed;
} rfft;

static rfft *rfft_new(int n) {
    rfft *p = (rfft *)xcalloc(1, sizeof(rfft));
    p->n = n;
    p->nc = n / 2;
    p->fwd = cfft_new(p->nc, 0);
    p->inv = cfft_new(p->nc, 1);
    p->st_fwd = (cpx *)xcalloc(p->nc, sizeof(cpx));
    p->st_inv = (cpx *)xcalloc(p->nc, sizeof(cpx));
    for (int i = 0; i < p->nc; ++i) {
        double phase = -3.14159265358979323846264338327 * ((double)i / p->nc + .5);
        p->st_fwd[i].r = (float)cos(phase);
        p->st_fwd[i].i = (float)sin(phase);
        phase *= -1;
        p->st_inv[i].r = (float)cos(phase);
        p->st_inv[i].i = (float)sin(phase);
    }
    p->tmp = (cpx *)xcalloc(p->nc + 1, sizeof(cpx));
    p->packed = (cpx *)xcalloc(p->nc + 2, sizeof(cpx));
    return p;
}

static void rfft_free(rfft *p) {
    if (!p) return;
    cfft_free(p->fwd);
    cfft_free(p->inv);
    free(p->st_fwd);
    free(p->st_inv);
    free(p->tmp);
    free(p->packed);
    free(p);
}

static void rfft_forward(rfft *p, const float *timedata, cpx *X) {
    const int nc = p->nc;
    cfft_run(p->fwd, (const cpx *)timedata, p->tmp);
    cpx tdc = p->tmp[0];
    X[0].r = tdc.r + tdc.i;
    X[nc].r = tdc.r - tdc.i;
    X[nc].i = X[0].i = 0;
    for (int k = 1; k <= nc / 2; ++k) {
        cpx fpk = p->tmp[k];
        cpx fpnk;
        fpnk.r = p->tmp[nc - k].r;
        fpnk.i = -p->tmp[nc - k].i;
        cpx f1k = cadd(fpk, fpnk);
        cpx f2k = csub(fpk, fpnk);
        cpx tw = cmul(f2k, p->st_fwd[k]);
        X[k].r = (float)((f1k.r + tw.r) * .5);
        X[k].i = (float)((f1k.i + tw.i) * .5);
        X[nc - k].r = (float)((f1k.r - tw.r) * .5);
        X[nc - k].i = (float)((tw.i - f1k.i) * .5);
    }
}

static void rfft_inverse(rfft *p, const cpx *X, float *timedata) {
    const int nc = p->nc;
    p->tmp[0].r = X[0].r + X[nc].r;
    p->tmp[0].i = X[0].r - X[nc].r;
    for (int k = 1; k <= nc / 2; ++k) {
        cpx fk = X[k];
        cpx fnkc;
        fnkc.r = X[nc - k].r;
        fnkc.i = -X[nc - k].i;
        cpx fek = cadd(fk, fnkc);
        cpx t = csub(fk, fnkc);
        cpx fok = cmul(t, p->st_inv[k]);
        p->tmp[k] = cadd(fek, fok);
        p->tmp[nc - k] = csub(fek, fok);
        p->tmp[nc - k].i *= -1;
    }
    cfft_run(p->inv, p->tmp, (cpx *)timedata);
}

static void rfft_forward_polar(rfft *p, const float *in, float *mag, float *phase) {
    const int hs = p->n / 2;
    rfft_forward(p, in, p->packed);
    for (int i = 0; i <= hs; ++i)
        mag[i] = sqrtf(p->packed[i].r * p->packed[i].r + p->packed[i].i * p->packed[i].i);
    for (int i = 0; i <= hs; ++i) phase[i] = atan2f(p->packed[i].i, p->packed[i].r);
}

static void rfft_inverse_polar(rfft *p, const float *mag, const float *phase, float *out) {
    const int hs = p->n / 2;
    for (int i = 0; i <= hs; ++i) {
        p->packed[i].r = mag[i] * cosf(phase[i]);
        p->packed[i].i = mag[i] * sinf(phase[i]);
    }
    rfft_inverse(p, p->packed, out);
}

void pvo_forward_polar(int n, const float *in, float *mag, float *phase) {
    rfft *p = rfft_new(n);
    rfft_forward_polar(p, in, mag, phase);
    rfft_free(p);
}

void pvo_inverse_polar(int n, const float *mag, const float *phase, float *out) {
    rfft *p = rfft_new(n);
    rfft_inverse_polar(p, mag, phase, out);
    rfft_free(p);
}

/* ------------------------------------------------------------------------------------------
 * Speex-style resampler, quality 4, one mono channel.
 * Restates src/common/dsp/resampler.cc:696-825 (RS_Speex) and src/common/speex/resample.c:
 * quality table :285-297, kaiser8 :233-240, compute_func :300-322, sinc :325-337,
 * cubic_coef :339-351, direct :353-401, interpolate :462-560, update_filter :661-913,
 * process_native :986-1059, set_rate_frac :1117-1158, skip_zeros :1220-1229.
 * ---------------------------------------------------------------------------------------- */
static const double kaiser8[36] = {
    0.99635258, 1.00000000, 0.99635258, 0.98548012, 0.96759014, 0.94302200, 0.91223751, 0.87580811, 0.83439927,
    0.78875245, 0.73966538, 0.68797126, 0.63451750, 0.58014482, 0.52566725, 0.47185369, 0.41941150, 0.36897272,
    0.32108304, 0.27619388, 0.23465776, 0.19672670, 0.16255380, 0.13219758, 0.10562887, 0.08273982, 0.06335451,
    0.04724088, 0.03412321, 0.02369490, 0.01563093, 0.00959968, 0.00527363, 0.00233883, 0.00050000, 0.00000000};
#define K8_OVERSAMPLE 32
#define Q4_BASE_LEN 64
#define Q4_OVERSAMPLE 8
#define Q4_DOWN_BW 0.921f
#define Q4_UP_BW 0.940f

static double kaiser_window(float x) {
    float y, frac;
    double interp[4];
    int ind;
    y = x * K8_OVERSAMPLE;
    ind = (int)floor(y);
    frac = (y - ind);
    interp[3] = -0.1666666667 * frac + 0.1666666667 * (frac * frac * frac);
    interp[2] = frac + 0.5 * (frac * frac) - 0.5 * (frac * frac * frac);
    interp[0] = -0.3333333333 * frac + 0.5 * (frac * frac) - 0.1666666667 * (frac * frac * frac);
    interp[1] = 1.f - interp[3] - interp[2] - interp[0];
    return interp[0] * kaiser8[ind] + interp[1] * kaiser8[ind + 1] + interp[2] * kaiser8[ind + 2] +
           interp[3] * kaiser8[ind + 3];
}

static float windowed_sinc(float cutoff, float x, int N) {
    float xx = x * cutoff;
    if (fabsf(x) < 1e-6)
        return cutoff;
    else if (fabsf(x) > .5 * N)
        return 0;
    return cutoff * sin(M_PI * xx) / (M_PI * xx) * kaiser_window(fabs(2. * x / N));
}

struct pvo_resampler {
    uint32_t num_rate, den_rate; /* speex naming: step = num/den input samples per output */
    uint32_t filt_len, oversample;
    int int_advance, frac_advance;
    float cutoff;
    int interp;   /* 1: interpolated sinc table, 0: direct table */
    int started;
    int32_t last_sample;
    uint32_t samp_frac_num;
    float *mem;         /* filt_len-1 history samples */
    float *table;
    int table_len;
    float lastratio;
    int initial;
};

static uint32_t gcd_u32(uint32_t a, uint32_t b) {
    while (b) { uint32_t t = b; b = a % b; a = t; }
    return a;
}

static void res_update_filter(pvo_resampler *r) {
    uint32_t old_length = r->filt_len;
    r->oversample = Q4_OVERSAMPLE;
    r->filt_len = Q4_BASE_LEN;
    if (r->num_rate > r->den_rate) {
        r->cutoff = Q4_DOWN_BW * r->den_rate / r->num_rate;
        r->filt_len = (uint32_t)ceil(r->filt_len * ((double)r->num_rate / (double)r->den_rate));
        r->filt_len &= (~0x3u);
        if (2 * r->den_rate < r->num_rate) r->oversample >>= 1;
        if (4 * r->den_rate < r->num_rate) r->oversample >>= 1;
        if (8 * r->den_rate < r->num_rate) r->oversample >>= 1;
        if (16 * r->den_rate < r->num_rate) r->oversample >>= 1;
        if (r->oversample < 1) r->oversample = 1;
    } else {
        r->cutoff = Q4_UP_BW;
    }
    free(r->table);
    if (r->den_rate <= r->oversample) {
        r->interp = 0;
        r->table_len = (int)(r->filt_len * r->den_rate);
        r->table = (float *)xcalloc(r->table_len, sizeof(float));
        for (uint32_t i = 0; i < r->den_rate; i++)
            for (int j = 0; j < (int)r->filt_len; j++)
                r->table[i * r->filt_len + j] =
                    windowed_sinc(r->cutoff, ((j - (int)r->filt_len / 2 + 1) - ((float)i) / r->den_rate), r->filt_len);
    } else {
        r->interp = 1;
        r->table_len = (int)(r->filt_len * r->oversample + 8);
        r->table = (float *)xcalloc(r->table_len, sizeof(float));
        for (int i = -4; i < (int)(r->oversample * r->filt_len + 4); i++)
            r->table[i + 4] = windowed_sinc(r->cutoff, (i / (float)r->oversample - r->filt_len / 2), r->filt_len);
    }
    r->int_advance = r->num_rate / r->den_rate;
    r->frac_advance = r->num_rate % r->den_rate;
    if (!r->started) {
        free(r->mem);
        r->mem = (float *)xcalloc(r->filt_len - 1, sizeof(float));
    } else if (r->filt_len != old_length) {
        /* ratio change after start: the reference's "magic samples" path
         * (resample.c:819-912).  The phase vocoder never changes ratio after construction. */
        fprintf(stderr, "pv_oracle: resampler ratio change after start is not restated\n");
        abort();
    }
}

pvo_resampler *pvo_res_create(void) {
    pvo_resampler *r = (pvo_resampler *)xcalloc(1, sizeof(*r));
    /* speex_resampler_init_frac(1, 1, 1, ...) then RS_Speex::reset() (channelinfo.cc:97) */
    r->num_rate = r->den_rate = 1;
    r->filt_len = 0;
    r->cutoff = 1.f;
    res_update_filter(r);
    r->lastratio = -1.0f;
    r->initial = 1;
    r->last_sample = 0;
    r->samp_frac_num = 0;
    return r;
}

void pvo_res_destroy(pvo_resampler *r) {
    if (!r) return;
    free(r->mem);
    free(r->table);
    free(r);
}

static void res_setratio(pvo_resampler *r, float ratio) {
    uint32_t big = 272408136U;
    uint32_t denom = 1, num = 1;
    if (ratio < 1.f) {
        denom = big;
        double dnum = (double)big * (double)ratio;
        num = (uint32_t)dnum;
    } else if (ratio > 1.f) {
        num = big;
        double ddenom = (double)big / (double)ratio;
        denom = (uint32_t)ddenom;
    }
    /* speex_resampler_set_rate_frac(st, ratio_num = denom, ratio_den = num) */
    if (!(r->num_rate == denom && r->den_rate == num)) {
        uint32_t old_den = r->den_rate;
        r->num_rate = denom;
        r->den_rate = num;
        uint32_t g = gcd_u32(r->num_rate, r->den_rate);
        r->num_rate /= g;
        r->den_rate /= g;
        if (old_den > 0) {
            r->samp_frac_num = r->samp_frac_num * r->den_rate / old_den;
            if (r->samp_frac_num >= r->den_rate) r->samp_frac_num = r->den_rate - 1;
        }
        res_update_filter(r);
    }
    r->lastratio = ratio;
    if (r->initial) {
        r->last_sample = r->filt_len / 2;
        r->initial = 0;
    }
}

static void cubic_coef(float frac, float interp[4]) {
    interp[0] = -0.16667f * frac + 0.16667f * frac * frac * frac;
    interp[1] = frac + 0.5f * frac * frac - 0.5f * frac * frac * frac;
    interp[3] = -0.33333f * frac + 0.5f * frac * frac - 0.16667f * frac * frac * frac;
    interp[2] = 1. - interp[0] - interp[1] - interp[3];
}

static inline float res_sample(const pvo_resampler *r, const float *in, int idx) {
    /* idx relative to the start of the current input chunk; negative -> history */
    return idx < 0 ? r->mem[(int)r->filt_len - 1 + idx] : in[idx];
}

int pvo_res_process(pvo_resampler *r, const float *in, int incount, float ratio, float *out) {
    if (ratio != r->lastratio) res_setratio(r, ratio);
    uint32_t in_len = (uint32_t)incount;
    uint32_t out_len = (uint32_t)lrintf(ceilf(incount * ratio));
    const int N = (int)r->filt_len;
    int out_sample = 0;
    int last_sample = r->last_sample;
    uint32_t frac_num = r->samp_frac_num;
    r->started = 1;
    while (!(last_sample >= (int)in_len || out_sample >= (int)out_len)) {
        if (r->interp) {
            float accum[4] = {0.f, 0.f, 0.f, 0.f};
            float interp[4];
            int offset = frac_num * r->oversample / r->den_rate;
            float frac = ((float)((frac_num * r->oversample) % r->den_rate)) / r->den_rate;
            for (int j = 0; j < N; j++) {
                float x = res_sample(r, in, last_sample - N + 1 + j);
                const float *t = r->table + 4 + (j + 1) * (int)r->oversample - offset;
                accum[0] += x * t[-2];
                accum[1] += x * t[-1];
                accum[2] += x * t[0];
                accum[3] += x * t[1];
            }
            cubic_coef(frac, interp);
            out[out_sample] = (interp[0] * accum[0]) + (interp[1] * accum[1]) + (interp[2] * accum[2]) +
                              (interp[3] * accum[3]);
        } else {
            float sum = 0;
            for (int j = 0; j < N; j++)
                sum += res_sample(r, in, last_sample - N + 1 + j) * r->table[frac_num * r->filt_len + j];
            out[out_sample] = sum;
        }
        out_sample++;
        last_sample += r->int_advance;
        frac_num += r->frac_advance;
        if (frac_num >= r->den_rate) {
            frac_num -= r->den_rate;
            last_sample++;
        }
    }
    if (last_sample < (int)in_len) in_len = last_sample;
    last_sample -= in_len;
    /* history <- last N-1 consumed samples */
    {
        int j;
        float *nm = (float *)xcalloc(N - 1 > 0 ? N - 1 : 1, sizeof(float));
        for (j = 0; j < N - 1; j++) nm[j] = res_sample(r, in, (int)in_len - (N - 1) + j);
        memcpy(r->mem, nm, sizeof(float) * (N - 1));
        free(nm);
    }
    r->last_sample = last_sample;
    r->samp_frac_num = frac_num;
    return out_sample;
}

void pvo_res_info(const pvo_resampler *r, unsigned *num, unsigned *den, int *filt_len, int *oversample, int *interp) {
    if (num) *num = r->num_rate;
    if (den) *den = r->den_rate;
    if (filt_len) *filt_len = (int)r->filt_len;
    if (oversample) *oversample = (int)r->oversample;
    if (interp) *interp = r->interp;
}

int pvo_res_table(const pvo_resampler *r, float *dst, int max) {
    int n = r->table_len < max ? r->table_len : max;
    if (dst) memcpy(dst, r->table, sizeof(float) * n);
    return r->table_len;
}

/* ------------------------------------------------------------------------------------------
 * Ring buffer with the reference's capacity semantics (src/common/base/circularqueue.h:
 * capacity n, storage n+1; write :311-345 short-writes with a warning, read :262-293).
 * ---------------------------------------------------------------------------------------- */
typedef struct { float *buf; int size, r, w; } ring;

static void ring_init(ring *q, int n) { q->buf = (float *)xcalloc(n + 1, sizeof(float)); q->size = n + 1; q->r = q->w = 0; }
static void ring_free(ring *q) { free(q->buf); }
static int ring_readspace(const ring *q) { return q->w >= q->r ? q->w - q->r : q->w + q->size - q->r; }
static int ring_writespace(const ring *q) { int s = q->r + q->size - q->w - 1; if (s >= q->size) s -= q->size; return s; }
static int ring_write(ring *q, const float *src, int n) {
    int a = ring_writespace(q);
    if (n > a) n = a;
    for (int i = 0; i < n; ++i) { q->buf[q->w] = src[i]; if (++q->w == q->size) q->w = 0; }
    return n;
}
static int ring_peek(const ring *q, float *dst, int n) {
    int a = ring_readspace(q);
    if (n > a) { memset(dst + a, 0, sizeof(float) * (n - a)); n = a; }
    int r = q->r;
    for (int i = 0; i < n; ++i) { dst[i] = q->buf[r]; if (++r == q->size) r = 0; }
    return n;
}
static int ring_discard(ring *q, int n) {
    int a = ring_readspace(q);
    if (n > a) n = a;
    q->r = (q->r + n) % q->size;
    return n;
}
static int ring_read(ring *q, float *dst, int n) {
    int a = ring_readspace(q);
    if (n > a) n = a;
    for (int i = 0; i < n; ++i) { dst[i] = q->buf[q->r]; if (++q->r == q->size) q->r = 0; }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * The phase vocoder proper
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    ring inbuf, outbuf;
    float *mag, *phase, *prev_phase, *prev_outphase, *locked_phase;
    float *oacc, *wacc; /* outputAccumulator / windowAccumulator */
    float *frame_t;     /* interfacebuffer */
    float *frame_f;     /* internalbuffer */
    rfft *fft;
    pvo_resampler *res;
    size_t prev_increment;
    long slicecnt;
} chan;

/* Rosenberg glottal-pulse carrier (src/common/gen/rosenberg.cc:19-53) and the 3-voice chord
 * (src/common/gen/rosenbergchord.cc:19-43) */
typedef struct {
    int period, n1, n2, phase;
    float inv_n1, inv_2n2;
} rsb;

static void rsb_init(rsb *g, float sample_rate, float freq, float alpha, float beta) {
    g->period = round(1.f / freq * sample_rate);
    g->phase = 0;
    g->n1 = round(alpha * g->period);
    g->inv_n1 = 1.f / (float)(g->n1);
    g->n2 = round(beta * g->period);
    g->inv_2n2 = 0.5 / (float)(g->n2);
}

static float rsb_next(rsb *g) {
    float res = 0;
    if (g->phase <= g->n1) {
        res = 0.5 * (1 - cosf(M_PI * g->phase * g->inv_n1));
    } else if (g->phase - g->n1 <= g->n2) {
        res = cosf(M_PI * (g->phase - g->n1) * g->inv_2n2);
    } else {
        res = 0;
    }
    if (++g->phase > g->period) g->phase = 0;
    return res;
}

struct pvo {
    pvo_config cfg;
    chan *car;        /* carrier channelinfo per channel (vocoder modes) */
    rsb *gen;         /* [channels] single-voice generators */
    rsb *chord;       /* [channels][3] */
    int opt_gender, opt_formant, opt_robotic, opt_whisper, opt_cepstral;
    float *cep, *cenv; /* scratch of the cepstral formant shift */
    float time_ratio, pitch_scale;
    size_t N, hop, outbuf_size;
    int hop_out_nominal;
    float *win; float win_area;
    chan *ch;
    /* state that is process-global (function statics / Impl members) in the reference, kept per
     * instance here == "each stream behaves like a fresh reference process"  */
    int firstentry;
    int *peak, npeak, *prev_peak, nprev_peak;
    float recovery, divergence;
    float *resamplebuf; size_t resamplebuf_size;
    /* record of increments for planner pinning */
    int *rec_shift, *rec_phase; long nrec, caprec;
    /* set once a slice's shift increment exceeds N: writeSlice (phasevocoderprocess.cc:1181-1190) then calls
     * memmove with (N - shiftIncrement) wrapped to a huge size_t -- undefined behaviour in the reference, only
     * reachable with a caller-chosen hop.  The oracle stops there instead of reproducing the overflow. */
    int undefined;
};

static float hs_ratio(const pvo *h) { return h->time_ratio * h->pitch_scale; }

/* phasevocoderimpl.cc:149-157 (unqualified abs(float) resolves to the float overload in the
 * oracle build -- verified against oracle/_ref by tests) */
static int is_int_ratio(const pvo *h) {
    float efr = hs_ratio(h);
    return fabsf(efr - floorf(efr)) <= 0.001;
}

static size_t nextpow2(size_t v) {
    if (!(v & (v - 1))) return v;
    int bits = 0;
    while (v) { ++bits; v >>= 1; }
    return (size_t)1 << bits;
}

pvo *pvo_create(const pvo_config *cfg) {
    pvo *h = (pvo *)xcalloc(1, sizeof(pvo));
    h->cfg = *cfg;
    /* phasevocoder.cc:24-26,31-42 */
    h->time_ratio = cfg->time_ratio;
    h->pitch_scale = cfg->pitch_semitones != 0 ? pow(2.0, cfg->pitch_semitones / 12) : 1.0;
    h->opt_gender = cfg->mode == PVO_GENDER_CHANGE;
    h->opt_formant = cfg->mode == PVO_FORMANT_PRESERVE;
    h->opt_robotic = cfg->mode == PVO_ROBOTIC;
    h->opt_whisper = cfg->mode == PVO_WHISPER;
    h->opt_cepstral = cfg->mode == PVO_FORMANT_CEPSTRAL;

    /* calculateSizes, phasevocoderimpl.cc:169-263 */
    size_t windowSize = nextpow2((size_t)cfg->fftsize);
    if (h->pitch_scale <= 0.0) h->pitch_scale = 1.0;
    if (h->time_ratio <= 0.0) h->time_ratio = 1.0;
    float hsr = hs_ratio(h);
    size_t inHop, outHop;
    if (cfg->hopsize > 0) {
        inHop = cfg->hopsize;
        outHop = (size_t)(int)(floor(inHop * hsr));
    } else {
        float wir = 4.5;
        if (hsr < 1) {
            if (hsr == 1.0) wir = 4;
            else if (h->pitch_scale < 1.0) wir = 4.5;
            else wir = 6;
            inHop = (size_t)(int)(windowSize / wir);
            outHop = (size_t)(int)(inHop * hsr);
        } else {
            if (hsr == 1.0) wir = 4;
            else wir = 8;
            outHop = (size_t)(int)(windowSize / wir);
            inHop = (size_t)(int)(outHop / hsr);
        }
    }
    h->N = windowSize;
    h->hop = inHop;
    h->hop_out_nominal = (int)outHop;
    h->outbuf_size = (size_t)(hsr > 1 ? windowSize * 16 * hsr : windowSize * 16);

    /* configure, phasevocoderimpl.cc:265-322 */
    h->win = (float *)xcalloc(h->N, sizeof(float));
    pvo_hann((int)h->N, h->win, &h->win_area);
    const int C = cfg->channels;
    const size_t N = h->N, H = N / 2 + 1;
    h->ch = (chan *)xcalloc(C, sizeof(chan));
    for (int c = 0; c < C; ++c) {
        chan *a = &h->ch[c];
        /* channelinfo.cc:26-67: bufferSize = 2*max(window,fft); outbuf = max(outbufSize, bufferSize) */
        size_t bufferSize = 2 * N;
        size_t ob = h->outbuf_size < bufferSize ? bufferSize : h->outbuf_size;
        ring_init(&a->inbuf, (int)bufferSize);
        ring_init(&a->outbuf, (int)ob);
        a->mag = (float *)xcalloc(H, 4); a->phase = (float *)xcalloc(H, 4);
        a->prev_phase = (float *)xcalloc(H, 4); a->prev_outphase = (float *)xcalloc(H, 4);
        a->locked_phase = (float *)xcalloc(H, 4);
        a->oacc = (float *)xcalloc(bufferSize, 4); a->wacc = (float *)xcalloc(bufferSize, 4);
        a->frame_t = (float *)xcalloc(bufferSize, 4); a->frame_f = (float *)xcalloc(bufferSize, 4);
        a->fft = rfft_new((int)N);
        a->res = pvo_res_create();
        a->wacc[0] = 1.f; /* channelinfo.cc:108 */
    }
    h->car = (chan *)xcalloc(C, sizeof(chan));
    h->gen = (rsb *)xcalloc(C, sizeof(rsb));
    h->chord = (rsb *)xcalloc(3 * C, sizeof(rsb));
    for (int c = 0; c < C; ++c) {
        chan *a = &h->car[c];
        size_t bufferSize = 2 * N;
        size_t ob = h->outbuf_size < bufferSize ? bufferSize : h->outbuf_size;
        ring_init(&a->inbuf, (int)bufferSize);
        ring_init(&a->outbuf, (int)ob);
        a->mag = (float *)xcalloc(H, 4); a->phase = (float *)xcalloc(H, 4);
        a->prev_phase = (float *)xcalloc(H, 4); a->prev_outphase = (float *)xcalloc(H, 4);
        a->locked_phase = (float *)xcalloc(H, 4);
        a->oacc = (float *)xcalloc(bufferSize, 4); a->wacc = (float *)xcalloc(bufferSize, 4);
        a->frame_t = (float *)xcalloc(bufferSize, 4); a->frame_f = (float *)xcalloc(bufferSize, 4);
        a->fft = rfft_new((int)N);
        a->res = pvo_res_create();
        a->wacc[0] = 1.f;
        /* phasevocoderimpl.cc:312-320: 440 Hz pulse; A-minor chord on A4 */
        rsb_init(&h->gen[c], (float)cfg->sample_rate, 440, 0.01, 0.06);
        const float chordmin[3] = {440, 523.251, 659.255};
        for (int v = 0; v < 3; ++v) rsb_init(&h->chord[3 * c + v], (float)cfg->sample_rate, chordmin[v], 0.01, 0.06);
    }
    {
        size_t rbs = lrintf(ceil((h->hop * h->time_ratio * 2) / h->pitch_scale));
        if (rbs < h->hop * 16) rbs = h->hop * 16;
        h->resamplebuf_size = rbs;
        h->resamplebuf = (float *)xcalloc(rbs + 64, 4);
    }
    h->firstentry = 1;
    /* whisperSlice draws from libc rand() (phasevocoderprocess.cc:820), never seeded by the reference: one
     * oracle instance == one fresh process, so restart the default sequence */
    if (h->opt_whisper) srand(1);
    h->peak = (int *)xcalloc(H, sizeof(int));
    h->prev_peak = (int *)xcalloc(H, sizeof(int));
    return h;
}

void pvo_destroy(pvo *h) {
    if (!h) return;
    free(h->cep);
    free(h->cenv);
    for (int c = 0; c < h->cfg.channels; ++c) {
        chan *a = &h->ch[c];
        ring_free(&a->inbuf); ring_free(&a->outbuf);
        free(a->mag); free(a->phase); free(a->prev_phase); free(a->prev_outphase); free(a->locked_phase);
        free(a->oacc); free(a->wacc); free(a->frame_t); free(a->frame_f);
        rfft_free(a->fft);
        pvo_res_destroy(a->res);
    }
    for (int c = 0; c < h->cfg.channels; ++c) {
        chan *a = &h->car[c];
        ring_free(&a->inbuf); ring_free(&a->outbuf);
        free(a->mag); free(a->phase); free(a->prev_phase); free(a->prev_outphase); free(a->locked_phase);
        free(a->oacc); free(a->wacc); free(a->frame_t); free(a->frame_f);
        rfft_free(a->fft);
        pvo_res_destroy(a->res);
    }
    free(h->car); free(h->gen); free(h->chord);
    free(h->ch); free(h->win); free(h->peak); free(h->prev_peak); free(h->resamplebuf);
    free(h->rec_shift); free(h->rec_phase);
    free(h);
}

/* phasevocoderprocess.cc:379-410 (function statics -> per-instance) */
static int this_increment(pvo *h, float ratio, size_t increment, size_t samplerate) {
    h->recovery = h->divergence / ((samplerate / 10.0) / increment);
    int incr = lrint(increment * ratio - h->recovery);
    if (incr < lrint((increment * ratio) / 2)) {
        incr = lrint((increment * ratio) / 2);
    } else if (incr > lrint(increment * ratio * 2)) {
        incr = lrint(increment * ratio * 2);
    }
    float divdiff = (increment * ratio) - incr;
    float prevDivergence = h->divergence;
    h->divergence -= divdiff;
    if ((prevDivergence < 0 && h->divergence > 0) || (prevDivergence > 0 && h->divergence < 0)) {
        h->recovery = h->divergence / ((samplerate / 10.0) / increment);
    }
    return incr;
}

/* phasevocoderprocess.cc:492-503 + phasevocoderimpl.h:167-181 */
static void analyze(pvo *h, chan *a) {
    const int N = (int)h->N, hs = N / 2;
    for (int i = 0; i < N; ++i) a->frame_t[i] *= h->win[i];
    for (int i = 0; i < hs; ++i) a->frame_f[i] = a->frame_t[i + hs];
    for (int i = 0; i < hs; ++i) a->frame_f[i + hs] = a->frame_t[i];
    rfft_forward_polar(a->fft, a->frame_f, a->mag, a->phase);
}

/* the per-bin propagation shared by modifySliceSimple (:732-748) and the no-peaks branch of
 * modifySlicePhaseLocked (:620-636) */
static void propagate_bins(pvo *h, chan *a, size_t phaseIncrement) {
    const int hs = (int)h->N / 2;
    for (int i = 0; i < hs; i++) {
        float omega = (2 * M_PI * h->hop * i) / (h->N);
        float delta_phi = omega + pvo_princarg(a->phase[i] - a->prev_phase[i] - omega);
        float advance = delta_phi * phaseIncrement / h->hop;
        float outphase = pvo_princarg(a->prev_outphase[i] + advance);
        a->prev_phase[i] = a->phase[i];
        a->phase[i] = outphase;
        a->prev_outphase[i] = outphase;
    }
}

static void init_bins(pvo *h, chan *a) {
    const int hs = (int)h->N / 2;
    for (int i = 0; i < hs; i++) {
        float tp = a->phase[i];
        a->prev_phase[i] = tp;
        a->prev_outphase[i] = tp;
    }
}

/* phasevocoderprocess.cc:708-753 */
static void modify_simple(pvo *h, chan *a, size_t phaseIncrement) {
    if (h->firstentry) init_bins(h, a);
    else if (h->npeak == 0 || h->nprev_peak == 0) propagate_bins(h, a, phaseIncrement);
    h->firstentry = 0;
}

/* phasevocoderprocess.cc:574-706 */
static void modify_locked(pvo *h, chan *a, size_t phaseIncrement) {
    const int hs = (int)h->N / 2;
    const float *mag = a->mag;
    h->npeak = 0;
    int b = 2;
    while (b + 2 < hs) {
        if (mag[b] > mag[b - 1] && mag[b] > mag[b - 2] && mag[b] > mag[b + 1] && mag[b] > mag[b + 2]) {
            h->peak[h->npeak++] = b;
            b += 3;
        } else {
            b += 1;
        }
    }
    if (h->firstentry) {
        init_bins(h, a);
    } else if (h->npeak == 0 || h->nprev_peak == 0) {
        propagate_bins(h, a, phaseIncrement);
    } else {
        size_t prev_p = 0;
        for (size_t p = 0; p < (size_t)h->npeak; p++) {
            int p2 = h->peak[p];
            while (prev_p < (size_t)h->nprev_peak - 1) {
                if (abs(p2 - h->prev_peak[prev_p + 1]) < abs(p2 - h->prev_peak[prev_p])) prev_p += 1;
                else break;
            }
            int p1 = h->prev_peak[prev_p];
            float avg_p = (p1 + p2) * 0.5;
            float pomega = (2 * M_PI * h->hop * (avg_p - 1)) / (h->N);
            float peak_delta_phi = pomega + pvo_princarg(a->phase[p2] - a->prev_phase[p1] - pomega);
            float peak_target_phase = pvo_princarg(a->prev_outphase[p1] + (peak_delta_phi * phaseIncrement) / h->hop);
            float rot = pvo_princarg(peak_target_phase - a->phase[p2]);
            int bin1 = 0, bin2 = 0;
            if (h->npeak == 1) {
                bin1 = 0; bin2 = hs;
            } else if (p == 0) {
                bin1 = 0; bin2 = round((h->peak[p + 1] + p2) * 0.5);
            } else if (p == (size_t)h->npeak - 1) {
                bin1 = round((h->peak[p - 1] + p2) * 0.5); bin2 = hs;
            } else {
                bin1 = round((h->peak[p - 1] + p2) * 0.5);
                bin2 = round((h->peak[p + 1] + p2) * 0.5);
            }
            for (int i = bin1; i < bin2; i++) a->locked_phase[i] = pvo_princarg(a->phase[i] + rot);
        }
        for (int i = 0; i < hs; i++) {
            a->prev_phase[i] = a->phase[i];
            a->prev_outphase[i] = a->locked_phase[i];
            a->phase[i] = a->locked_phase[i];
        }
    }
    memcpy(h->prev_peak, h->peak, sizeof(int) * h->npeak);
    h->nprev_peak = h->npeak;
    h->firstentry = 0;
}

/* phasevocoderprocess.cc:558-572 */
static void modify_intratio(pvo *h, chan *a, size_t phaseIncrement) {
    const int hs = (int)h->N / 2;
    for (int i = 0; i < hs; i++) a->phase[i] = a->phase[i] * phaseIncrement / h->hop;
}

/* phasevocoderprocess.cc:842-923 */
static void freq_comp(pvo *h, chan *a, float freq_comp) {
    float *phi = a->phase, *mag = a->mag;
    const int hs = (int)h->N / 2;
    float absps = h->pitch_scale > 1 ? h->pitch_scale : 1 / h->pitch_scale;
    const float fixedgain = absps;
    if (freq_comp > 1.0) {
        for (int target = 0; target <= hs; ++target) {
            int source = lrint(target * freq_comp);
            float delta_omega = (2 * M_PI * h->hop * (target - source)) / (h->N);
            if (source > hs) {
                mag[target] = 0.0;
                phi[target] = 0.0;
            } else {
                mag[target] = mag[source];
                phi[target] = phi[source] + delta_omega;
            }
        }
    } else {
        for (int target = hs; target > 0;) {
            --target;
            int source = lrint(target * freq_comp);
            float delta_omega = (2 * M_PI * h->hop * (target - source)) / (h->N);
            mag[target] = mag[source];
            phi[target] = phi[source] + delta_omega;
        }
    }
    for (int i = 0; i < hs + 1; ++i) mag[i] *= fixedgain;
}

/* Cepstral formant shift: phasevocoderprocess.cc:925-999 (formantShiftSlice) with FFT.cc:2723-2733
 * (D_KISSFFT::inverseCepstral) and :2606-2610 (forward).  Dead code upstream (every call site is commented out,
 * :826,832,838,1017,1021); restated because it is what "formant-envelope scaling" in the north star describes
 *  and pinned on the real function through oracle/ref_formant.cc.
 *   cep = irfft(log(mag + 1e-6)), lifter to the first 60 quefrencies (ends halved), envelope = exp(Re rfft(cep/N)),
 *   mag = mag / envelope * envelope[lrint(k * env_comp)] */
static void formant_shift(rfft *p, float *mag, float env_comp, float *cep, float *envelope) {
    const int N = p->n, hs = N / 2;
    const float factor = 1.0 / N;
    for (int i = 0; i <= hs; ++i) {
        p->packed[i].r = logf(mag[i] + 0.000001f);
        p->packed[i].i = 0.0f;
    }
    rfft_inverse(p, p->packed, cep);
    const int cutoff = 60;
    cep[0] /= 2;
    cep[cutoff - 1] /= 2;
    for (int i = cutoff; i < N; ++i) cep[i] = 0.0;
    for (int i = 0; i < cutoff; ++i) cep[i] *= factor;
    rfft_forward(p, cep, p->packed);
    for (int i = 0; i <= hs; ++i) envelope[i] = p->packed[i].r;
    for (int i = 0; i <= hs; ++i) envelope[i] = expf(envelope[i]);
    for (int i = 0; i <= hs; ++i) mag[i] /= envelope[i];
    if (env_comp > 1.0) {
        for (int target = 0; target <= hs; ++target) {
            int source = lrint(target * env_comp);
            if (source > hs) envelope[target] = 0.0;
            else envelope[target] = envelope[source];
        }
    } else {
        for (int target = hs; target > 0;) {
            --target;
            int source = lrint(target * env_comp);
            envelope[target] = envelope[source];
        }
    }
    for (int i = 0; i <= hs; ++i) mag[i] *= envelope[i];
}

void pvo_formant_shift(int n, float *mag, float env_comp) {
    rfft *p = rfft_new(n);
    float *cep = (float *)xcalloc(n, sizeof(float)), *env = (float *)xcalloc(n / 2 + 1, sizeof(float));
    formant_shift(p, mag, env_comp, cep, env);
    free(cep);
    free(env);
    rfft_free(p);
}

/* phasevocoderprocess.cc:1001-1075 */
static void synthesise(pvo *h, chan *a) {
    const int N = (int)h->N, hs = N / 2;
    if (h->opt_formant && (h->pitch_scale != 1.0)) freq_comp(h, a, h->pitch_scale);
    if (h->opt_cepstral && (h->pitch_scale != 1.0)) { /* formantPreserveSlice's commented-out alternative (:838) */
        if (!h->cep) {
            h->cep = (float *)xcalloc(N, sizeof(float));
            h->cenv = (float *)xcalloc(hs + 1, sizeof(float));
        }
        formant_shift(a->fft, a->mag, h->pitch_scale, h->cep, h->cenv);
    }
    if (h->opt_gender && (h->pitch_scale != 1.0)) {
        if (h->pitch_scale > 1) freq_comp(h, a, 0.85 * h->pitch_scale);
        else freq_comp(h, a, 1.17 * h->pitch_scale);
    } else if (h->opt_gender) {
        freq_comp(h, a, 0.8);
    }
    float factor = 1.f / h->N;
    for (int i = 0; i < hs + 1; ++i) a->mag[i] *= factor;
    rfft_inverse_polar(a->fft, a->mag, a->phase, a->frame_f);
    for (int i = 0; i < hs; ++i) a->frame_t[i] = a->frame_f[i + hs];
    for (int i = 0; i < hs; ++i) a->frame_t[i + hs] = a->frame_f[i];
    for (int i = 0; i < N; ++i) a->frame_t[i] *= h->win[i];
    for (int i = 0; i < N; ++i) a->oacc[i] += a->frame_t[i];
    float gain = h->win_area * 1.5;
    for (int i = 0; i < N; ++i) a->wacc[i] += h->win[i] * gain;
}

/* phasevocoderprocess.cc:1140-1194 */
static int write_slice(pvo *h, chan *a, size_t shiftIncrement) {
    const int N = (int)h->N;
    const int s = (int)shiftIncrement;
    if (s > N) {
        h->undefined = 1;
        return 0;
    }
    for (int i = 0; i < s; ++i) a->oacc[i] /= a->wacc[i];
    int outframes;
    if (h->pitch_scale != 1.0) {
        size_t req = (int)(ceil(shiftIncrement / h->pitch_scale));
        if (req > h->resamplebuf_size) {
            free(h->resamplebuf);
            h->resamplebuf = (float *)xcalloc(req + 64, 4);
            h->resamplebuf_size = req;
        }
        outframes = pvo_res_process(a->res, a->oacc, s, 1.0 / h->pitch_scale, h->resamplebuf);
        ring_write(&a->outbuf, h->resamplebuf, outframes);
    } else {
        outframes = s;
        ring_write(&a->outbuf, a->oacc, s);
    }
    memmove(a->oacc, a->oacc + s, sizeof(float) * (N - s));
    memset(a->oacc + N - s, 0, sizeof(float) * s);
    memmove(a->wacc, a->wacc + s, sizeof(float) * (N - s));
    memset(a->wacc + N - s, 0, sizeof(float) * s);
    return outframes;
}

static void record_incr(pvo *h, size_t shift, size_t phase) {
    if (h->nrec == h->caprec) {
        h->caprec = h->caprec ? h->caprec * 2 : 1024;
        h->rec_shift = (int *)realloc(h->rec_shift, sizeof(int) * h->caprec);
        h->rec_phase = (int *)realloc(h->rec_phase, sizeof(int) * h->caprec);
    }
    h->rec_shift[h->nrec] = (int)shift;
    h->rec_phase[h->nrec] = (int)phase;
    h->nrec++;
}

/* phasevocoderprocess.cc:236-287 + :305-376 + :412-489 */
static int process_one_slice(pvo *h) {
    const int C = h->cfg.channels;
    const int N = (int)h->N;
    for (int c = 0; c < C; ++c) {
        chan *a = &h->ch[c];
        if (ring_readspace(&a->inbuf) < N) return -1;
        int ready = ring_readspace(&a->inbuf);
        ring_peek(&a->inbuf, a->frame_t, ready < N ? ready : N);
        ring_discard(&a->inbuf, (int)h->hop);
        analyze(h, a);
    }
    size_t phaseIncrement, shiftIncrement;
    if (h->opt_robotic || h->opt_whisper) {
        phaseIncrement = h->hop;
        shiftIncrement = h->hop;
    } else if (is_int_ratio(h)) {
        phaseIncrement = h->hop * hs_ratio(h);
        shiftIncrement = h->hop * hs_ratio(h);
    } else {
        int incr = this_increment(h, hs_ratio(h), h->hop, (size_t)h->cfg.sample_rate);
        shiftIncrement = incr;
        chan *a0 = &h->ch[0];
        if (a0->prev_increment == 0) phaseIncrement = shiftIncrement;
        else phaseIncrement = a0->prev_increment;
        a0->prev_increment = shiftIncrement;
    }
    record_incr(h, shiftIncrement, phaseIncrement);

    int outframes = 0;
    for (int c = 0; c < C; ++c) {
        chan *a = &h->ch[c];
        if (h->opt_robotic) {
            for (int i = 0; i < N / 2 + 1; ++i) a->phase[i] = 0;
        } else if (h->opt_whisper) {
            float two_pi = 2 * M_PI;
            for (int i = 0; i < N / 2 + 1; ++i) a->phase[i] = two_pi * (float)rand() / (float)RAND_MAX;
        } else if (h->cfg.coremode == 1) {
            modify_locked(h, a, phaseIncrement);
        } else if (h->cfg.coremode == 2) {
            modify_intratio(h, a, phaseIncrement);
        } else {
            modify_simple(h, a, phaseIncrement);
        }
        synthesise(h, a);
        int required = (int)(shiftIncrement / h->pitch_scale) + 1;
        int ws = ring_writespace(&a->outbuf);
        if (ws < required) {
            fprintf(stderr, "pv_oracle: Buffer overrun on output for channel %d\n", c);
            outframes = 0;
        } else {
            outframes = write_slice(h, a, shiftIncrement);
        }
        a->slicecnt++;
    }
    return outframes;
}

/* phasevocoderprocess.cc:122-156: CONSTANT mode -- no phase modification, in hop == out hop */
static int process_one_slice_constant(pvo *h) {
    const int C = h->cfg.channels;
    const int N = (int)h->N;
    for (int c = 0; c < C; ++c) {
        chan *a = &h->ch[c];
        if (ring_readspace(&a->inbuf) < N) return -1;
        int ready = ring_readspace(&a->inbuf);
        ring_peek(&a->inbuf, a->frame_t, ready < N ? ready : N);
        ring_discard(&a->inbuf, (int)h->hop);
        analyze(h, a);
    }
    record_incr(h, h->hop, h->hop);
    int outframes = 0;
    for (int c = 0; c < C; ++c) {
        chan *a = &h->ch[c];
        synthesise(h, a);
        size_t ws = ring_writespace(&a->outbuf);
        if (ws < h->hop) {
            fprintf(stderr, "pv_oracle: Buffer overrun on output for channel %d\n", c);
            return 0;
        }
        outframes = write_slice(h, a, h->hop);
        a->slicecnt++;
    }
    return outframes;
}

static int is_vocoder(const pvo *h) {
    return h->cfg.mode == PVO_VOCODER_ROSENBERG || h->cfg.mode == PVO_VOCODER_CHORD;
}

/* phasevocoderprocess.cc:158-195 processOneSliceVocoder, :755-776 modifySliceVocoder,
 * :1077-1107 synthesiseSliceCarrier, :1196-1231 writeSliceCarrier */
static int process_one_slice_vocoder(pvo *h) {
    const int C = h->cfg.channels;
    const int N = (int)h->N, hs = N / 2;
    for (int c = 0; c < C; ++c) {
        chan *a = &h->ch[c];
        if (ring_readspace(&a->inbuf) < N) return -1;
        int ready = ring_readspace(&a->inbuf);
        ring_peek(&a->inbuf, a->frame_t, ready < N ? ready : N);
        ring_discard(&a->inbuf, (int)h->hop);
        analyze(h, a);
    }
    for (int c = 0; c < C; ++c) {
        chan *a = &h->car[c];
        int ready = ring_readspace(&a->inbuf);
        ring_peek(&a->inbuf, a->frame_t, ready < N ? ready : N);
        ring_discard(&a->inbuf, (int)h->hop);
        analyze(h, a);
    }
    record_incr(h, h->hop, h->hop);
    for (int c = 0; c < C; ++c) {
        chan *ad = &h->ch[c], *ca = &h->car[c];
        /* modifySliceVocoder: per band, carrier magnitude *= mean modulator magnitude */
        int num_bands = 512;
        int band_len = (int)floor((float)(h->N) / (float)(num_bands * 2));
        for (int band_no = 0; band_no < num_bands; band_no++) {
            float mean_modul_mag = 0;
            for (int i = 0, j = band_no * band_len; i < band_len; i++, j++) mean_modul_mag += ad->mag[j];
            mean_modul_mag /= (band_len * 2);
            for (int i = 0, j = band_no * band_len; i < band_len; i++, j++) ca->mag[j] *= mean_modul_mag;
            ca->mag[0] = 0;
            ca->mag[hs] = 0;
        }
        /* synthesiseSliceCarrier */
        float factor = 1.f / h->N;
        for (int i = 0; i < hs + 1; ++i) ca->mag[i] *= factor;
        rfft_inverse_polar(ca->fft, ca->mag, ca->phase, ca->frame_f);
        for (int i = 0; i < hs; ++i) ca->frame_t[i] = ca->frame_f[i + hs];
        for (int i = 0; i < hs; ++i) ca->frame_t[i + hs] = ca->frame_f[i];
        for (int i = 0; i < N; ++i) ca->frame_t[i] *= h->win[i];
        for (int i = 0; i < N; ++i) ca->oacc[i] += ca->frame_t[i];
        float gain = h->win_area * 1.5;
        for (int i = 0; i < N; ++i) ca->wacc[i] += h->win[i] * gain;
        /* writeSliceCarrier */
        const int s = (int)h->hop;
        if (ring_writespace(&ca->outbuf) < s) {
            /* writeSliceCarrier returns 0 (:1205-1212) and processOneSliceVocoder ignores it (:186-191): the
             * frame stays in the accumulators, the loop goes on to the next channel */
            fprintf(stderr, "pv_oracle: Buffer overrun on output for channel\n");
        } else {
            for (int i = 0; i < s; ++i) ca->oacc[i] /= ca->wacc[i];
            ring_write(&ca->outbuf, ca->oacc, s);
            memmove(ca->oacc, ca->oacc + s, sizeof(float) * (N - s));
            memset(ca->oacc + N - s, 0, sizeof(float) * s);
            memmove(ca->wacc, ca->wacc + s, sizeof(float) * (N - s));
            memset(ca->wacc + N - s, 0, sizeof(float) * s);
        }
        ad->slicecnt++;
        ca->slicecnt++;
    }
    return 0;
}

int pvo_available(const pvo *h) {
    int ret = 0;
    const chan *set = is_vocoder(h) ? h->car : h->ch;
    for (int c = 0; c < h->cfg.channels; ++c) {
        int a = ring_readspace(&set[c].outbuf);
        if (c == 0 || a < ret) ret = a;
    }
    return ret;
}

/* phasevocoderimpl.cc:340-369 + phasevocoderprocess.cc:43-64 */
int pvo_process(pvo *h, const float *const *in, int n) {
    const int C = h->cfg.channels;
    int allread = 0;
    size_t *nread = (size_t *)xcalloc(C, sizeof(size_t));
    while (!allread) {
        for (int c = 0; c < C; ++c) {
            size_t remaining = (size_t)n - nread[c];
            size_t writable = ring_writespace(&h->ch[c].inbuf);
            size_t towrite = remaining < writable ? remaining : writable;
            if (is_vocoder(h)) {
                /* enbufferChannelVocoder (phasevocoderprocess.cc:66-120): the carrier is generated in
                 * lock-step with the samples written */
                size_t cw = ring_writespace(&h->car[c].inbuf);
                if (cw < towrite) towrite = cw;
                float *ci = (float *)xcalloc(towrite ? towrite : 1, sizeof(float));
                for (size_t i = 0; i < towrite; ++i) {
                    if (h->cfg.mode == PVO_VOCODER_ROSENBERG) {
                        ci[i] = rsb_next(&h->gen[c]) * 0.3;
                    } else {
                        float res = 0;
                        for (int v = 0; v < 3; ++v) res += rsb_next(&h->chord[3 * c + v]) / 3;
                        ci[i] = res * 0.3;
                    }
                }
                ring_write(&h->car[c].inbuf, ci, (int)towrite);
                free(ci);
            }
            ring_write(&h->ch[c].inbuf, in[c] + nread[c], (int)towrite);
            nread[c] += towrite;
            allread = !(nread[c] < (size_t)n);
        }
        if (h->cfg.mode == PVO_CONSTANT) process_one_slice_constant(h);
        else if (is_vocoder(h)) process_one_slice_vocoder(h);
        else process_one_slice(h);
    }
    free(nread);
    if (h->undefined) return PVO_UNDEFINED;
    return pvo_available(h);
}

int pvo_retrieve(pvo *h, float *const *out, int n) {
    int ret = n;
    chan *set = is_vocoder(h) ? h->car : h->ch;
    for (int c = 0; c < h->cfg.channels; ++c) {
        int got = ring_read(&set[c].outbuf, out[c], ret);
        if (got < ret) ret = got;
    }
    return ret;
}

void pvo_get_info(const pvo *h, pvo_info *o) {
    memset(o, 0, sizeof(*o));
    o->fftsize = (int)h->N;
    o->hop_in = (int)h->hop;
    o->hop_out_nominal = h->hop_out_nominal;
    o->outbuf_capacity = h->ch[0].outbuf.size - 1;
    o->pitch_scale = h->pitch_scale;
    o->hs_ratio = hs_ratio(h);
    o->int_ratio = is_int_ratio(h);
    o->resample = h->pitch_scale != 1.0;
    const pvo_resampler *r = h->ch[0].res;
    if (r->lastratio > 0) {
        o->res_num = r->num_rate; o->res_den = r->den_rate;
        o->res_filt_len = (int)r->filt_len; o->res_oversample = (int)r->oversample; o->res_interp = r->interp;
    }
    o->slices = h->ch[0].slicecnt;
}

long pvo_get_increments(const pvo *h, int *shift, int *phase, long max) {
    long n = h->nrec < max ? h->nrec : max;
    if (shift) memcpy(shift, h->rec_shift, sizeof(int) * n);
    if (phase) memcpy(phase, h->rec_phase, sizeof(int) * n);
    return h->nrec;
}

/* libm's atan2f over arrays (the analysis phases of the reference come from it, FFT.cc:2623-2630): what the
 * device's restatement is compared with on the GPU box */
void pvo_atan2f_array(const float *y, const float *x, float *out, long n) {
    for (long i = 0; i < n; ++i) out[i] = atan2f(y[i], x[i]);
}
