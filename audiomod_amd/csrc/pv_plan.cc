// pv_plan.cc -- see pv_plan.h.  Host-only; compiled with -ffp-contract=off so the float/double
// expression shapes below evaluate exactly as written (they are part of the specification:
// the reference computes these on x86-64 without FMA).
#include "pv_plan.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace pv {

// ------------------------------------------------------------------------------------------
// window  (reference dsp/windowfunc.h:159-169: cosine-sum evaluated in double, stored float;
// area = float running sum / n, :152-156)
// ------------------------------------------------------------------------------------------
static void make_hann(int n, std::vector<float> &w, float &area) {
    w.resize(n);
    const float a0 = 0.50f, a1 = 0.50f, a2 = 0.0f, a3 = 0.0f;
    for (int i = 0; i < n; ++i) {
        float v = 1.0f;
        double e = (a0 - a1 * std::cos(2 * M_PI * i / n) + a2 * std::cos(4 * M_PI * i / n) -
                    a3 * std::cos(6 * M_PI * i / n));
        v = (float)(v * e);
        w[i] = v;
    }
    float acc = 0;
    for (int i = 0; i < n; ++i) acc += w[i];
    acc /= n;
    area = acc;
}

// ------------------------------------------------------------------------------------------
// FFT plan  (reference kissfft/kiss_fft.c: factor order 4s then 2s :292-316; recursion leaf
// copy :264-268 gives the input permutation; twiddles in double then float :341-347;
// kiss_fftr.c:57-63 super twiddles)
// ------------------------------------------------------------------------------------------
static void perm_rec(const int *radix, const int *m, const int *fs, int lev, int out_base, int in_base,
                     std::vector<int32_t> &perm) {
    if (m[lev] == 1) {
        for (int q = 0; q < radix[lev]; ++q) perm[out_base + q] = in_base + q * fs[lev];
    } else {
        for (int q = 0; q < radix[lev]; ++q)
            perm_rec(radix, m, fs, lev + 1, out_base + q * m[lev], in_base + q * fs[lev], perm);
    }
}

static int make_fft(int nc, FftPlan &p) {
    p.nc = nc;
    int radix[kMaxStages], mm[kMaxStages], fs[kMaxStages];
    int rem = nc, lev = 0, stride = 1;
    while (rem > 1) {
        int r = (rem % 4 == 0) ? 4 : 2; // powers of two only: 4s first, a final 2 if log2 is odd
        if (rem % r) return PV_ERR_INVALID_ARG;
        rem /= r;
        if (lev >= kMaxStages) return PV_ERR_INVALID_ARG;
        radix[lev] = r;
        mm[lev] = rem;
        fs[lev] = stride;
        stride *= r;
        ++lev;
    }
    p.nstages = lev;
    p.perm.assign(nc, 0);
    perm_rec(radix, mm, fs, 0, 0, 0, p.perm);
    for (int s = 0; s < lev; ++s) { // execution order: deepest level first
        p.radix[s] = radix[lev - 1 - s];
        p.m[s] = mm[lev - 1 - s];
        p.fstride[s] = fs[lev - 1 - s];
    }
    p.tw_fwd.resize(nc);
    p.tw_inv.resize(nc);
    p.st_fwd.resize(nc);
    p.st_inv.resize(nc);
    for (int i = 0; i < nc; ++i) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / nc;
        p.tw_fwd[i].r = (float)std::cos(phase);
        p.tw_fwd[i].i = (float)std::sin(phase);
        phase *= -1;
        p.tw_inv[i].r = (float)std::cos(phase);
        p.tw_inv[i].i = (float)std::sin(phase);
        double sp = -3.14159265358979323846264338327 * ((double)i / nc + .5);
        p.st_fwd[i].r = (float)std::cos(sp);
        p.st_fwd[i].i = (float)std::sin(sp);
        sp *= -1;
        p.st_inv[i].r = (float)std::cos(sp);
        p.st_inv[i].i = (float)std::sin(sp);
    }
    return PV_OK;
}

// ------------------------------------------------------------------------------------------
// Speex quality-4 filter design  (reference speex/resample.c: Q4 row of the quality table :290,
// KAISER8 :233-240 with oversample 32 :262, compute_func :300-322, sinc :325-337,
// update_filter :661-775)
// ------------------------------------------------------------------------------------------
// (the window table is data; it is what defines the filter)
static const double kKaiser8[36] = {
    0.99635258, 1.00000000, 0.99635258, 0.98548012, 0.96759014, 0.94302200, 0.91223751, 0.87580811, 0.83439927,
    0.78875245, 0.73966538, 0.68797126, 0.63451750, 0.58014482, 0.52566725, 0.47185369, 0.41941150, 0.36897272,
    0.32108304, 0.27619388, 0.23465776, 0.19672670, 0.16255380, 0.13219758, 0.10562887, 0.08273982, 0.06335451,
    0.04724088, 0.03412321, 0.02369490, 0.01563093, 0.00959968, 0.00527363, 0.00233883, 0.00050000, 0.00000000};

// Arithmetic contract (resample.c:300-322): the window value at u in [0, 1] is a 4-point cubic through the table
// at spacing 1/32, with these weights -- float position and fraction, double weights, this operand order.
static double kaiser8_window(float u) {
    const float pos = u * 32;
    const int cell = (int)std::floor(pos);
    const float f = pos - cell;
    const float f2 = f * f, f3 = f * f * f;
    const double w_hi = -0.1666666667 * f + 0.1666666667 * f3;
    const double w_mid = f + 0.5 * f2 - 0.5 * f3;
    const double w_lo = -0.3333333333 * f + 0.5 * f2 - 0.1666666667 * f3;
    const double w_rest = 1.f - w_hi - w_mid - w_lo;
    const double *t = kKaiser8 + cell;
    return w_lo * t[0] + w_rest * t[1] + w_mid * t[2] + w_hi * t[3];
}

// One tap of the windowed sinc (resample.c:325-337): x in input samples from the centre, taps = filter length.
// Arithmetic contract: float product cutoff * x, double sine and quotient, window at float |2x / taps|.
static float filter_tap(float cutoff, float x, int taps) {
    if (std::fabs(x) < 1e-6) return cutoff;
    if (std::fabs(x) > .5 * taps) return 0;
    const float arg = x * cutoff;
    return (float)(cutoff * std::sin(M_PI * arg) / (M_PI * arg) * kaiser8_window((float)std::fabs(2. * x / taps)));
}

static uint32_t gcd_u32(uint32_t a, uint32_t b) {
    while (b) {
        const uint32_t r = a % b;
        a = b;
        b = r;
    }
    return a;
}

// The stream rate pair of the resampler that writeSlice feeds (phasevocoderprocess.cc:1174 passes 1 / pitchScale):
// the ratio becomes a fraction over 272408136 with the other term truncated from a double product / quotient
// (dsp/resampler.cc:746-757), handed over as (input rate, output rate) and reduced (resample.c:1136-1139).
static void stream_rates(float out_per_in, uint32_t &num_rate, uint32_t &den_rate) {
    const uint32_t unit = 272408136U;
    num_rate = den_rate = 1;
    if (out_per_in < 1.f) {
        num_rate = unit;
        den_rate = (uint32_t)(double(unit) * double(out_per_in));
    } else if (out_per_in > 1.f) {
        num_rate = (uint32_t)(double(unit) / double(out_per_in));
        den_rate = unit;
    }
    const uint32_t g = gcd_u32(num_rate, den_rate);
    num_rate /= g;
    den_rate /= g;
}

// Quality 4 of the Speex table: 64 taps, 8x oversampling, Kaiser-8 window, bandwidth 0.921 when down-sampling and
// 0.940 when up-sampling (resample.c:290); down-sampling stretches the filter and narrows the cutoff by the rate
// ratio and coarsens the oversampling for large ratios (:671-697).
static void make_resampler(Derived &d) {
    d.res_ratio = (float)(1.0 / d.pitch_scale);
    stream_rates(d.res_ratio, d.res_num, d.res_den);
    const uint32_t num_rate = d.res_num, den_rate = d.res_den;
    uint32_t taps = 64, oversample = 8;
    float cutoff = 0.940f;
    if (num_rate > den_rate) {
        cutoff = 0.921f * den_rate / num_rate;
        taps = (uint32_t)std::ceil(taps * ((double)num_rate / (double)den_rate)) & ~0x3u;
        // (four independent 32-bit tests, as upstream: 16 * den_rate wraps for rates just below 272408136)
        for (uint32_t fold : {2u, 4u, 8u, 16u})
            if (fold * den_rate < num_rate) oversample >>= 1;
        if (oversample < 1) oversample = 1;
    }
    d.filt_len = (int)taps;
    d.oversample = (int)oversample;
    d.interp = den_rate > oversample; // few distinct phases: one exact filter per phase instead of interpolation
    const int half = (int)taps / 2;
    if (!d.interp) {
        d.sinc.assign((size_t)taps * den_rate, 0.f);
        for (uint32_t phase = 0; phase < den_rate; ++phase)
            for (int j = 0; j < (int)taps; ++j)
                d.sinc[phase * taps + j] = filter_tap(cutoff, (j - half + 1) - ((float)phase) / den_rate, (int)taps);
    } else {
        const int n = (int)(oversample * taps);
        d.sinc.assign((size_t)n + 8, 0.f);
        for (int i = -4; i < n + 4; ++i) d.sinc[(size_t)(i + 4)] = filter_tap(cutoff, i / (float)oversample - half, (int)taps);
    }
    d.int_adv = (int)(num_rate / den_rate);
    d.frac_adv = (int)(num_rate % den_rate);
}

static size_t next_pow2(size_t v) {
    if (!(v & (v - 1))) return v;
    int bits = 0;
    while (v) {
        ++bits;
        v >>= 1;
    }
    return (size_t)1 << bits;
}

int derive(const pv_config &cfg, Derived &d) {
    d = Derived();
    d.cfg = cfg;
    if (cfg.channels < 1 || cfg.channels > 64 || cfg.sample_rate < 1 || cfg.fftsize < 1 || cfg.hopsize < 0)
        return PV_ERR_INVALID_ARG;
    switch (cfg.mode) {
    case PV_MODE_NORMAL_SHIFT:
    case PV_MODE_GENDER_CHANGE:
    case PV_MODE_FORMANT_PRESERVE:
    case PV_MODE_NORMAL_STRETCH:
    case PV_MODE_ROBOTIC:
    case PV_MODE_WHISPER:
    case PV_MODE_CONSTANT:
    case PV_MODE_VOCODER_ROSENBERG:
    case PV_MODE_VOCODER_CHORD:
    case PV_MODE_FORMANT_CEPSTRAL:
        break;
    default:
        return PV_ERR_UNSUPPORTED;
    }
    // phasevocoder.cc:25-26: float semis/12, double pow, stored float
    d.time_ratio = cfg.time_ratio;
    d.pitch_scale = cfg.pitch_semitones != 0 ? (float)std::pow(2.0, cfg.pitch_semitones / 12) : 1.0f;
    d.robotic = cfg.mode == PV_MODE_ROBOTIC;
    d.whisper = cfg.mode == PV_MODE_WHISPER;
    d.constant = cfg.mode == PV_MODE_CONSTANT;
    d.vocoder = cfg.mode == PV_MODE_VOCODER_ROSENBERG || cfg.mode == PV_MODE_VOCODER_CHORD;
    d.chord = cfg.mode == PV_MODE_VOCODER_CHORD;

    size_t windowSize = next_pow2((size_t)cfg.fftsize);
    if (windowSize < 64 || windowSize > 16384) return PV_ERR_INVALID_ARG;
    if (d.pitch_scale <= 0.0) d.pitch_scale = 1.0;
    if (d.time_ratio <= 0.0) d.time_ratio = 1.0;
    const float hsr = d.time_ratio * d.pitch_scale;
    d.hs_ratio = hsr;
    // Hops (calculateSizes, phasevocoderimpl.cc:196-226).  A caller-chosen input hop fixes the output hop; otherwise
    // the window-to-hop ratio depends on which way the slices move: when they shrink (hsr < 1) the INPUT hop is
    // window / 4.5 (pitch down) or window / 6 and the output hop follows; when they grow or stay, the OUTPUT hop is
    // window / 8 (window / 4 at exactly 1) and the input hop follows.  Arithmetic contract: float quotients and
    // products truncated through int.
    size_t inHop, outHop;
    if (cfg.hopsize > 0) {
        inHop = (size_t)cfg.hopsize;
        outHop = (size_t)int(std::floor(inHop * hsr));
    } else if (hsr < 1) {
        const float window_per_hop = d.pitch_scale < 1.0 ? 4.5f : 6.f;
        inHop = (size_t)int(windowSize / window_per_hop);
        outHop = (size_t)int(inHop * hsr);
    } else {
        const float window_per_hop = hsr == 1.0 ? 4.f : 8.f;
        outHop = (size_t)int(windowSize / window_per_hop);
        inHop = (size_t)int(outHop / hsr);
    }
    if (inHop < 1 || inHop > windowSize) return PV_ERR_INVALID_ARG;
    d.N = (int)windowSize;
    d.hs = d.N / 2;
    d.H = d.hs + 1;
    d.hop = (int)inHop;
    d.hop_out_nominal = (int)outHop;
    {
        size_t ob = (size_t)(hsr > 1 ? windowSize * 16 * hsr : windowSize * 16);
        size_t buf = 2 * windowSize;
        d.outbuf_cap = (int)(ob < buf ? buf : ob);
    }
    {
        float efr = hsr;
        d.int_ratio = std::fabs(efr - std::floor(efr)) <= 0.001; // float abs overload in the reference build
    }
    d.resample = d.pitch_scale != 1.0 && !d.vocoder; // writeSliceCarrier never resamples (:1196-1231)
    d.voc_band_len = (int)std::floor(float(windowSize) / float(512 * 2));
    d.two_pi_hop = 2 * M_PI * (size_t)d.hop;
    d.inv_n = 1.f / (size_t)d.N;

    // freqCompSlice dispatch (phasevocoderprocess.cc:1006-1022, 824-840)
    const bool formant = cfg.mode == PV_MODE_FORMANT_PRESERVE, gender = cfg.mode == PV_MODE_GENDER_CHANGE;
    if (formant && d.pitch_scale != 1.0) {
        d.do_freq_comp = true;
        d.freq_comp = d.pitch_scale;
    }
    if (gender && d.pitch_scale != 1.0) {
        d.do_freq_comp = true;
        d.freq_comp = d.pitch_scale > 1 ? (float)(0.85 * d.pitch_scale) : (float)(1.17 * d.pitch_scale);
    } else if (gender) {
        d.do_freq_comp = true;
        d.freq_comp = (float)0.8;
    }
    d.fixed_gain = d.pitch_scale > 1 ? d.pitch_scale : 1 / d.pitch_scale;
    // extension mode: formantShiftSlice(channel, m_pitchScale) where formantPreserveSlice has it commented out
    if (cfg.mode == PV_MODE_FORMANT_CEPSTRAL) {
        if (d.N < 128) return PV_ERR_UNSUPPORTED; // the lifter keeps 60 quefrencies: needs at least 64 bins
        if (d.pitch_scale != 1.0) {
            d.cepstral = true;
            d.env_comp = d.pitch_scale;
        }
    }

    make_hann(d.N, d.window, d.win_area);
    d.win_gain = (float)(d.win_area * 1.5);
    if (d.resample) make_resampler(d);
    int st = make_fft(d.N / 2, d.fft);
    if (st != PV_OK) return st;

    // smallest shift any slice can take: the clamp lrint(h*ratio/2) (phasevocoderprocess.cc:394-395)
    if (d.robotic || d.whisper || d.constant || d.vocoder) d.min_shift = d.hop;
    else if (d.int_ratio) d.min_shift = (int)(size_t)(d.hop * hsr);
    else d.min_shift = (int)std::lrint(((size_t)d.hop * hsr) / 2);
    if (d.min_shift < 1) return PV_ERR_INVALID_ARG;
    return PV_OK;
}

// ------------------------------------------------------------------------------------------
// Planner
// ------------------------------------------------------------------------------------------
// The shift increment of the next slice (calculateThisIncrement, phasevocoderprocess.cc:379-410).  The ideal
// output hop, hop * ratio, is rarely an integer; the increments are integers that follow it, a running "divergence"
// keeps the accumulated difference and a recovery term pulls the next increment back by the divergence spread over
// a tenth of a second's worth of slices.  Arithmetic contract: `ideal` is a float product; the recovery divides a
// float by a double and is stored as float; lrint (half to even) of float expressions; clamp to [ideal/2, 2 ideal].
int Planner::next_increment() {
    const float ideal = (size_t)d_.hop * d_.hs_ratio;
    const double slices_per_tenth = ((size_t)d_.cfg.sample_rate / 10.0) / (size_t)d_.hop;
    recovery_ = divergence_ / slices_per_tenth;
    const long lowest = std::lrint(ideal / 2), highest = std::lrint(ideal * 2);
    long step = std::lrint(ideal - recovery_);
    if (step < lowest) step = lowest;
    else if (step > highest) step = highest;
    const float before = divergence_;
    divergence_ -= ideal - (int)step;
    const bool crossed_zero = (before < 0 && divergence_ > 0) || (before > 0 && divergence_ < 0);
    if (crossed_zero) recovery_ = divergence_ / slices_per_tenth;
    return (int)step;
}

static thread_local const char *g_plan_reason = "";
const char *plan_reason() { return g_plan_reason; }
void plan_reason_clear() { g_plan_reason = ""; }

int Planner::try_slice(std::vector<SliceRec> &out) {
    if (in_fill_ < d_.N) return PV_OK; // inbufReady false -> processOneSlice returns early
    in_fill_ -= d_.hop;
    size_t phaseInc, shiftInc;
    if (d_.robotic || d_.whisper || d_.constant || d_.vocoder) {
        phaseInc = shiftInc = (size_t)d_.hop; // :267-269 (robotic / whisper); processOneSliceConstant :139-150
    } else if (d_.int_ratio) {
        phaseInc = shiftInc = (size_t)((size_t)d_.hop * d_.hs_ratio);
    } else {
        int incr = next_increment();
        shiftInc = (size_t)incr;
        phaseInc = prev_increment_ == 0 ? shiftInc : (size_t)prev_increment_;
        prev_increment_ = (int64_t)shiftInc;
    }
    if ((int64_t)shiftInc < 1 || (int64_t)shiftInc > d_.N) {
        // writeSlice (phasevocoderprocess.cc:1181-1190) moves N - shiftIncrement floats: beyond N that count wraps
        // and the reference overruns its accumulators -- undefined there, refused here
        g_plan_reason = "a slice's shift increment falls outside [1, fftsize] (hop too large for this ratio): "
                        "undefined behaviour in the reference";
        return PV_ERR_UNSUPPORTED;
    }
    SliceRec r;
    r.shift = (int32_t)shiftInc;
    r.phase_inc = (int32_t)phaseInc;
    r.P = P_;
    r.K0 = K_;
    r.adv = (int32_t)shiftInc;
    r.flags = 0;
    // output-ring guard (phasevocoderprocess.cc:337-364; CONSTANT :139-150; vocoder :1203-1212).  All channels hold
    // the same amount, so they decide alike.  A slice that finds the ring too full is DROPPED, not refused: its frame
    // is already in the accumulators, writeSlice does not run, nothing is emitted and nothing shifts -- the following
    // frames pile up on the same overlap-add position until the caller has retrieved enough.
    {
        const int required = (d_.constant || d_.vocoder) ? d_.hop : int(shiftInc / d_.pitch_scale) + 1;
        const int64_t ws = d_.outbuf_cap - out_fill_;
        if (ws < required) {
            r.adv = 0;
            r.cnt = 0;
            if (d_.constant) r.flags |= kSliceUpperChannelsSkip;
            ++slices_;
            ++dropped_;
            out.push_back(r);
            return PV_OK;
        }
    }
    const int64_t Pn = P_ + (int64_t)shiftInc;
    int64_t Kn;
    if (d_.resample) {
        // outputs k with filt_len/2 + floor(k*num/den) < Pn  (speex loop condition, resample.c:362,474)
        int64_t X = Pn - d_.filt_len / 2;
        if (X <= 0) Kn = 0;
        else Kn = (int64_t)(((unsigned __int128)X * d_.res_den + d_.res_num - 1) / d_.res_num);
        // the per-call output cap of RS_Speex::doresample (dsp/resampler.cc:783) must not bind
        int64_t cap = std::lrintf(std::ceil((float)((int)shiftInc * d_.res_ratio)));
        if (Kn - K_ > cap) return PV_ERR_UNSUPPORTED;
    } else {
        Kn = Pn;
    }
    r.cnt = (int32_t)(Kn - K_);
    if (r.cnt > d_.outbuf_cap - out_fill_) {
        // The guard above admits a slice by an estimate (CONSTANT mode: one input hop) that an up-sampling resampler
        // exceeds: the reference's ring then takes only part of the slice's output and loses the rest
        // (circularqueue.h:337-341).  A stream with samples missing mid-slice is not worth reproducing: refused.
        g_plan_reason = "the reference's output ring would truncate a slice's output here (CONSTANT mode through an "
                        "up-sampling resampler with more output pending than the ring holds): retrieve between calls";
        return PV_ERR_OUTPUT_OVERRUN;
    }
    out_fill_ += r.cnt;
    P_ = Pn;
    K_ = Kn;
    ++slices_;
    out.push_back(r);
    return PV_OK;
}

int Planner::feed(int64_t n, std::vector<SliceRec> &out) {
    if (n < 0) return PV_ERR_INVALID_ARG;
    int64_t remaining = n;
    bool allread = false;
    const int64_t cap = 2 * (int64_t)d_.N; // input ring capacity (channelinfo.cc:31-35)
    while (!allread) {
        int64_t w = cap - in_fill_;
        if (w > remaining) w = remaining;
        in_fill_ += w;
        remaining -= w;
        allread = remaining == 0;
        int st = try_slice(out);
        if (st != PV_OK) return st;
    }
    return PV_OK;
}

int plan_batch(const Derived &d, int64_t frames, int block, bool flush, BatchPlan &bp) {
    if (frames < 0 || block < 1) return PV_ERR_INVALID_ARG;
    Planner pl(d);
    bp.slices.clear();
    int64_t produced = 0, fed = 0;
    for (int64_t i = 0; i < frames; i += block) {
        int64_t n = frames - i < block ? frames - i : block;
        int st = pl.feed(n, bp.slices);
        if (st != PV_OK) return st;
        fed += n;
        produced += pl.retrieve(pl.available());
    }
    if (flush) {
        int guard = 0;
        while (produced < frames) {
            int st = pl.feed(block, bp.slices);
            if (st != PV_OK) return st;
            fed += block;
            int32_t got = pl.retrieve(pl.available());
            produced += (frames - produced > got) ? got : (frames - produced);
            if (++guard > (1 << 24)) return PV_ERR_UNSUPPORTED;
        }
    }
    bp.out_frames = produced;
    bp.in_frames = fed;
    return PV_OK;
}

// Rosenberg glottal pulse (gen/rosenberg.cc:19-53): per period of round(sample_rate / f) samples an opening phase
// of n1 = round(alpha * period) samples, 0.5 (1 - cos(pi n / n1)), a closing phase of n2 = round(beta * period)
// samples, cos(pi (n - n1) / (2 n2)), then silence; the sample counter runs 0 .. period inclusive.  Arithmetic
// contract: reciprocals stored as float (1 / n1 in float, 0.5 / n2 in double), cosf of a float argument formed in
// double, the opening phase scaled in double.  (n1 = 0 at low sample rates makes the reference emit NaN: kept.)
void CarrierGen::init(Voice &v, float sample_rate, float freq, float alpha, float beta) {
    v.phase = 0;
    v.period = (int)std::round(1.f / freq * sample_rate);
    v.n1 = (int)std::round(alpha * v.period);
    v.n2 = (int)std::round(beta * v.period);
    v.inv_n1 = 1.f / static_cast<float>(v.n1);
    v.inv_2n2 = (float)(0.5 / static_cast<float>(v.n2));
}

float CarrierGen::step(Voice &v) {
    const int n = v.phase;
    v.phase = n + 1 > v.period ? 0 : n + 1;
    if (n <= v.n1) return (float)(0.5 * (1 - cosf((float)(M_PI * n * v.inv_n1))));
    if (n - v.n1 <= v.n2) return cosf((float)(M_PI * (n - v.n1) * v.inv_2n2));
    return 0;
}

// one voice at 440 Hz, or the A-minor triad 440 / 523.251 / 659.255 Hz averaged (rosenbergchord.cc:19-43); both
// scaled by 0.3 in double (phasevocoderprocess.cc:96-107)
CarrierGen::CarrierGen(float sample_rate, bool chord) : nv_(chord ? 3 : 1) {
    static const float kTriad[3] = {440, 523.251f, 659.255f};
    for (int i = 0; i < nv_; ++i) init(v_[i], sample_rate, kTriad[i], 0.01f, 0.06f);
}

float CarrierGen::next() {
    if (nv_ == 1) return (float)(step(v_[0]) * 0.3);
    float mix = 0;
    for (int i = 0; i < nv_; ++i) mix += step(v_[i]) / 3;
    return (float)(mix * 0.3);
}

// glibc rand(): TYPE_3 additive feedback generator x[i] = x[i-3] + x[i-31] (mod 2^32), output x >> 1, state
// seeded from seed 1 by the Lehmer generator 16807 * s mod (2^31 - 1) and warmed up with 310 draws.  Written
// from the published description of the algorithm; pinned against libc rand() in tests/test_host.py.
WhisperRng::WhisperRng() {
    int32_t r[34];
    r[0] = 1;
    for (int i = 1; i < 31; ++i) {
        const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        int64_t w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r[i] = (int32_t)w;
    }
    for (int i = 0; i < 31; ++i) state_[i] = r[i];
    f_ = 3;
    b_ = 0;
    for (int i = 0; i < 310; ++i) (void)next_raw();
}

uint32_t WhisperRng::next_raw() {
    const uint32_t v = (uint32_t)state_[f_] + (uint32_t)state_[b_];
    state_[f_] = (int32_t)v;
    const uint32_t out = v >> 1;
    if (++f_ >= 31) f_ = 0;
    if (++b_ >= 31) b_ = 0;
    return out;
}

float WhisperRng::next_phase() {
    const float two_pi = (float)(2 * M_PI);
    return two_pi * (float)next_raw() / (float)2147483647; // RAND_MAX
}

int64_t bytes_per_slice(const Derived &d) {
    // B_slice = 4*(3N + 7H + 2s + h); without resampling 4*(3N + 7H + s)
    int64_t N = d.N, H = d.H, s = d.hop_out_nominal, h = d.hop;
    return d.resample ? 4 * (3 * N + 7 * H + 2 * s + h) : 4 * (3 * N + 7 * H + s);
}

} // namespace pv
