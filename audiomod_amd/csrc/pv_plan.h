// pv_plan.h -- host-side, data-independent part of the phase-vocoder engine:
// derived constants, precomputed tables and the integer "slice scheduler".
//
// Everything here is pure host C++ (no HIP), so it is unit-tested on CPU through
// pv_plan_simulate() (include/audiomod_pv.h).  It restates, from the behaviour documented
// in SURVEY.md section 8(a), the reference's
//   - constructor / calculateSizes   (phasevocoder.cc:24-85, phasevocoderimpl.cc:169-263)
//   - processNormal scheduling loop   (phasevocoderimpl.cc:340-369, phasevocoderprocess.cc:43-64,236-303)
//   - calculateIncrements             (phasevocoderprocess.cc:379-489)
//   - Speex rate set-up + filter table(resampler.cc:740-770, speex/resample.c:285-351,661-913,1117-1158)
//   - kissfft twiddles / factor order (kissfft/kiss_fft.c:292-347, kiss_fftr.c:57-63)
//   - periodic Hann window            (dsp/windowfunc.h:129-169)
#pragma once

#include <cstdint>
#include <vector>

#include "audiomod_pv.h"

namespace pv {

struct cpx {
    float r, i;
};

constexpr int kMaxStages = 16;

struct FftPlan {
    int nc = 0;      // complex FFT length = N/2
    int nstages = 0; // executed innermost first: stage 0 is the deepest recursion level
    int radix[kMaxStages];
    int m[kMaxStages];
    int fstride[kMaxStages];
    std::vector<int32_t> perm;   // out index -> in index (the recursion's leaf copy)
    std::vector<cpx> tw_fwd, tw_inv; // nc twiddles each
    std::vector<cpx> st_fwd, st_inv; // nc "super twiddles" each
};

struct Derived {
    pv_config cfg;
    int N = 0, hs = 0, H = 0, hop = 0, hop_out_nominal = 0;
    int outbuf_cap = 0; // output ring capacity (frames)
    float time_ratio = 1, pitch_scale = 1, hs_ratio = 1;
    bool int_ratio = false, resample = false;
    bool robotic = false;  // phase = 0 (roboticSlice)
    bool whisper = false;  // phase = 2*pi*rand()/RAND_MAX (whisperSlice)
    bool constant = false; // CONSTANT mode: no phase modification, out hop == in hop
    bool vocoder = false;  // channel vocoder: Rosenberg carrier shaped by the input's band magnitudes
    bool chord = false;    // VOCODER_CHORD: three-voice A-minor carrier
    bool cepstral = false; // FORMANT_CEPSTRAL (extension): cepstral envelope shift by env_comp before synthesis
    float env_comp = 1;
    int voc_band_len = 0;  // bins per band = floor(N / 1024) (modifySliceVocoder)
    bool do_freq_comp = false;
    float freq_comp = 1, fixed_gain = 1;
    double two_pi_hop = 0; // (2*M_PI)*hop in double, the common factor of omega / pomega / delta_omega
    float inv_n = 0;       // 1.f / N
    // window
    std::vector<float> window;
    float win_area = 0, win_gain = 0; // gain = float(area * 1.5)
    // resampler (valid when resample)
    float res_ratio = 1;
    uint32_t res_num = 1, res_den = 1; // speex num_rate / den_rate (input step = num/den)
    int filt_len = 0, oversample = 0, int_adv = 0, frac_adv = 0;
    bool interp = true;
    std::vector<float> sinc;
    FftPlan fft;
    int min_shift = 1; // lower bound of any shift increment (sizes the frame ring)
};

// Fills d from cfg.  Returns PV_OK / PV_ERR_INVALID_ARG / PV_ERR_UNSUPPORTED.
int derive(const pv_config &cfg, Derived &d);

struct SliceRec {
    int32_t shift;     // shiftIncrement s_t
    int32_t phase_inc; // phaseIncrement
    int64_t P;         // OLA-stream position of this slice's frame = sum of the advances before it
    int64_t K0;        // outputs emitted before this slice
    int32_t cnt;       // outputs emitted by this slice
    int32_t adv;       // how far the OLA stream advances after this slice: its shift, or 0 when the reference finds its
                       // output ring too full and drops the slice (phasevocoderprocess.cc:337-364): the frame has been
                       // added to the accumulators, writeSlice is skipped, the next frame lands on the same position
    int32_t flags;     // kSliceUpperChannelsSkip: CONSTANT mode returns from its channel loop at the first full ring
                       // (processOneSliceConstant :139-150), so only channel 0 has added this slice's frame
};
enum : int32_t { kSliceUpperChannelsSkip = 1 };

// Integer simulation of the reference's ring occupancy and increment recurrences.
// why the planner last refused something (thread-local static text, "" if it has not); cleared by the C-ABI entry
// points and folded into pv_last_error()
const char *plan_reason();
void plan_reason_clear();

class Planner {
  public:
    explicit Planner(const Derived &d) : d_(d) {}
    // == one processNormal(n) call; appends the slices it runs to `out`.
    int feed(int64_t n, std::vector<SliceRec> &out);
    int32_t available() const { return (int32_t)out_fill_; }
    int32_t retrieve(int32_t n) {
        int32_t g = n < out_fill_ ? n : (int32_t)out_fill_;
        if (g < 0) g = 0;
        out_fill_ -= g;
        return g;
    }
    int64_t slices() const { return slices_; }
    int64_t dropped() const { return dropped_; } // slices the reference would have dropped so far
    int64_t outputs() const { return K_; }
    int64_t ola_len() const { return P_; }
    // The whole mutable state, so that a caller can make feed() transactional: save, feed, and put the state back
    // when anything about the call fails (pv_feed: the engine then stays exactly where it was).
    struct State {
        int64_t in_fill, out_fill, slices, dropped, K, P, prev_increment;
        float recovery, divergence;
    };
    State save() const {
        return State{in_fill_, out_fill_, slices_, dropped_, K_, P_, prev_increment_, recovery_, divergence_};
    }
    void restore(const State &s) {
        in_fill_ = s.in_fill, out_fill_ = s.out_fill, slices_ = s.slices, dropped_ = s.dropped, K_ = s.K, P_ = s.P;
        prev_increment_ = s.prev_increment, recovery_ = s.recovery, divergence_ = s.divergence;
    }

  private:
    int try_slice(std::vector<SliceRec> &out);
    int next_increment();
    const Derived &d_;
    int64_t in_fill_ = 0, out_fill_ = 0, slices_ = 0, dropped_ = 0, K_ = 0, P_ = 0;
    float recovery_ = 0, divergence_ = 0;
    int64_t prev_increment_ = 0;
};

// Whole-job plan of the batch API: the reference CLI loop (main/main.cc:471-510) with `block`-frame calls.
struct BatchPlan {
    std::vector<SliceRec> slices;
    int64_t out_frames = 0; // frames the CLI would write
    int64_t in_frames = 0;
};
int plan_batch(const Derived &d, int64_t frames, int block, bool flush, BatchPlan &bp);

int64_t bytes_per_slice(const Derived &d);

// The phases whisperSlice (phasevocoderprocess.cc:814-822) draws in a fresh reference process: glibc's rand()
// from its default seed, two_pi * (float)rand() / (float)RAND_MAX in float.
// The carrier of the vocoder modes: Rosenberg glottal pulse train at 440 Hz, or the A-minor chord
// {440, 523.251, 659.255} Hz, scaled by 0.3 (reference common/gen/rosenberg.cc:19-53, rosenbergchord.cc:19-43,
// phasevocoderimpl.cc:312-320, phasevocoderprocess.cc:96-107).  Data-independent and identical for every channel.
class CarrierGen {
  public:
    CarrierGen(float sample_rate, bool chord);
    float next();

  private:
    struct Voice {
        int period, n1, n2, phase;
        float inv_n1, inv_2n2;
    };
    static void init(Voice &v, float sample_rate, float freq, float alpha, float beta);
    static float step(Voice &v);
    Voice v_[3];
    int nv_;
};

class WhisperRng {
  public:
    WhisperRng();
    float next_phase();
    uint32_t next_raw();

  private:
    int32_t state_[31];
    int f_, b_;
};

} // namespace pv
