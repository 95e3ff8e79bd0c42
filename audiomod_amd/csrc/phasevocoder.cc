// phasevocoder.cc -- the drop-in class over the C ABI.  Mirrors the call semantics of the reference
// wrapper (reference src/phasevocoder/phasevocoder.cc:87-183): which modes each entry point serves,
// when outputReady() turns false, and that getOutData clamps to what the last processInData made
// available.  The reference's stdout chatter (:76-83,130,147) is not reproduced.
#include "dafx/phasevocoder.h"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "audiomod_pv.h"

namespace audiomod {

static bool served_by_process_normal(int mode) {
    // the engine runs CONSTANT through the same scheduling entry point (processConstant differs only in the
    // per-slice work, reference phasevocoderimpl.cc:372-396)
    return mode == NORMAL_STRETCH || mode == NORMAL_SHIFT || mode == GENDER_CHANGE || mode == FORMANT_PRESERVE ||
           mode == ROBOTIC || mode == WHISPER || mode == CONSTANT || mode == VOCODER_ROSENBERG ||
           mode == VOCODER_CHORD || // (the vocoder modes return the shaped carrier, retrieveCarrier)
           mode == FORMANT_CEPSTRAL;
}

phasevocoder::phasevocoder(int sampleRate, int numChannels, float timeratio, float pitchshift, int mode,
                           int coremode, int fftsize, int hopsize)
    : engine_(nullptr), mode_(mode), outready_(false) {
    modbase::sample_rate_ = sampleRate;
    modbase::num_channels_ = numChannels;
    modbase_offline::sample_rate_ = sampleRate;
    modbase_offline::num_channels_ = numChannels;
    pv_config cfg;
    cfg.sample_rate = sampleRate;
    cfg.channels = numChannels;
    cfg.time_ratio = timeratio;
    cfg.pitch_semitones = pitchshift;
    cfg.mode = mode;
    cfg.coremode = coremode;
    cfg.fftsize = fftsize;
    cfg.hopsize = hopsize;
    int device = 0;
    if (const char *env = std::getenv("AUDIOMOD_PV_DEVICE")) device = std::atoi(env);
    const int st = pv_create(&cfg, device, &engine_);
    if (st != PV_OK) {
        throw std::runtime_error(std::string("audiomod::phasevocoder (MI355X engine): ") + pv_strerror(st) + ": " +
                                 pv_last_error());
    }
}

phasevocoder::~phasevocoder() {
    pv_destroy(engine_);
    engine_ = nullptr;
}

// The reference's entry points neither return a status nor throw: a shortfall shows as fewer samples /
// outputReady() == false plus a line on stderr (phasevocoder.cc:146-151, circularqueue.h:337-341).  Same here: a
// call the engine refuses leaves it untouched (pv_feed is transactional), the caller sees no new samples.
static void report(const char *where, int st) {
    std::fprintf(stderr, "audiomod::phasevocoder (MI355X engine): %s: %s: %s\n", where, pv_strerror(st), pv_last_error());
}

void phasevocoder::processInData(float *const *inData, int num_in_samples) {
    int numres = 0;
    if (served_by_process_normal(mode_)) {
        const int st = pv_feed(engine_, inData, num_in_samples);
        if (st != PV_OK) report("processInData", st);
        numres = pv_available(engine_);
        if (numres < 0) numres = 0;
    }
    num_res_ = numres;
}

void phasevocoder::getOutData(float *const *outData, int num_out_samples) {
    if (num_out_samples > num_res_) num_out_samples = num_res_;
    if (served_by_process_normal(mode_) && num_out_samples > 0) pv_retrieve(engine_, outData, num_out_samples);
    outready_ = true; // offline mode: always ready
}

void phasevocoder::processBlock(float *const *bufferData, int num_samples) {
    // the reference routes NORMAL_STRETCH nowhere in processBlock (phasevocoder.cc:134-144): it is a no-op
    int ret = 0;
    if (served_by_process_normal(mode_) && mode_ != NORMAL_STRETCH) {
        const int st = pv_feed(engine_, bufferData, num_samples);
        if (st != PV_OK) {
            report("processBlock", st);
            outready_ = false; // the block stays as the caller passed it
            return;
        }
        const int numres = pv_available(engine_);
        num_res_ = numres;
        if (numres >= num_samples) {
            pv_retrieve(engine_, bufferData, num_samples);
            ret = 0;
        } else {
            ret = -1;
        }
    }
    outready_ = ret >= 0;
}

} // namespace audiomod
