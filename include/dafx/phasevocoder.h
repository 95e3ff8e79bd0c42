// phasevocoder.h -- drop-in audiomod::phasevocoder backed by the MI355X engine.
//
// Same public surface as the reference class (reference include/dafx/phasevocoder.h:42-117):
// constructor signature and defaults (:54), processBlock / outputReady (real-time, :74,80),
// processInData / getOutData / getOutSamples (offline, :76-78), no-op setParams / getParams
// (:62-72) and the mode / coremode macros (:22-36).  Underneath it binds the C ABI of
// include/audiomod_pv.h instead of the reference's phasevocodercore.
#pragma once

#include "modbase.h"

// mode
#define CONSTANT -1
#define NORMAL_SHIFT 0
#define GENDER_CHANGE 1
#define FORMANT_PRESERVE 2
#define VOCODER_ROSENBERG 3
#define VOCODER_CHORD 4
#define NORMAL_STRETCH 5
#define ROBOTIC 6
#define WHISPER 7
// extension of this engine (not a reference mode): pitch shift with cepstral formant restoration, see
// PV_MODE_FORMANT_CEPSTRAL in audiomod_pv.h
#define FORMANT_CEPSTRAL 8
// coremode
#define NORMAL_PV 0
#define PHASE_LOCKED 1
#define INT_RATIO 2

struct pv_engine;

namespace audiomod {

class phasevocoder : public modbase, public modbase_offline {
  public:
    // timeratio: output length / input length; pitchshift: semitones; hopsize 0 = automatic.
    // Throws std::runtime_error when no MI355X is usable (there is no CPU fallback) or the arguments are invalid.
    phasevocoder(int sampleRate, int numChannels, float timeratio, float pitchshift, int mode = NORMAL_SHIFT,
                 int coremode = PHASE_LOCKED, int fftsize = 2048, int hopsize = 0);
    ~phasevocoder();

    void setParams(std::map<std::string, float> params) { (void)params; }
    void getParams(std::map<std::string, float> &params) { (void)params; }

    void processBlock(float *const *bufferData, int num_samples);
    void processInData(float *const *inData, int num_in_samples);
    void getOutData(float *const *outData, int num_out_samples);
    bool outputReady() { return outready_; }

  private:
    phasevocoder(const phasevocoder &) = delete;
    void operator=(const phasevocoder &) = delete;

    pv_engine *engine_;
    int mode_;
    bool outready_;
};

} // namespace audiomod
