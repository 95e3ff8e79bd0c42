// modbase.h -- abstract effect interfaces the phase-vocoder class plugs into.
//
// Source-compatible restatement of the two interfaces of the reference that the phase-vocoder
// path uses (reference include/dafx/modbase.h:26-66 `modbase`, :75-126 `modbase_offline`):
// same class names, virtuals, argument meaning and protected members, so a caller written
// against the reference header (e.g. main/main.cc:170,471-510,561-572) compiles unchanged.
// When building inside the reference tree, use the reference's own modbase.h instead -- this file
// exists so the drop-in class can be built and tested stand-alone.
#pragma once

#include <map>
#include <string>

// real-time (in-place, block by block) effect
class modbase {
  public:
    modbase() : sample_rate_(48000), num_channels_(1) {}
    virtual ~modbase() {}
    // bufferData[c] -> num_samples floats of channel c, processed in place
    virtual void processBlock(float *const *bufferData, int num_samples) = 0;
    virtual void setParams(std::map<std::string, float> params) = 0;
    virtual void getParams(std::map<std::string, float> &params) = 0;
    // false when the last processBlock could not fill the block (caller skips it)
    virtual bool outputReady() { return true; }

  protected:
    int sample_rate_;
    int num_channels_;
};

// offline effect: output length may differ from input length
class modbase_offline {
  public:
    modbase_offline() : sample_rate_(48000), num_channels_(1), num_res_(0) {}
    virtual ~modbase_offline() {}
    virtual void processInData(float *const *inData, int num_in_samples) = 0;
    // copies min(num_out_samples, getOutSamples()) frames per channel
    virtual void getOutData(float *const *outData, int num_out_samples) = 0;
    virtual void setParams(std::map<std::string, float> params) = 0;
    virtual void getParams(std::map<std::string, float> &params) = 0;
    virtual int getOutSamples() const { return num_res_; }
    virtual bool outputReady() { return true; }

  protected:
    int sample_rate_;
    int num_channels_;
    int num_res_; // frames available after the last processInData
};
