// oracle/ref_formant.cc -- thin caller of the reference's cepstral formant shift (TEST INFRASTRUCTURE ONLY).
//
// phasevocodercore::Impl::formantShiftSlice (reference src/phasevocoder/phasevocoderprocess.cc:925-999) is dead
// code upstream: every call site is commented out (:826,832,838,1017,1021), so no public API reaches it.  This
// harness links the compiled reference (oracle/ref.mk) and calls the method directly on magnitudes we supply, so
// the oracle's restatement (and through it the GPU kernel) can be pinned on the real function.  It is OUR code;
// the access-specifier defines below only let this one translation unit name a protected member, the reference
// objects it links against are compiled unmodified.
//
//   ref_formant <fftsize> <env_comp> <in.f32> <out.f32>     in/out: frames of fftsize/2+1 float32 magnitudes
#include <cstdio>
#include <cstdlib>
#include <vector>

#define private public
#define protected public
#include "phasevocoder/phasevocoderinterface.h"
#include "phasevocoder/phasevocoderimpl.h"
#include "phasevocoder/channelinfo.h"
#undef private
#undef protected

using namespace audiomod;

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: ref_formant fftsize env_comp in.f32 out.f32\n");
        return 2;
    }
    const int N = std::atoi(argv[1]);
    const float env = (float)std::atof(argv[2]);
    const int H = N / 2 + 1;
    FILE *fi = std::fopen(argv[3], "rb"), *fo = std::fopen(argv[4], "wb");
    if (!fi || !fo) return 2;
    phasevocodercore::setDefaultFftSize(N);
    phasevocodercore::Impl impl(48000, 1, 0, 1.0f, 1.0f);
    std::vector<float> m(H);
    while (std::fread(m.data(), sizeof(float), H, fi) == (size_t)H) {
        float *mag = impl.m_audioData[0]->mag;
        for (int i = 0; i < H; ++i) mag[i] = m[i];
        impl.formantShiftSlice(0, env);
        for (int i = 0; i < H; ++i) m[i] = mag[i];
        std::fwrite(m.data(), sizeof(float), H, fo);
    }
    std::fclose(fi);
    std::fclose(fo);
    return 0;
}
