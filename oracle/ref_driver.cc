// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never shipped, never measured as the product).
//
// Thin caller of the REAL reference's public class audiomod::phasevocoder
// (/root/reference/include/dafx/phasevocoder.h:42-117), linked against the reference
// objects that oracle/ref.mk compiles from the sources where they lie.  It reproduces the
// two drive loops of the reference CLI on raw float32 buffers instead of WAV files:
//   api=offline : processInData/getOutSamples/getOutData, main/main.cc:471-510
//                 (flush=1 -> the pitch-shift "flush to input length" loop, :492-509;
//                  flush=0 -> the time_stretch loop without flush, :471-478)
//   api=rt      : processBlock/outputReady, main/main.cc:561-572
// The reference keeps DSP state in process-global statics, so one process
// per run: that is why this is an executable and not a library.
//
// usage: ref_driver api in.f32 out.f32 counts.txt channels frames sr timeratio semitones
//                   mode coremode fftsize block flush
//   in.f32  : planar float32 [channels][frames]
//   out.f32 : planar float32 [channels][out_frames]   (out_frames printed to counts.txt line 1)
//   counts.txt : line 1 = out_frames, then one line per call = samples made available by it
#include "phasevocoder.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 15) {
        fprintf(stderr, "usage: %s api in out counts ch frames sr timeratio semis mode coremode fft block flush [hopsize]\n", argv[0]);
        return 2;
    }
    std::string api = argv[1];
    const char *inpath = argv[2], *outpath = argv[3], *cntpath = argv[4];
    int ch = atoi(argv[5]);
    long frames = atol(argv[6]);
    int sr = atoi(argv[7]);
    float timeratio = (float)atof(argv[8]);
    float semis = (float)atof(argv[9]);
    int mode = atoi(argv[10]);
    int coremode = atoi(argv[11]);
    int fftsize = atoi(argv[12]);
    int block = atoi(argv[13]);
    int flush = atoi(argv[14]);
    int hopsize = argc > 15 ? atoi(argv[15]) : 0; // 0 = automatic (phasevocoder.h:54)

    std::vector<std::vector<float>> in(ch, std::vector<float>(frames));
    FILE *f = fopen(inpath, "rb");
    if (!f) { perror("open in"); return 1; }
    for (int c = 0; c < ch; ++c) {
        if (fread(in[c].data(), sizeof(float), frames, f) != (size_t)frames) { fprintf(stderr, "short read\n"); return 1; }
    }
    fclose(f);

    audiomod::phasevocoder pv(sr, ch, timeratio, semis, mode, coremode, fftsize, hopsize);
    modbase *rt = &pv;
    modbase_offline *off = &pv;

    std::vector<std::vector<float>> out(ch);
    std::vector<int> counts;
    std::vector<float *> buff(ch), outbuff(ch);
    std::vector<std::vector<float>> bstore(ch, std::vector<float>(block)), ostore(ch, std::vector<float>(block * 64));
    for (int c = 0; c < ch; ++c) { buff[c] = bstore[c].data(); outbuff[c] = ostore[c].data(); }
    auto grow = [&](int n) { // one call can yield more than 64 blocks' worth (small calls, large hops)
        if ((size_t)n <= ostore[0].size()) return;
        for (int c = 0; c < ch; ++c) { ostore[c].resize((size_t)n); outbuff[c] = ostore[c].data(); }
    };

    if (api == "offline") {
        long produced = 0;
        for (long i = 0; i < frames; i += block) {
            int n = (int)((frames - i) < block ? (frames - i) : block);
            for (int c = 0; c < ch; ++c) memcpy(buff[c], in[c].data() + i, n * sizeof(float));
            off->processInData(buff.data(), n);
            int got = off->getOutSamples();
            grow(got);
            off->getOutData(outbuff.data(), got);
            for (int c = 0; c < ch; ++c) out[c].insert(out[c].end(), outbuff[c], outbuff[c] + got);
            counts.push_back(got);
            produced += got;
        }
        if (flush) {
            for (int c = 0; c < ch; ++c) memset(buff[c], 0, sizeof(float) * block);
            while (produced < frames) {
                off->processInData(buff.data(), block);
                int got = off->getOutSamples();
                grow(got);
                off->getOutData(outbuff.data(), got);
                counts.push_back(got);
                int w = got;
                if (frames - produced <= got) w = (int)(frames - produced);
                for (int c = 0; c < ch; ++c) out[c].insert(out[c].end(), outbuff[c], outbuff[c] + w);
                produced += w;
            }
        }
    } else if (api == "rt") {
        for (long i = 0; i < frames; i += block) {
            int n = (int)((frames - i) < block ? (frames - i) : block);
            for (int c = 0; c < ch; ++c) memcpy(buff[c], in[c].data() + i, n * sizeof(float));
            rt->processBlock(buff.data(), n);
            if (rt->outputReady()) {
                for (int c = 0; c < ch; ++c) out[c].insert(out[c].end(), buff[c], buff[c] + n);
                counts.push_back(n);
            } else {
                counts.push_back(-1);
            }
        }
    } else {
        fprintf(stderr, "unknown api %s\n", api.c_str());
        return 2;
    }

    f = fopen(outpath, "wb");
    if (!f) { perror("open out"); return 1; }
    for (int c = 0; c < ch; ++c) fwrite(out[c].data(), sizeof(float), out[c].size(), f);
    fclose(f);
    f = fopen(cntpath, "w");
    fprintf(f, "%zu\n", out[0].size());
    for (int v : counts) fprintf(f, "%d\n", v);
    fclose(f);
    return 0;
}
