// tools/shim_demo.cc -- drives the drop-in audiomod::phasevocoder exactly like the reference CLI's two loops
// (reference main/main.cc:471-510 offline, :561-572 real-time) on a raw planar float32 file.
// usage: shim_demo api in.f32 out.f32 counts.txt ch frames sr timeratio semis mode coremode fft block flush
// (same argument order as oracle/ref_driver.cc, so tests can compare the two programs' outputs directly)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "phasevocoder.h"

int main(int argc, char **argv) {
    if (argc < 15) return 2;
    std::string api = argv[1];
    int ch = atoi(argv[5]);
    long frames = atol(argv[6]);
    int sr = atoi(argv[7]);
    float timeratio = (float)atof(argv[8]), semis = (float)atof(argv[9]);
    int mode = atoi(argv[10]), coremode = atoi(argv[11]), fftsize = atoi(argv[12]), block = atoi(argv[13]);
    int flush = atoi(argv[14]);
    std::vector<std::vector<float>> in(ch, std::vector<float>(frames)), out(ch);
    FILE *f = fopen(argv[2], "rb");
    if (!f) return 1;
    for (int c = 0; c < ch; ++c)
        if (fread(in[c].data(), 4, frames, f) != (size_t)frames) return 1;
    fclose(f);
    audiomod::phasevocoder pv(sr, ch, timeratio, semis, mode, coremode, fftsize);
    modbase *rt = &pv;
    modbase_offline *off = &pv;
    std::vector<std::vector<float>> bs(ch, std::vector<float>(block)), os(ch, std::vector<float>(block * 64));
    std::vector<float *> buff(ch), outbuff(ch);
    for (int c = 0; c < ch; ++c) {
        buff[c] = bs[c].data();
        outbuff[c] = os[c].data();
    }
    std::vector<int> counts;
    if (api == "offline") {
        long produced = 0;
        for (long i = 0; i < frames; i += block) {
            int n = (int)((frames - i) < block ? (frames - i) : block);
            for (int c = 0; c < ch; ++c) memcpy(buff[c], in[c].data() + i, n * 4);
            off->processInData(buff.data(), n);
            int got = off->getOutSamples();
            off->getOutData(outbuff.data(), got);
            for (int c = 0; c < ch; ++c) out[c].insert(out[c].end(), outbuff[c], outbuff[c] + got);
            counts.push_back(got);
            produced += got;
        }
        if (flush) {
            for (int c = 0; c < ch; ++c) memset(buff[c], 0, 4 * block);
            while (produced < frames) {
                off->processInData(buff.data(), block);
                int got = off->getOutSamples();
                off->getOutData(outbuff.data(), got);
                counts.push_back(got);
                int w = (frames - produced > got) ? got : (int)(frames - produced);
                for (int c = 0; c < ch; ++c) out[c].insert(out[c].end(), outbuff[c], outbuff[c] + w);
                produced += w;
            }
        }
    } else {
        for (long i = 0; i < frames; i += block) {
            int n = (int)((frames - i) < block ? (frames - i) : block);
            for (int c = 0; c < ch; ++c) memcpy(buff[c], in[c].data() + i, n * 4);
            rt->processBlock(buff.data(), n);
            if (rt->outputReady()) {
                for (int c = 0; c < ch; ++c) out[c].insert(out[c].end(), buff[c], buff[c] + n);
                counts.push_back(n);
            } else {
                counts.push_back(-1);
            }
        }
    }
    f = fopen(argv[3], "wb");
    for (int c = 0; c < ch; ++c) fwrite(out[c].data(), 4, out[c].size(), f);
    fclose(f);
    f = fopen(argv[4], "w");
    fprintf(f, "%zu\n", out[0].size());
    for (int v : counts) fprintf(f, "%d\n", v);
    fclose(f);
    return 0;
}
