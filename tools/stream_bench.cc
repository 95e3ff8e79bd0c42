// tools/stream_bench.cc -- wall-clock speed of the DROP-IN class (audiomod::phasevocoder over the C ABI, host
// buffers, PCIe inclusive) driven exactly like audiomod-exe: 480-frame processInData/getOutData calls.
// usage: stream_bench [seconds=60] [block=480] [channels=2]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "phasevocoder.h"

int main(int argc, char **argv) {
    const int secs = argc > 1 ? atoi(argv[1]) : 60, block = argc > 2 ? atoi(argv[2]) : 480, ch = argc > 3 ? atoi(argv[3]) : 2;
    const int sr = 48000;
    const long frames = (long)secs * sr;
    std::vector<std::vector<float>> x(ch, std::vector<float>(frames));
    for (int c = 0; c < ch; ++c)
        for (long i = 0; i < frames; ++i) {
            double t = (double)i / sr, v = 0;
            for (int k = 1; k < 12; ++k) v += std::sin(2 * M_PI * (220.0 + 57.0 * c) * k * t + k) / k;
            x[c][i] = (float)(std::round(0.1 * v * 32768.0) / 32768.0);
        }
    audiomod::phasevocoder pv(sr, ch, 1.0f, 4.0f, NORMAL_SHIFT, PHASE_LOCKED, 2048);
    modbase_offline *off = &pv;
    std::vector<std::vector<float>> ob(ch, std::vector<float>(block * 8));
    std::vector<float *> in(ch), out(ch);
    for (int c = 0; c < ch; ++c) out[c] = ob[c].data();
    std::vector<double> lat;
    long produced = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (long i = 0; i + block <= frames; i += block) {
        for (int c = 0; c < ch; ++c) in[c] = x[c].data() + i;
        auto a = std::chrono::steady_clock::now();
        off->processInData(in.data(), block);
        const int got = off->getOutSamples();
        off->getOutData(out.data(), got);
        auto b = std::chrono::steady_clock::now();
        lat.push_back(std::chrono::duration<double, std::micro>(b - a).count());
        produced += got;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::sort(lat.begin(), lat.end());
    printf("{\"seconds\": %d, \"block\": %d, \"channels\": %d, \"wall_s\": %.4f, \"x_realtime\": %.1f, \"call_us_median\": %.1f, "
           "\"call_us_p99\": %.1f, \"produced\": %ld}\n",
           secs, block, ch, dt, secs / dt, lat[lat.size() / 2], lat[(size_t)(lat.size() * 0.99)], produced);
    return 0;
}
