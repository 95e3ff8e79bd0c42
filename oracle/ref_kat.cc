// oracle/ref_kat.cc -- TEST INFRASTRUCTURE ONLY.
//
// Unit-level known-answer generator: thin caller of the REAL reference's DSP primitives
// (linked from oracle/_ref/libaudiomod_ref.a, built by oracle/ref.mk):
//   audiomod::FFT::forwardPolar / inversePolar   /root/reference/src/common/dsp/FFT.h:50-113
//   audiomod::resampler::doresample              /root/reference/src/common/dsp/resampler.h:27-60
//   audiomod::windowfunc<float>(Hanning, N)      /root/reference/src/common/dsp/windowfunc.h:39-99
// Raw float32 files in and out; used by tools/make_golden.py to pin oracle/pv_oracle.c.
//
//   ref_kat fwd N nframes in.f32 out.f32        in: [nframes][N]        out: [nframes][2][N/2+1] (mag, phase)
//   ref_kat inv N nframes in.f32 out.f32        in: [nframes][2][N/2+1] out: [nframes][N]
//   ref_kat win N out.f32                       out: N window values then 1 float area
//   ref_kat res ratio nin in.f32 out.f32 counts.txt chunk0 chunk1 ...   (chunks cycle until nin consumed)
#include "common/dsp/FFT.h"
#include "common/dsp/resampler.h"
#include "common/dsp/windowfunc.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace audiomod;

static std::vector<float> slurp(const char *p, size_t n) {
    std::vector<float> v(n);
    FILE *f = fopen(p, "rb");
    if (!f || fread(v.data(), 4, n, f) != n) { fprintf(stderr, "read fail %s\n", p); exit(1); }
    fclose(f);
    return v;
}
static void dump(const char *p, const std::vector<float> &v) {
    FILE *f = fopen(p, "wb");
    fwrite(v.data(), 4, v.size(), f);
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::string cmd = argv[1];
    if (cmd == "fwd" || cmd == "inv") {
        int N = atoi(argv[2]), nf = atoi(argv[3]);
        int H = N / 2 + 1;
        FFT fft(N);
        if (cmd == "fwd") {
            std::vector<float> in = slurp(argv[4], (size_t)nf * N), out((size_t)nf * 2 * H);
            for (int t = 0; t < nf; ++t)
                fft.forwardPolar(in.data() + (size_t)t * N, out.data() + (size_t)t * 2 * H, out.data() + (size_t)t * 2 * H + H);
            dump(argv[5], out);
        } else {
            std::vector<float> in = slurp(argv[4], (size_t)nf * 2 * H), out((size_t)nf * N);
            for (int t = 0; t < nf; ++t)
                fft.inversePolar(in.data() + (size_t)t * 2 * H, in.data() + (size_t)t * 2 * H + H, out.data() + (size_t)t * N);
            dump(argv[5], out);
        }
    } else if (cmd == "win") {
        int N = atoi(argv[2]);
        windowfunc<float> w(Hanning, N);
        std::vector<float> out(N + 1);
        for (int i = 0; i < N; ++i) out[i] = w.GetValue(i);
        out[N] = w.GetArea();
        dump(argv[3], out);
    } else if (cmd == "res") {
        float ratio = (float)atof(argv[2]);
        long nin = atol(argv[3]);
        std::vector<float> in = slurp(argv[4], nin);
        std::vector<int> chunks;
        for (int i = 7; i < argc; ++i) chunks.push_back(atoi(argv[i]));
        resampler rs(resampler::FastestTolerable, 1, 4096 * 16);
        rs.reset();  // channelinfo::reset() does this before first use (channelinfo.cc:97)
        std::vector<float> out, tmp(1 << 20);
        FILE *fc = fopen(argv[6], "w");
        long pos = 0;
        size_t ci = 0;
        while (pos < nin) {
            int c = chunks[ci++ % chunks.size()];
            if (pos + c > nin) c = (int)(nin - pos);
            const float *ip = in.data() + pos;
            float *op = tmp.data();
            int got = rs.doresample(&ip, &op, c, ratio, false);
            out.insert(out.end(), tmp.begin(), tmp.begin() + got);
            fprintf(fc, "%d %d\n", c, got);
            pos += c;
        }
        fclose(fc);
        dump(argv[5], out);
    } else {
        return 2;
    }
    return 0;
}
