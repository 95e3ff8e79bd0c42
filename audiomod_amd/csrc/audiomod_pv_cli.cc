// audiomod_pv_cli.cc -- command-line driver for the phase-vocoder effects, same argv grammar and drive
// loops as the reference CLI for these effects (reference main/main.cc:24-93 usage, :196-286 argument
// parsing, :471-510 offline loops) over the drop-in audiomod::phasevocoder class, plus a from-scratch RIFF/WAVE
// reader/writer with the reference's sample conversions (reference main/wavfile.cc:733-755 int16 -> float
// * 1/32768; :1295-1306,1508-1527 float -> int16 saturate then truncate toward zero; header with a 'fact'
// chunk, wavfile.h:62-106, wavfile.cc:1135-1182).
//
//   audiomod-pv-exe time_stretch       in.wav out.wav <time_ratio> <coremode> <fftsize>
//   audiomod-pv-exe normal_pitchshift  in.wav out.wav <semitones>  <coremode> <fftsize>
//   audiomod-pv-exe formant_pitchshift in.wav out.wav <semitones>  <coremode> <fftsize>
//   audiomod-pv-exe gender_change      in.wav out.wav <semitones>  <coremode> <fftsize>
//   audiomod-pv-exe formant_cepstral   in.wav out.wav <semitones>  <coremode> <fftsize>   (extension of this engine)
//   audiomod-pv-exe robotic | whisper | vocoder | vocoder_chord   in.wav out.wav
//   audiomod-pv-exe constant           in.wav out.wav      (real-time processBlock loop, main.cc:561-572)
// Effects of the reference that are not phase-vocoder modes are not provided here.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "phasevocoder.h"

namespace {

struct WavIn {
    int channels = 0, rate = 0, bits = 0;
    long frames = 0;
    std::vector<unsigned char> data; // raw little-endian PCM of the 'data' chunk
    long pos = 0;                    // bytes consumed

    explicit WavIn(const char *path) {
        FILE *f = fopen(path, "rb");
        if (!f) throw std::runtime_error(std::string("cannot open ") + path);
        unsigned char h[12];
        if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) {
            fclose(f);
            throw std::runtime_error("not a RIFF/WAVE file");
        }
        // chunk lengths come from the file: never allocate or skip by more than the file still holds
        long file_size = 0;
        if (fseek(f, 0, SEEK_END) == 0) file_size = ftell(f);
        fseek(f, 12, SEEK_SET);
        bool have_fmt = false, have_data = false;
        int block_align = 0;
        while (!have_data) {
            unsigned char ch[8];
            if (fread(ch, 1, 8, f) != 8) break;
            const uint32_t len = ch[4] | (ch[5] << 8) | (ch[6] << 16) | ((uint32_t)ch[7] << 24);
            const long here = ftell(f);
            const long remaining = here >= 0 && file_size > here ? file_size - here : 0;
            if (!memcmp(ch, "fmt ", 4)) {
                if (len < 16 || len > 4096 || (long)len > remaining) break; // a format chunk is 16-40 bytes
                std::vector<unsigned char> b(len);
                if (fread(b.data(), 1, len, f) != len) break;
                if (len & 1) fseek(f, 1, SEEK_CUR); // chunks are word aligned: skip the pad byte
                const int fmt = b[0] | (b[1] << 8);
                channels = b[2] | (b[3] << 8);
                rate = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                block_align = b[12] | (b[13] << 8);
                bits = b[14] | (b[15] << 8);
                if (fmt != 1) {
                    fclose(f);
                    throw std::runtime_error("only PCM WAV files are supported");
                }
                have_fmt = true;
            } else if (!memcmp(ch, "data", 4)) {
                const size_t want = (long)len > remaining ? (size_t)remaining : (size_t)len; // a truncated file: what is there
                data.resize(want);
                const size_t got = want ? fread(data.data(), 1, want, f) : 0;
                data.resize(got);
                have_data = true;
            } else {
                const long skip = (long)len + (len & 1);
                if (skip > remaining || fseek(f, skip, SEEK_CUR) != 0) break;
            }
        }
        fclose(f);
        if (!have_fmt || !have_data || channels < 1 || (bits != 8 && bits != 16 && bits != 24 && bits != 32) ||
            block_align != channels * bits / 8)
            throw std::runtime_error("unsupported, inconsistent or truncated WAV file");
        frames = (long)(data.size() / (size_t)(channels * bits / 8));
    }

    // planar read of up to n frames; returns frames read (reference WavInFile::read(float**, int))
    int read(float *const *buf, int n) {
        const int bps = bits / 8;
        const long left = (long)(data.size() - (size_t)pos) / (channels * bps);
        if (n > left) n = (int)left;
        const unsigned char *p = data.data() + pos;
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < channels; ++c, p += bps) {
                float v;
                if (bps == 1) {
                    v = (float)(p[0] * (1.0 / 128.0) - 1.0);
                } else if (bps == 2) {
                    const int16_t s = (int16_t)(p[0] | (p[1] << 8));
                    v = (float)(s * (1.0 / 32768.0));
                } else if (bps == 3) {
                    int32_t s = p[0] | (p[1] << 8) | (p[2] << 16);
                    if (s & 0x00800000) s |= (int32_t)0xff000000;
                    v = (float)(s * (1.0 / 8388608.0));
                } else {
                    const int32_t s = (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24));
                    v = (float)(s * (1.0 / 2147483648.0));
                }
                buf[c][i] = v;
            }
        pos += (long)n * channels * bps;
        return n;
    }
};

struct WavOut16 {
    FILE *f = nullptr;
    int channels, rate;
    uint32_t bytes = 0;
    WavOut16(const char *path, int rate_, int channels_) : channels(channels_), rate(rate_) {
        f = fopen(path, "wb");
        if (!f) throw std::runtime_error(std::string("cannot open ") + path + " for writing");
        header();
    }
    static void put32(unsigned char *p, uint32_t v) {
        p[0] = v & 255;
        p[1] = (v >> 8) & 255;
        p[2] = (v >> 16) & 255;
        p[3] = (v >> 24) & 255;
    }
    static void put16(unsigned char *p, uint32_t v) {
        p[0] = v & 255;
        p[1] = (v >> 8) & 255;
    }
    // 56-byte header: RIFF(12) + fmt(24) + fact(12) + data(8), the layout the reference writes
    void header() {
        unsigned char h[56];
        const uint32_t bpf = (uint32_t)(2 * channels);
        memcpy(h, "RIFF", 4);
        put32(h + 4, bytes + 56 - 12 + 4);
        memcpy(h + 8, "WAVE", 4);
        memcpy(h + 12, "fmt ", 4);
        put32(h + 16, 16);
        put16(h + 20, 1);
        put16(h + 22, (uint32_t)channels);
        put32(h + 24, (uint32_t)rate);
        put32(h + 28, bpf * (uint32_t)rate);
        put16(h + 32, bpf);
        put16(h + 34, 16);
        memcpy(h + 36, "fact", 4);
        put32(h + 40, 4);
        put32(h + 44, bytes / bpf);
        memcpy(h + 48, "data", 4);
        put32(h + 52, bytes);
        fseek(f, 0, SEEK_SET);
        fwrite(h, 1, 56, f);
        fseek(f, 0, SEEK_END);
    }
    void write(float *const *buf, int n) {
        std::vector<unsigned char> tmp((size_t)n * channels * 2);
        unsigned char *p = tmp.data();
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < channels; ++c, p += 2) {
                float v = buf[c][i] * 32768.0f;
                if (v > 32767.0f) v = 32767.0f;
                else if (v < -32768.0f) v = -32768.0f;
                const int16_t s = (int16_t)(int)v; // truncation toward zero, like the reference
                put16(p, (uint16_t)s);
            }
        fwrite(tmp.data(), 1, tmp.size(), f);
        bytes += (uint32_t)tmp.size();
    }
    ~WavOut16() {
        if (f) {
            header();
            fclose(f);
        }
    }
};

int usage() {
    fprintf(stderr,
            "usage: audiomod-pv-exe dafx_name infile outfile <args> (dafx: constant, time_stretch, normal_pitchshift, "
            "formant_pitchshift, gender_change, vocoder, vocoder_chord, robotic, whisper)\n");
    return -1;
}

} // namespace

int main(int argc, char **argv) {
    if (argc < 4) return usage();
    const std::string model = argv[1];
    try {
        WavIn in(argv[2]);
        const int ch = in.channels, sr = in.rate;
        const long file_length = in.frames;
        std::unique_ptr<audiomod::phasevocoder> pv;
        bool flush = true;
        if (model == "time_stretch") {
            if (argc < 7) { fprintf(stderr, "err: not enough para (time_ratio, coremode, fftsize)\n"); return -1; }
            pv.reset(new audiomod::phasevocoder(sr, ch, (float)atof(argv[4]), 0, NORMAL_STRETCH, atoi(argv[5]), atoi(argv[6])));
            flush = false; // the reference's time_stretch loop has no flush (main.cc:471-478)
        } else if (model == "normal_pitchshift" || model == "formant_pitchshift" || model == "gender_change" ||
                   model == "formant_cepstral") { // formant_cepstral: extension, see PV_MODE_FORMANT_CEPSTRAL
            if (argc < 7) { fprintf(stderr, "err: not enough para (pitchshift_amount, coremode, fftsize)\n"); return -1; }
            const int mode = model == "normal_pitchshift"    ? NORMAL_SHIFT
                             : model == "formant_pitchshift" ? FORMANT_PRESERVE
                             : model == "formant_cepstral"   ? FORMANT_CEPSTRAL
                                                             : GENDER_CHANGE;
            pv.reset(new audiomod::phasevocoder(sr, ch, 1, (float)atof(argv[4]), mode, atoi(argv[5]), atoi(argv[6])));
        } else if (model == "robotic" || model == "whisper" || model == "vocoder" || model == "vocoder_chord" ||
                   model == "constant") {
            const int mode = model == "robotic" ? ROBOTIC : model == "whisper" ? WHISPER : model == "vocoder" ? VOCODER_ROSENBERG
                             : model == "vocoder_chord" ? VOCODER_CHORD : CONSTANT;
            pv.reset(new audiomod::phasevocoder(sr, ch, 1, 0, mode));
        } else {
            fprintf(stderr, "fx not supported by the MI355X phase-vocoder driver: %s\n", model.c_str());
            return usage();
        }
        modbase_offline *off = pv.get();
        WavOut16 out(argv[3], sr, ch);
        const int block = sr / 100 < 480 ? 480 : sr / 100; // main.cc:149
        if (model == "constant") { // "everything else - real time application" (main.cc:560-572)
            modbase *rt = pv.get();
            std::vector<std::vector<float>> bs(ch, std::vector<float>(block));
            std::vector<float *> buff(ch);
            for (int c = 0; c < ch; ++c) buff[c] = bs[c].data();
            for (long i = 0; i < file_length; i += block) {
                const int n = in.read(buff.data(), block);
                rt->processBlock(buff.data(), n);
                if (rt->outputReady()) out.write(buff.data(), n);
            }
            return 0;
        }
        std::vector<std::vector<float>> bs(ch, std::vector<float>(block)), os(ch, std::vector<float>(block * 4));
        std::vector<float *> buff(ch), outbuff(ch);
        for (int c = 0; c < ch; ++c) {
            buff[c] = bs[c].data();
            outbuff[c] = os[c].data();
        }
        long produced = 0;
        for (long i = 0; i < file_length; i += block) {
            const int n = in.read(buff.data(), block);
            off->processInData(buff.data(), n);
            const int got = off->getOutSamples();
            off->getOutData(outbuff.data(), got);
            out.write(outbuff.data(), got);
            produced += got;
        }
        if (flush) {
            for (int c = 0; c < ch; ++c) memset(buff[c], 0, sizeof(float) * block);
            while (produced < file_length) {
                off->processInData(buff.data(), block);
                const int got = off->getOutSamples();
                off->getOutData(outbuff.data(), got);
                const int w = (file_length - produced > got) ? got : (int)(file_length - produced);
                out.write(outbuff.data(), w);
                produced += w;
            }
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "audiomod-pv-exe: %s\n", e.what());
        return 1;
    }
    return 0;
}
