// tests/native/host_plan_sanitize.cc -- the host planner (audiomod_amd/csrc/pv_plan.cc: derived constants, tables,
// integer slice scheduler, whole-job plans, carrier and whisper generators) driven over a matrix of
// configurations under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; GPU sanitizers are not
// available on the pool).  Also checks a few planner invariants that must hold for any configuration.
// Build: g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off \
//            -Iinclude -Iaudiomod_amd/csrc tests/native/host_plan_sanitize.cc audiomod_amd/csrc/pv_plan.cc
#include <cstdio>
#include <vector>

#include "pv_plan.h"

using namespace pv;

static int check_plan(const pv_config &cfg, int64_t frames, int block, bool flush) {
    Derived d;
    const int st = derive(cfg, d);
    if (st != PV_OK) return st == PV_ERR_UNSUPPORTED || st == PV_ERR_INVALID_ARG ? 0 : 1;
    BatchPlan bp;
    const int sp = plan_batch(d, frames, block, flush, bp);
    if (sp != PV_OK) return sp == PV_ERR_UNSUPPORTED || sp == PV_ERR_OUTPUT_OVERRUN ? 0 : 1;
    int64_t P = 0, K = 0;
    for (const SliceRec &s : bp.slices) {
        if (s.P != P || s.K0 != K || s.shift < d.min_shift || s.cnt < 0) {
            std::printf("plan invariant broken: mode %d ratio %g semis %g fft %d\n", cfg.mode, cfg.time_ratio,
                        cfg.pitch_semitones, cfg.fftsize);
            return 1;
        }
        P += s.adv; // (a dropped slice -- adv 0 -- leaves the overlap-add position where it is)
        K += s.cnt;
        if ((s.adv != s.shift && s.adv != 0) || (s.adv == 0 && s.cnt != 0)) return 1;
    }
    if (bp.out_frames < 0 || bp.out_frames > K) return 1;
    // the streaming planner fed in odd-sized calls must run the same slices
    Planner pl(d);
    std::vector<SliceRec> got;
    int64_t fed = 0;
    const int sizes[] = {1, 63, 480, 4800, 7, 2049};
    int si = 0;
    while (fed < frames) {
        int64_t n = sizes[si++ % 6];
        if (n > frames - fed) n = frames - fed;
        const int fs = pl.feed(n, got);
        if (fs == PV_ERR_OUTPUT_OVERRUN) return 0; // the one overrun case the planner refuses (truncated slice output)
        if (fs != PV_OK) return 1;
        pl.retrieve(pl.available());
        fed += n;
    }
    // (a call larger than the reference's output ring makes it drop slices: which ones depends on the call sizes,
    // so the two plans are comparable only while neither has dropped anything)
    if (pl.dropped() > 0) return 0;
    for (const SliceRec &s : bp.slices)
        if (s.adv == 0) return 0;
    for (size_t i = 0; i < got.size() && i < bp.slices.size(); ++i)
        if (got[i].shift != bp.slices[i].shift || got[i].P != bp.slices[i].P || got[i].K0 != bp.slices[i].K0) {
            std::printf("streaming / batch plans differ at slice %zu\n", i);
            return 1;
        }
    return 0;
}

int main() {
    int bad = 0, n = 0;
    const int modes[] = {PV_MODE_CONSTANT, PV_MODE_NORMAL_SHIFT, PV_MODE_GENDER_CHANGE, PV_MODE_FORMANT_PRESERVE,
                         PV_MODE_VOCODER_ROSENBERG, PV_MODE_VOCODER_CHORD, PV_MODE_NORMAL_STRETCH, PV_MODE_ROBOTIC,
                         PV_MODE_WHISPER};
    const float semis[] = {-12.f, -7.f, -0.5f, 0.f, 4.f, 7.f, 12.f};
    const float ratios[] = {0.5f, 0.75f, 1.f, 1.5f, 2.f, 3.f};
    const int ffts[] = {256, 1000, 2048, 4096, 8192};
    for (int mode : modes)
        for (int core = 0; core < 3; ++core)
            for (int fft : ffts)
                for (float st : semis)
                    for (float r : ratios) {
                        if (mode != PV_MODE_NORMAL_STRETCH && r != 1.f) continue;
                        if (mode == PV_MODE_NORMAL_STRETCH && st != 0.f) continue;
                        pv_config cfg{48000, 2, r, st, mode, core, fft, 0};
                        bad += check_plan(cfg, 48000, 480, mode != PV_MODE_NORMAL_STRETCH);
                        ++n;
                    }
    // odd sample rates / channel counts / explicit hops / nonsense
    const pv_config odd[] = {{44100, 1, 1.f, 3.f, PV_MODE_NORMAL_SHIFT, 1, 2048, 0},
                             {8000, 6, 1.f, -5.f, PV_MODE_FORMANT_PRESERVE, 0, 512, 0},
                             {96000, 2, 1.25f, 0.f, PV_MODE_NORMAL_STRETCH, 1, 4096, 0},
                             {48000, 2, 1.f, 4.f, PV_MODE_NORMAL_SHIFT, 1, 2048, 128},
                             {48000, 2, -1.f, 4.f, PV_MODE_NORMAL_SHIFT, 1, 2048, 0},
                             {48000, 0, 1.f, 4.f, PV_MODE_NORMAL_SHIFT, 1, 2048, 0},
                             {48000, 2, 1.f, 4.f, 99, 7, 3, 0}};
    for (const pv_config &cfg : odd) {
        bad += check_plan(cfg, 30000, 333, true);
        ++n;
    }
    CarrierGen cg(48000.f, true), cg1(44100.f, false);
    double acc = 0;
    for (int i = 0; i < 100000; ++i) acc += cg.next() + cg1.next();
    WhisperRng wr;
    for (int i = 0; i < 100000; ++i) acc += wr.next_phase();
    std::printf("%d configurations, %d failures (checksum %.3f)\n", n, bad, acc);
    return bad != 0;
}
