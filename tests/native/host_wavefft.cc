// tests/native/host_wavefft.cc -- CPU check of the wave-FFT core's index math (audiomod_amd/csrc/pv_wavefft.h).
// Runs the exact __host__ __device__ code lane by lane (64 emulated lanes -- 128 for the two-wave spec --, a plain array as the
// wave-private LDS region) and compares, bit for bit, with the straightforward permutation +
// level-by-level butterflies over the same plan tables (which the oracle pins to the reference).
// Build: g++ -O2 -std=c++17 -ffp-contract=off -Iinclude -Iaudiomod_amd/csrc tests/native/host_wavefft.cc \
//            audiomod_amd/csrc/pv_plan.cc -o /tmp/host_wavefft
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pv_plan.h"
#include "pv_wavefft.h"

using namespace pv;

static void ref_fft(const FftPlan &p, const std::vector<cf> &tw, bool inv, const std::vector<cf> &in, std::vector<cf> &out) {
    const int nc = p.nc;
    out.resize(nc);
    for (int j = 0; j < nc; ++j) out[j] = in[p.perm[j]];
    for (int s = 0; s < p.nstages; ++s) {
        const int radix = p.radix[s], m = p.m[s], fs = p.fstride[s];
        for (int base = 0; base < nc; base += radix * m)
            for (int k = 0; k < m; ++k) {
                cf *F = &out[base + k];
                if (radix == 4) {
                    if (inv) wf_bfly4<true>(F[0], F[m], F[2 * m], F[3 * m], tw[k * fs], tw[2 * k * fs], tw[3 * k * fs]);
                    else wf_bfly4<false>(F[0], F[m], F[2 * m], F[3 * m], tw[k * fs], tw[2 * k * fs], tw[3 * k * fs]);
                } else {
                    wf_bfly2(F[0], F[m], tw[k * fs]);
                }
            }
    }
}

template <class W, bool INV> static int check_spec(unsigned seed, const char *label) {
    constexpr int NC = W::N_C, LANES = W::LANES;
    pv_config cfg{48000, 1, 1.0f, 4.0f, 0, 1, 2 * NC, 0};
    Derived d;
    if (derive(cfg, d) != PV_OK) return 1;
    std::vector<cf> tw(NC);
    for (int i = 0; i < NC; ++i) tw[i] = INV ? cf{d.fft.tw_inv[i].r, d.fft.tw_inv[i].i} : cf{d.fft.tw_fwd[i].r, d.fft.tw_fwd[i].i};
    int bad = 0;
    for (int e = 0; e < NC; ++e) {
        if (wf_src_of<W>(e) != d.fft.perm[e]) ++bad;
        if (wf_e_of_src<W>(d.fft.perm[e]) != e) ++bad;
    }
    if (bad) { printf("NC %d: permutation map mismatch (%d)\n", NC, bad); return 1; }
    srand(seed);
    std::vector<cf> in(NC), want;
    for (auto &c : in) c = cf{(float)rand() / RAND_MAX - 0.5f, (float)rand() / RAND_MAX - 0.5f};
    in[3] = cf{0.f, -0.f};
    ref_fft(d.fft, tw, INV, in, want);
    std::vector<cf> lds(W::LDS_CF, cf{0, 0});
    
    // pass 0: load in pass-0 layout straight from the source array
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        const int lp = wf_lane_part<W>(0, lane);
        for (int r = 0; r < W::R; ++r) v[r] = in[wf_src_of<W>(lp | wf_reg_part<W>(0, r))];
        wf_fft_pass<W, 0, INV>(v, lane, lds.data(), tw.data());
    }
    for (int lane = 0; lane < LANES; ++lane) { cf v[W::R]; wf_fft_pass<W, 1, INV>(v, lane, lds.data(), tw.data()); }
    for (int lane = 0; lane < LANES; ++lane) { cf v[W::R]; wf_fft_pass<W, 2, INV>(v, lane, lds.data(), tw.data()); }
    if constexpr (W::NPASS == 4)
        for (int lane = 0; lane < LANES; ++lane) { cf v[W::R]; wf_fft_pass<W, 3, INV>(v, lane, lds.data(), tw.data()); }
    for (int e = 0; e < NC; ++e) {
        const cf g = lds[W::pad(e)];
        if (memcmp(&g, &want[e], sizeof(cf)) != 0) ++bad;
    }
    // the same transform through the prefetched-twiddle entry points the kernels use
    std::fill(lds.begin(), lds.end(), cf{0, 0});
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        const int lp = wf_lane_part<W>(0, lane);
        for (int r = 0; r < W::R; ++r) v[r] = in[wf_src_of<W>(lp | wf_reg_part<W>(0, r))];
        wf_load_pass_tw<W, 0>(T, lane, tw.data());
        wf_fft_pass_tw<W, 0, INV>(v, lane, lds.data(), T);
    }
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        wf_load_pass_tw<W, 1>(T, lane, tw.data());
        wf_fft_pass_tw<W, 1, INV>(v, lane, lds.data(), T);
    }
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        wf_load_pass_tw<W, 2>(T, lane, tw.data());
        wf_fft_pass_tw<W, 2, INV>(v, lane, lds.data(), T);
    }
    if constexpr (W::NPASS == 4)
        for (int lane = 0; lane < LANES; ++lane) {
            cf v[W::R];
            WfTw<W> T;
            wf_load_pass_tw<W, 3>(T, lane, tw.data());
            wf_fft_pass_tw<W, 3, INV>(v, lane, lds.data(), T);
        }
    for (int e = 0; e < NC; ++e) {
        const cf g = lds[W::pad(e)];
        if (memcmp(&g, &want[e], sizeof(cf)) != 0) ++bad;
    }
    // and with passes 1 and 2 taking their twiddles from the lane-major table the engine uploads
    std::vector<cf> table(2 * (size_t)wf_lane_table_entries<W>() * LANES);
    wf_build_lane_table<W>(tw.data(), table.data());
    const cf2 *tab2 = reinterpret_cast<const cf2 *>(table.data());
    std::fill(lds.begin(), lds.end(), cf{0, 0});
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        const int lp = wf_lane_part<W>(0, lane);
        for (int r = 0; r < W::R; ++r) v[r] = in[wf_src_of<W>(lp | wf_reg_part<W>(0, r))];
        wf_load_pass_tw<W, 0>(T, lane, tw.data());
        wf_fft_pass_tw<W, 0, INV>(v, lane, lds.data(), T);
    }
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        WfTwRaw<W, 1> raw;
        wf_fetch_pass_tw<W, 1>(raw, lane, tab2);
        wf_unpack_pass_tw<W, 1>(T, raw);
        wf_fft_pass_tw<W, 1, INV>(v, lane, lds.data(), T);
    }
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        WfTwRaw<W, 2> raw;
        wf_fetch_pass_tw<W, 2>(raw, lane, tab2);
        wf_unpack_pass_tw<W, 2>(T, raw);
        wf_fft_pass_tw<W, 2, INV>(v, lane, lds.data(), T);
    }
    if constexpr (W::NPASS == 4)
        for (int lane = 0; lane < LANES; ++lane) {
            cf v[W::R];
            WfTw<W> T;
            WfTwRaw<W, 3> raw;
            wf_fetch_pass_tw<W, 3>(raw, lane, tab2);
            wf_unpack_pass_tw<W, 3>(T, raw);
            wf_fft_pass_tw<W, 3, INV>(v, lane, lds.data(), T);
        }
    for (int e = 0; e < NC; ++e) {
        const cf g = lds[W::pad(e)];
        if (memcmp(&g, &want[e], sizeof(cf)) != 0) ++bad;
    }
    // ... and the free-form (PV_ARITH_FAST) stages: same transform, fma products and literal twiddles where the twiddle
    // index lives in register bits -- equal to the exact result up to rounding (relative to the spectrum's RMS)
    std::fill(lds.begin(), lds.end(), cf{0, 0});
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        const int lp = wf_lane_part<W>(0, lane);
        for (int r = 0; r < W::R; ++r) v[r] = in[wf_src_of<W>(lp | wf_reg_part<W>(0, r))];
        if (!wf_pass_all_const<W, 0>()) wf_load_pass_tw<W, 0>(T, lane, tw.data());
        wf_apply_pass_stages_fast<W, 0, INV>(v, T);
        wf_store<W, 0>(lds.data(), v, lp);
    }
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        const int lp = wf_lane_part<W>(1, lane);
        wf_load<W, 1>(lds.data(), v, lp);
        wf_load_pass_tw<W, 1>(T, lane, tw.data());
        wf_apply_pass_stages_fast<W, 1, INV>(v, T);
        wf_store<W, 1>(lds.data(), v, lp);
    }
    for (int lane = 0; lane < LANES; ++lane) {
        cf v[W::R];
        WfTw<W> T;
        const int lp = wf_lane_part<W>(2, lane);
        wf_load<W, 2>(lds.data(), v, lp);
        wf_load_pass_tw<W, 2>(T, lane, tw.data());
        wf_apply_pass_stages_fast<W, 2, INV>(v, T);
        wf_store<W, 2>(lds.data(), v, lp);
    }
    if constexpr (W::NPASS == 4)
        for (int lane = 0; lane < LANES; ++lane) {
            cf v[W::R];
            WfTw<W> T;
            const int lp = wf_lane_part<W>(3, lane);
            wf_load<W, 3>(lds.data(), v, lp);
            wf_load_pass_tw<W, 3>(T, lane, tw.data());
            wf_apply_pass_stages_fast<W, 3, INV>(v, T);
            wf_store<W, 3>(lds.data(), v, lp);
        }
    double err = 0, ref = 0;
    for (int e = 0; e < NC; ++e) {
        const cf g = lds[W::pad(e)];
        err += ((double)g.x - want[e].x) * ((double)g.x - want[e].x) + ((double)g.y - want[e].y) * ((double)g.y - want[e].y);
        ref += (double)want[e].x * want[e].x + (double)want[e].y * want[e].y;
    }
    const double rel = std::sqrt(err / ref);
    if (!(rel < 2e-7)) ++bad;
    printf("NC %d%s inv %d: %s (%d mismatches; lane table %d + %d (+ %d) slots; free-form stages: relative RMS %.2e, pass 0 %s)\n", NC, label,
           (int)INV, bad ? "FAIL" : "bit-exact", bad, wf_pass_slots<W>(1), wf_pass_slots<W>(2), W::NPASS == 4 ? wf_pass_slots<W>(3) : 0, rel,
           wf_pass_all_const<W, 0>() ? "all literal twiddles" : "fetched twiddles");
    return bad != 0;
}

template <int NC, bool INV> static int check(unsigned seed) { return check_spec<WF<NC>, INV>(seed, ""); }

int main() {
    int rc = 0;
    rc |= check<256, false>(7);
    rc |= check<256, true>(8);
    rc |= check<512, false>(5);
    rc |= check<512, true>(6);
    rc |= check<1024, false>(1);
    rc |= check<1024, true>(2);
    rc |= check<2048, false>(3);
    rc |= check<2048, true>(4);
    rc |= check_spec<WF2048S, false>(9, " on two waves");
    rc |= check_spec<WF2048S, true>(10, " on two waves");
    return rc;
}
