// pv_wavefft.h -- one-wavefront-per-frame complex FFT core (NC = 1024 or 2048 points) for gfx950.
//
// A 64-lane wave owns a whole frame: every lane keeps R = NC/64 complex elements in registers, runs
// two (or three) kissfft butterfly stages on them, and the wave re-distributes the elements through a
// wave-private LDS region between the three passes.  No workgroup barrier, three LDS round trips
// instead of five-plus, and 16/32 independent butterflies per lane for latency hiding.
//
// Arithmetic is kissfft's, operation for operation (kiss_fft.c:36-103): same stage order (innermost
// recursion level first: m = 1, 4, 16, ... for 1024 = 4^5; radix-2 then m = 2, 8, 32, ... for
// 2048 = 4^5 * 2), same twiddle table entries tw[k*fstride*q], same operand order -- so results are
// bit-identical to the reference (and to the generic LDS kernel in pv_kernels.hip).
//
// The element index e (position in the butterfly-ordered array) is treated as a bit string; each pass
// assigns some bits to the register index r and the other six to the lane.  Everything about the
// register side is resolved at compile time after full unrolling.
//
// The functions are __host__ __device__ so tests/host_wavefft.cc can run the exact same index math on
// the CPU, lane by lane, against the oracle.
#pragma once

#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PV_HD __host__ __device__ __forceinline__
#else
#define PV_HD inline
#endif

namespace pv {

struct alignas(8) cf {
    float x, y;
};

PV_HD cf wf_cmul(cf a, cf b) {
    cf m;
    m.x = a.x * b.x - a.y * b.y;
    m.y = a.x * b.y + a.y * b.x;
    return m;
}
PV_HD cf wf_add(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
PV_HD cf wf_sub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }

template <int NC> struct WF;

// Round 3: 512 complex points (fft 1024), eight elements per lane.  kissfft's stages for 512 are 2, 4, 4, 4, 4; with
// three register bits per pass a radix-4 stage fills a pass on its own from the second one on: FOUR passes (NPASS).
template <> struct WF<512> {
    static constexpr int LB = 6, LANES = 64; // lane bits: one wave per transform
    static constexpr int N_C = 512, LOG = 9, R = 8, RB = 3, NSTAGE = 5, NPASS = 4;
    static constexpr int st_radix[6] = {2, 4, 4, 4, 4, 0};
    static constexpr int st_bit[6] = {0, 1, 3, 5, 7, 0};
    static constexpr int st_pass[6] = {0, 0, 1, 2, 3, 0};
    static constexpr int regpos[4][5] = {{0, 1, 2, 0, 0}, {3, 4, 5, 0, 0}, {5, 6, 7, 0, 0}, {6, 7, 8, 0, 0}};
    static constexpr int lanepos[4][6] = {{3, 4, 5, 6, 7, 8}, {0, 1, 2, 6, 7, 8}, {0, 1, 2, 3, 4, 8}, {0, 1, 2, 3, 4, 5}};
    // e = b + 2 (d1 + 4 d2 + 16 d3 + 64 d4)  <-  src = d4 + 4 d3 + 16 d2 + 64 d1 + 256 b
    static constexpr int srcbit[11] = {8, 6, 7, 4, 5, 2, 3, 0, 1, 0, 0};
    static PV_HD int pad(int e) { return e + (e >> 4); }
    static constexpr int LDS_CF = 512 + 32 + 8;
};

// ... and 256 complex points (fft 512), four elements per lane: kissfft's 4 4 4 4, one stage per pass.
template <> struct WF<256> {
    static constexpr int LB = 6, LANES = 64; // lane bits: one wave per transform
    static constexpr int N_C = 256, LOG = 8, R = 4, RB = 2, NSTAGE = 4, NPASS = 4;
    static constexpr int st_radix[6] = {4, 4, 4, 4, 0, 0};
    static constexpr int st_bit[6] = {0, 2, 4, 6, 0, 0};
    static constexpr int st_pass[6] = {0, 1, 2, 3, 0, 0};
    static constexpr int regpos[4][5] = {{0, 1, 0, 0, 0}, {2, 3, 0, 0, 0}, {4, 5, 0, 0, 0}, {6, 7, 0, 0, 0}};
    static constexpr int lanepos[4][6] = {{2, 3, 4, 5, 6, 7}, {0, 1, 4, 5, 6, 7}, {0, 1, 2, 3, 6, 7}, {0, 1, 2, 3, 4, 5}};
    // e = d0 + 4 d1 + 16 d2 + 64 d3  <-  src = d3 + 4 d2 + 16 d1 + 64 d0
    static constexpr int srcbit[11] = {6, 7, 4, 5, 2, 3, 0, 1, 0, 0, 0};
    static PV_HD int pad(int e) { return e + (e >> 4); }
    static constexpr int LDS_CF = 256 + 16 + 8;
};

template <> struct WF<1024> {
    static constexpr int LB = 6, LANES = 64; // lane bits: one wave per transform
    static constexpr int N_C = 1024, LOG = 10, R = 16, RB = 4, NSTAGE = 5, NPASS = 3;
    static constexpr int st_radix[6] = {4, 4, 4, 4, 4, 0};
    static constexpr int st_bit[6] = {0, 2, 4, 6, 8, 0};
    static constexpr int st_pass[6] = {0, 0, 1, 1, 2, 0};
    static constexpr int regpos[4][5] = {{0, 1, 2, 3, 0}, {4, 5, 6, 7, 0}, {6, 7, 8, 9, 0}, {0, 0, 0, 0, 0}};
    static constexpr int lanepos[4][6] = {{4, 5, 6, 7, 8, 9}, {0, 1, 2, 3, 8, 9}, {0, 1, 2, 3, 4, 5}, {0, 0, 0, 0, 0, 0}};
    // kissfft's leaf copy: e = d0 + 4 d1 + 16 d2 + 64 d3 + 256 d4  <-  src = d4 + 4 d3 + 16 d2 + 64 d1 + 256 d0
    static constexpr int srcbit[11] = {8, 9, 6, 7, 4, 5, 2, 3, 0, 1, 0};
    static PV_HD int pad(int e) { return e + (e >> 4); }
    static constexpr int LDS_CF = 1024 + 64 + 8; // cf slots of the wave-private region
};

template <> struct WF<2048> {
    static constexpr int LB = 6, LANES = 64; // lane bits: one wave per transform
    static constexpr int N_C = 2048, LOG = 11, R = 32, RB = 5, NSTAGE = 6, NPASS = 3;
    static constexpr int st_radix[6] = {2, 4, 4, 4, 4, 4};
    static constexpr int st_bit[6] = {0, 1, 3, 5, 7, 9};
    static constexpr int st_pass[6] = {0, 0, 0, 1, 1, 2};
    static constexpr int regpos[4][5] = {{0, 1, 2, 3, 4}, {0, 5, 6, 7, 8}, {0, 7, 8, 9, 10}, {0, 0, 0, 0, 0}};
    static constexpr int lanepos[4][6] = {{5, 6, 7, 8, 9, 10}, {1, 2, 3, 4, 9, 10}, {1, 2, 3, 4, 5, 6}, {0, 0, 0, 0, 0, 0}};
    // e = b + 2 (d1 + 4 d2 + 16 d3 + 64 d4 + 256 d5)  <-  src = d5 + 4 d4 + 16 d3 + 64 d2 + 256 d1 + 1024 b
    static constexpr int srcbit[11] = {10, 8, 9, 6, 7, 4, 5, 2, 3, 0, 1};
    static PV_HD int pad(int e) { return e + (e >> 4) + (e >> 8); }
    static constexpr int LDS_CF = 2048 + 128 + 8 + 8;
};

// 2048 complex points (fft 4096) on TWO waves: 128 lanes x 16 elements, so that a lane needs the registers of the
// 1024-point transform (four waves per SIMD instead of two) and a frame's latency is shared by twice the waves.  Same
// stages, same arithmetic, same leaf permutation as WF<2048>; the passes are separated by workgroup barriers.
// (pass 0 keeps element bit 3 in a register without using it: stages 0 and 1 need three register bits)
struct WF2048S {
    static constexpr int LB = 7, LANES = 128;
    static constexpr int N_C = 2048, LOG = 11, R = 16, RB = 4, NSTAGE = 6, NPASS = 3;
    static constexpr int st_radix[6] = {2, 4, 4, 4, 4, 4};
    static constexpr int st_bit[6] = {0, 1, 3, 5, 7, 9};
    static constexpr int st_pass[6] = {0, 0, 1, 1, 2, 2};
    static constexpr int regpos[4][5] = {{0, 1, 2, 3, 0}, {3, 4, 5, 6, 0}, {7, 8, 9, 10, 0}, {0, 0, 0, 0, 0}};
    static constexpr int lanepos[4][7] = {{4, 5, 6, 7, 8, 9, 10}, {0, 1, 2, 7, 8, 9, 10}, {0, 1, 2, 3, 4, 5, 6}, {0, 0, 0, 0, 0, 0, 0}};
    static constexpr int srcbit[11] = {10, 8, 9, 6, 7, 4, 5, 2, 3, 0, 1};
    // (one slot of padding per 32: of the e + (e >> a) [+ (e >> b)] family the fewest LDS bank conflicts over the three
    // passes' stores and loads and the split's natural-order reads -- 1.37 per access on average, 2.0 with WF<2048>'s)
    static PV_HD int pad(int e) { return e + (e >> 5); }
    static constexpr int LDS_CF = 2048 + 128 + 8 + 8;
};

// ---- compile-time helpers ------------------------------------------------------------------
template <class W> constexpr int wf_reg_part(int pass, int r) {
    int e = 0;
    for (int i = 0; i < W::RB; ++i) e |= ((r >> i) & 1) << W::regpos[pass][i];
    return e;
}
template <class W> constexpr int wf_find_regbit(int pass, int ebit) {
    for (int i = 0; i < W::RB; ++i)
        if (W::regpos[pass][i] == ebit) return i;
    return -1;
}
// source (pre-permutation) index of butterfly-order index e
template <class W> constexpr int wf_src_of_const(int e) {
    int s = 0;
    for (int i = 0; i < W::LOG; ++i) s |= ((e >> i) & 1) << W::srcbit[i];
    return s;
}
template <class W> PV_HD int wf_src_of(int e) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < W::LOG; ++i) s |= ((e >> i) & 1) << W::srcbit[i];
    return s;
}
// butterfly-order index of source index s (inverse map)
template <class W> PV_HD int wf_e_of_src(int s) {
    int e = 0;
#pragma unroll
    for (int i = 0; i < W::LOG; ++i) e |= ((s >> W::srcbit[i]) & 1) << i;
    return e;
}
template <class W> PV_HD int wf_lane_part(int pass, int lane) {
    int e = 0;
#pragma unroll
    for (int i = 0; i < W::LB; ++i) e |= ((lane >> i) & 1) << W::lanepos[pass][i];
    return e;
}

// ---- butterflies (kf_bfly4 kiss_fft.c:59-103, kf_bfly2 :36-57) --------------------------------
template <bool INV> PV_HD void wf_bfly4(cf &f0, cf &f1, cf &f2, cf &f3, cf t1, cf t2, cf t3) {
    const cf s0 = wf_cmul(f1, t1);
    const cf s1 = wf_cmul(f2, t2);
    const cf s2 = wf_cmul(f3, t3);
    const cf s5 = wf_sub(f0, s1);
    const cf g0 = wf_add(f0, s1);
    const cf s3 = wf_add(s0, s2);
    const cf s4 = wf_sub(s0, s2);
    f2 = wf_sub(g0, s3);
    f0 = wf_add(g0, s3);
    if (INV) {
        f1 = cf{s5.x - s4.y, s5.y + s4.x};
        f3 = cf{s5.x + s4.y, s5.y - s4.x};
    } else {
        f1 = cf{s5.x + s4.y, s5.y - s4.x};
        f3 = cf{s5.x - s4.y, s5.y + s4.x};
    }
}
PV_HD void wf_bfly2(cf &f0, cf &f1, cf t) {
    const cf u = wf_cmul(f1, t);
    f1 = wf_sub(f0, u);
    f0 = wf_add(f0, u);
}

// one butterfly stage S on the register array (lp = this lane's part of e in the stage's pass)
template <class W, int S, bool INV> PV_HD void wf_stage(cf (&v)[W::R], int lp, const cf *__restrict__ tw) {
    constexpr int pass = W::st_pass[S], eb = W::st_bit[S], radix = W::st_radix[S];
    constexpr int m = 1 << eb;
    constexpr int fs = W::N_C / (m * radix);
    constexpr int p = wf_find_regbit<W>(pass, eb);
    static_assert(p >= 0, "stage digit must live in registers in its pass");
#pragma unroll
    for (int r = 0; r < W::R; ++r) {
        if ((r >> p) & (radix - 1)) continue;
        const int k = (lp + wf_reg_part<W>(pass, r)) & (m - 1);
        if (radix == 4) {
            wf_bfly4<INV>(v[r], v[r + (1 << p)], v[r + (2 << p)], v[r + (3 << p)], tw[k * fs], tw[2 * k * fs],
                          tw[3 * k * fs]);
        } else {
            wf_bfly2(v[r], v[r + (1 << p)], tw[k * fs]);
        }
    }
}

template <class W, int P, bool INV> PV_HD void wf_run_pass_stages(cf (&v)[W::R], int lp, const cf *__restrict__ tw) {
    if constexpr (W::NSTAGE > 0 && W::st_pass[0] == P) wf_stage<W, 0, INV>(v, lp, tw);
    if constexpr (W::NSTAGE > 1 && W::st_pass[1] == P) wf_stage<W, 1, INV>(v, lp, tw);
    if constexpr (W::NSTAGE > 2 && W::st_pass[2] == P) wf_stage<W, 2, INV>(v, lp, tw);
    if constexpr (W::NSTAGE > 3 && W::st_pass[3] == P) wf_stage<W, 3, INV>(v, lp, tw);
    if constexpr (W::NSTAGE > 4 && W::st_pass[4] == P) wf_stage<W, 4, INV>(v, lp, tw);
    if constexpr (W::NSTAGE > 5 && W::st_pass[5] == P) wf_stage<W, 5, INV>(v, lp, tw);
}

// ---- the same stages with their twiddles fetched ahead of time ---------------------------------
// A global load costs the wave a full memory round trip when it is issued right where its value is needed, so
// the kernels issue a pass's twiddle loads one pass early (WfTw lives in registers; t[i][r] is the factor that
// multiplies v[r] in the i-th stage of the pass, unused slots are never materialised).
template <class W> struct WfTw {
    cf t[3][W::R];
};
template <class W, int S> PV_HD void wf_stage_load_tw(cf (&t)[W::R], int lp, const cf *__restrict__ tw) {
    constexpr int pass = W::st_pass[S], eb = W::st_bit[S], radix = W::st_radix[S];
    constexpr int m = 1 << eb;
    constexpr int fs = W::N_C / (m * radix);
    constexpr int p = wf_find_regbit<W>(pass, eb);
#pragma unroll
    for (int r = 0; r < W::R; ++r) {
        if ((r >> p) & (radix - 1)) continue;
        const int k = (lp + wf_reg_part<W>(pass, r)) & (m - 1);
        t[r + (1 << p)] = tw[k * fs];
        if (radix == 4) {
            t[r + (2 << p)] = tw[2 * k * fs];
            t[r + (3 << p)] = tw[3 * k * fs];
        }
    }
}
template <class W, int S, bool INV> PV_HD void wf_stage_apply(cf (&v)[W::R], const cf (&t)[W::R]) {
    constexpr int pass = W::st_pass[S], eb = W::st_bit[S], radix = W::st_radix[S];
    constexpr int p = wf_find_regbit<W>(pass, eb);
#pragma unroll
    for (int r = 0; r < W::R; ++r) {
        if ((r >> p) & (radix - 1)) continue;
        if (radix == 4) {
            wf_bfly4<INV>(v[r], v[r + (1 << p)], v[r + (2 << p)], v[r + (3 << p)], t[r + (1 << p)], t[r + (2 << p)],
                          t[r + (3 << p)]);
        } else {
            wf_bfly2(v[r], v[r + (1 << p)], t[r + (1 << p)]);
        }
    }
}
template <class W, int P> constexpr int wf_first_stage() {
    for (int s = 0; s < W::NSTAGE; ++s)
        if (W::st_pass[s] == P) return s;
    return 0;
}
// ---- lane-major twiddle tables -------------------------------------------------------------------
// A lane-dependent twiddle tw[q k fs], k = (lane part + register part) mod m, is a gather that touches up to
// 48 cache lines per load instruction.  The host lays the values out once per (stage, butterfly, q) "slot" and
// lane, two slots to a 16-byte entry, so that a pass fetches its twiddles with a few contiguous 16-byte loads:
//   table[(wf_pass_base4<W, P>() + j) * 64 + lane] = { slot 2j of pass P, slot 2j + 1 }        (cf2, 16 bytes)
// Butterflies of a stage whose k agree (k ignores the register bits above the stage's m) share their slots.
template <class W> constexpr int wf_stage_first_reg(int s, int r) { // first butterfly of stage s with r's k
    const int pass = W::st_pass[s], eb = W::st_bit[s], radix = W::st_radix[s], m = 1 << eb;
    const int p = wf_find_regbit<W>(pass, eb);
    const int rk = wf_reg_part<W>(pass, r) & (m - 1);
    for (int r2 = 0; r2 < r; ++r2) {
        if ((r2 >> p) & (radix - 1)) continue;
        if ((wf_reg_part<W>(pass, r2) & (m - 1)) == rk) return r2;
    }
    return r;
}
// slot (within its pass) of factor q of the butterfly whose first register is r in stage s; slots_of_pass when
// called with s == NSTAGE
template <class W> constexpr int wf_slot_in_pass(int pass, int s_want, int r_want, int q) {
    int slot = 0;
    for (int s = 0; s < W::NSTAGE; ++s) {
        if (W::st_pass[s] != pass) continue;
        const int eb = W::st_bit[s], radix = W::st_radix[s];
        const int p = wf_find_regbit<W>(pass, eb);
        for (int r = 0; r < W::R; ++r) {
            if ((r >> p) & (radix - 1)) continue;
            const int first = wf_stage_first_reg<W>(s, r);
            if (s == s_want && r == r_want) return first == r ? slot + (q - 1) : wf_slot_in_pass<W>(pass, s, first, q);
            if (first == r) slot += radix - 1;
        }
    }
    return slot;
}
template <class W> constexpr int wf_pass_slots(int pass) { return wf_slot_in_pass<W>(pass, W::NSTAGE, 0, 0); }
template <class W> constexpr int wf_pass_entries(int pass) { return (wf_pass_slots<W>(pass) + 1) / 2; } // float4s
template <class W, int P> constexpr int wf_pass_base4() {
    int b = 0;
    for (int q = 1; q < P; ++q) b += wf_pass_entries<W>(q);
    return b;
}
template <class W> constexpr int wf_lane_table_entries() {
    int n = 0;
    for (int q = 1; q < W::NPASS; ++q) n += wf_pass_entries<W>(q);
    return n;
}

// host: fill out[(entry * 64 + lane) * 2 + (slot & 1)] for passes 1 and 2 (out: 2 * entries * 64 values)
template <class W> inline void wf_build_lane_table(const cf *tw, cf *out) {
    for (int i = 0; i < 2 * wf_lane_table_entries<W>() * W::LANES; ++i) out[i] = cf{0.f, 0.f};
    for (int pass = 1; pass < W::NPASS; ++pass) {
        int base4 = 0;
        for (int q = 1; q < pass; ++q) base4 += wf_pass_entries<W>(q);
        for (int s = 0; s < W::NSTAGE; ++s) {
            if (W::st_pass[s] != pass) continue;
            const int eb = W::st_bit[s], radix = W::st_radix[s], m = 1 << eb;
            const int fs = W::N_C / (m * radix);
            const int p = wf_find_regbit<W>(pass, eb);
            for (int r = 0; r < W::R; ++r) {
                if ((r >> p) & (radix - 1)) continue;
                if (wf_stage_first_reg<W>(s, r) != r) continue;
                for (int q = 1; q < radix; ++q) {
                    const int slot = wf_slot_in_pass<W>(pass, s, r, q);
                    for (int lane = 0; lane < W::LANES; ++lane) {
                        int lp = 0;
                        for (int i = 0; i < W::LB; ++i) lp |= ((lane >> i) & 1) << W::lanepos[pass][i];
                        const int k = (lp + wf_reg_part<W>(pass, r)) & (m - 1);
                        out[((base4 + slot / 2) * W::LANES + lane) * 2 + (slot & 1)] = tw[q * k * fs];
                    }
                }
            }
        }
    }
}

// the pass's entries (16 bytes each, contiguous across the wave), then the factors picked out of them
struct alignas(16) cf2 {
    cf a, b;
};
template <class W, int P> struct WfTwRaw {
    cf2 e[wf_pass_entries<W>(P)];
};
template <class W, int P> PV_HD void wf_fetch_pass_tw(WfTwRaw<W, P> &raw, int lane, const cf2 *__restrict__ table) {
    constexpr int NE = wf_pass_entries<W>(P), B4 = wf_pass_base4<W, P>();
#pragma unroll
    for (int j = 0; j < NE; ++j) raw.e[j] = table[(B4 + j) * W::LANES + lane];
}
// (register and factor indices are template parameters so that every slot number is a compile-time constant)
template <class W, int P, int S, int RR, int Q> PV_HD void wf_unpack_one_tw(cf (&t)[W::R], const WfTwRaw<W, P> &raw) {
    constexpr int eb = W::st_bit[S], radix = W::st_radix[S];
    constexpr int p = wf_find_regbit<W>(P, eb);
    if constexpr (((RR >> p) & (radix - 1)) == 0 && Q < radix) {
        constexpr int slot = wf_slot_in_pass<W>(P, S, RR, Q);
        t[RR + (Q << p)] = (slot & 1) ? raw.e[slot / 2].b : raw.e[slot / 2].a;
    }
}
template <class W, int P, int S, int... I>
PV_HD void wf_unpack_stage_seq(cf (&t)[W::R], const WfTwRaw<W, P> &raw, std::integer_sequence<int, I...>) {
    (wf_unpack_one_tw<W, P, S, I / 3, 1 + I % 3>(t, raw), ...);
}
template <class W, int P, int S> PV_HD void wf_unpack_stage_tw(cf (&t)[W::R], const WfTwRaw<W, P> &raw) {
    wf_unpack_stage_seq<W, P, S>(t, raw, std::make_integer_sequence<int, 3 * W::R>{});
}
template <class W, int P> PV_HD void wf_unpack_pass_tw(WfTw<W> &T, const WfTwRaw<W, P> &raw) {
    constexpr int F = wf_first_stage<W, P>();
    if constexpr (F + 0 < W::NSTAGE && W::st_pass[F + 0] == P) wf_unpack_stage_tw<W, P, F + 0>(T.t[0], raw);
    if constexpr (F + 1 < W::NSTAGE && W::st_pass[F + 1] == P) wf_unpack_stage_tw<W, P, F + 1>(T.t[1], raw);
    if constexpr (F + 2 < W::NSTAGE && W::st_pass[F + 2] == P) wf_unpack_stage_tw<W, P, F + 2>(T.t[2], raw);
}

template <class W, int P> PV_HD void wf_load_pass_tw(WfTw<W> &T, int lane, const cf *__restrict__ tw) {
    const int lp = wf_lane_part<W>(P, lane);
    constexpr int F = wf_first_stage<W, P>();
    if constexpr (F + 0 < W::NSTAGE && W::st_pass[F + 0] == P) wf_stage_load_tw<W, F + 0>(T.t[0], lp, tw);
    if constexpr (F + 1 < W::NSTAGE && W::st_pass[F + 1] == P) wf_stage_load_tw<W, F + 1>(T.t[1], lp, tw);
    if constexpr (F + 2 < W::NSTAGE && W::st_pass[F + 2] == P) wf_stage_load_tw<W, F + 2>(T.t[2], lp, tw);
}
template <class W, int P, bool INV> PV_HD void wf_apply_pass_stages(cf (&v)[W::R], const WfTw<W> &T) {
    constexpr int F = wf_first_stage<W, P>();
    if constexpr (F + 0 < W::NSTAGE && W::st_pass[F + 0] == P) wf_stage_apply<W, F + 0, INV>(v, T.t[0]);
    if constexpr (F + 1 < W::NSTAGE && W::st_pass[F + 1] == P) wf_stage_apply<W, F + 1, INV>(v, T.t[1]);
    if constexpr (F + 2 < W::NSTAGE && W::st_pass[F + 2] == P) wf_stage_apply<W, F + 2, INV>(v, T.t[2]);
}

// ---- free-form arithmetic (PV_ARITH_FAST: synthesis side only, include/audiomod_pv.h) -----------------------------
// Same transform, same stage structure and data layout, but the operation order is no longer kissfft's: complex
// products are two multiplies + two fma, and a stage whose twiddle index k lives entirely in REGISTER bits of its pass
// (pass 0: k is a compile-time constant per butterfly) multiplies by literals -- nothing for k = 0, a swap for quarter
// turns, two adds + two multiplies for eighth turns -- instead of by twiddles fetched per lane.  For 1024 points that
// turns pass 0 into a 16-point transform with constant factors (24 fewer registers, 12 fewer loads, ~110 fewer
// instructions per lane); results differ from the exact variant by rounding only.
PV_HD cf wf_cmul_fma(cf a, cf b) {
    cf m;
    m.x = __builtin_fmaf(a.x, b.x, -(a.y * b.y));
    m.y = __builtin_fmaf(a.x, b.y, a.y * b.x);
    return m;
}
// cos / sin of 2 pi j / 32 (float of the double value, like kissfft's table entries)
constexpr float kWfCos32[32] = {1.0f, 0.9807852506637573f, 0.9238795042037964f, 0.8314695954322815f, 0.7071067690849304f, 0.5555702447891235f, 0.3826834261417389f, 0.19509032368659973f, 0.0f, -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f, -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f, -1.0f, -0.9807852506637573f, -0.9238795042037964f, -0.8314695954322815f, -0.7071067690849304f, -0.5555702447891235f, -0.3826834261417389f, -0.19509032368659973f, 0.0f, 0.19509032368659973f, 0.3826834261417389f, 0.5555702447891235f, 0.7071067690849304f, 0.8314695954322815f, 0.9238795042037964f, 0.9807852506637573f};
constexpr float kWfSin32[32] = {0.0f, 0.19509032368659973f, 0.3826834261417389f, 0.5555702447891235f, 0.7071067690849304f, 0.8314695954322815f, 0.9238795042037964f, 0.9807852506637573f, 1.0f, 0.9807852506637573f, 0.9238795042037964f, 0.8314695954322815f, 0.7071067690849304f, 0.5555702447891235f, 0.3826834261417389f, 0.19509032368659973f, 0.0f, -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f, -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f, -1.0f, -0.9807852506637573f, -0.9238795042037964f, -0.8314695954322815f, -0.7071067690849304f, -0.5555702447891235f, -0.3826834261417389f, -0.19509032368659973f};
// a * exp(-+ 2 pi i J / 32)  (forward: minus; INV: plus)
template <int J, bool INV> PV_HD cf wf_cmul_w32(cf a) {
    constexpr int j = J & 31;
    if constexpr (j == 0) {
        return a;
    } else if constexpr (j == 16) {
        return cf{-a.x, -a.y};
    } else if constexpr (j == 8) { // times (0, +-1)
        return INV ? cf{-a.y, a.x} : cf{a.y, -a.x};
    } else if constexpr (j == 24) {
        return INV ? cf{a.y, -a.x} : cf{-a.y, a.x};
    } else {
        constexpr float c = kWfCos32[j];
        constexpr float sI = INV ? kWfSin32[j] : -kWfSin32[j];
        if constexpr ((j & 3) == 0) { // eighth turns: |c| == |s|
            constexpr float h = 0.7071067690849304f;
            constexpr float sc = c > 0.f ? 1.f : -1.f, ss = sI > 0.f ? 1.f : -1.f;
            return cf{(sc * a.x - ss * a.y) * h, (ss * a.x + sc * a.y) * h};
        } else {
            return cf{__builtin_fmaf(a.x, c, -(a.y * sI)), __builtin_fmaf(a.x, sI, a.y * c)};
        }
    }
}
// radix-4 / radix-2 butterflies on already-multiplied inputs
template <bool INV> PV_HD void wf_bfly4_core(cf &f0, cf &f1, cf &f2, cf &f3, const cf s0, const cf s1, const cf s2) {
    const cf s5 = wf_sub(f0, s1);
    const cf g0 = wf_add(f0, s1);
    const cf s3 = wf_add(s0, s2);
    const cf s4 = wf_sub(s0, s2);
    f2 = wf_sub(g0, s3);
    f0 = wf_add(g0, s3);
    if (INV) {
        f1 = cf{s5.x - s4.y, s5.y + s4.x};
        f3 = cf{s5.x + s4.y, s5.y - s4.x};
    } else {
        f1 = cf{s5.x + s4.y, s5.y - s4.x};
        f3 = cf{s5.x - s4.y, s5.y + s4.x};
    }
}
// does stage S's twiddle index depend on register bits only?
template <class W, int S> constexpr bool wf_stage_k_in_regs() {
    const int pass = W::st_pass[S], eb = W::st_bit[S];
    for (int b = 0; b < eb; ++b)
        if (wf_find_regbit<W>(pass, b) < 0) return false;
    return (1 << eb) * W::st_radix[S] <= 32;
}
template <class W, int S, bool INV, int RR> PV_HD void wf_stage_const_one(cf (&v)[W::R]) {
    constexpr int pass = W::st_pass[S], eb = W::st_bit[S], radix = W::st_radix[S], m = 1 << eb;
    constexpr int p = wf_find_regbit<W>(pass, eb);
    if constexpr (((RR >> p) & (radix - 1)) == 0) {
        constexpr int k = wf_reg_part<W>(pass, RR) & (m - 1);
        constexpr int u = 32 / (m * radix); // twiddle q of this butterfly = W32^(k q u)
        if constexpr (radix == 4) {
            const cf s0 = wf_cmul_w32<k * u, INV>(v[RR + (1 << p)]);
            const cf s1 = wf_cmul_w32<2 * k * u, INV>(v[RR + (2 << p)]);
            const cf s2 = wf_cmul_w32<3 * k * u, INV>(v[RR + (3 << p)]);
            wf_bfly4_core<INV>(v[RR], v[RR + (1 << p)], v[RR + (2 << p)], v[RR + (3 << p)], s0, s1, s2);
        } else {
            const cf t = wf_cmul_w32<k * u, INV>(v[RR + (1 << p)]);
            v[RR + (1 << p)] = wf_sub(v[RR], t);
            v[RR] = wf_add(v[RR], t);
        }
    }
}
template <class W, int S, bool INV, int... I>
PV_HD void wf_stage_const_seq(cf (&v)[W::R], std::integer_sequence<int, I...>) {
    (wf_stage_const_one<W, S, INV, I>(v), ...);
}
// stage S with free-form arithmetic: literal twiddles where k is in registers, fetched ones (t) with fma otherwise
template <class W, int S, bool INV> PV_HD void wf_stage_apply_fast(cf (&v)[W::R], const cf (&t)[W::R]) {
    if constexpr (wf_stage_k_in_regs<W, S>()) {
        wf_stage_const_seq<W, S, INV>(v, std::make_integer_sequence<int, W::R>{});
    } else {
        constexpr int pass = W::st_pass[S], eb = W::st_bit[S], radix = W::st_radix[S];
        constexpr int p = wf_find_regbit<W>(pass, eb);
#pragma unroll
        for (int r = 0; r < W::R; ++r) {
            if ((r >> p) & (radix - 1)) continue;
            if (radix == 4) {
                const cf s0 = wf_cmul_fma(v[r + (1 << p)], t[r + (1 << p)]);
                const cf s1 = wf_cmul_fma(v[r + (2 << p)], t[r + (2 << p)]);
                const cf s2 = wf_cmul_fma(v[r + (3 << p)], t[r + (3 << p)]);
                wf_bfly4_core<INV>(v[r], v[r + (1 << p)], v[r + (2 << p)], v[r + (3 << p)], s0, s1, s2);
            } else {
                const cf u = wf_cmul_fma(v[r + (1 << p)], t[r + (1 << p)]);
                v[r + (1 << p)] = wf_sub(v[r], u);
                v[r] = wf_add(v[r], u);
            }
        }
    }
}
template <class W, int P, bool INV> PV_HD void wf_apply_pass_stages_fast(cf (&v)[W::R], const WfTw<W> &T) {
    constexpr int F = wf_first_stage<W, P>();
    if constexpr (F + 0 < W::NSTAGE && W::st_pass[F + 0] == P) wf_stage_apply_fast<W, F + 0, INV>(v, T.t[0]);
    if constexpr (F + 1 < W::NSTAGE && W::st_pass[F + 1] == P) wf_stage_apply_fast<W, F + 1, INV>(v, T.t[1]);
    if constexpr (F + 2 < W::NSTAGE && W::st_pass[F + 2] == P) wf_stage_apply_fast<W, F + 2, INV>(v, T.t[2]);
}
// true when no stage of pass P needs fetched twiddles in the fast variant
template <class W, int P> constexpr bool wf_pass_all_const() {
    for (int s = 0; s < W::NSTAGE; ++s)
        if (W::st_pass[s] == P) {
            const int eb = W::st_bit[s];
            for (int b = 0; b < eb; ++b)
                if (wf_find_regbit<W>(P, b) < 0) return false;
            if ((1 << eb) * W::st_radix[s] > 32) return false;
        }
    return true;
}

template <class W, int P> PV_HD void wf_store(cf *lds, const cf (&v)[W::R], int lp) {
#pragma unroll
    for (int r = 0; r < W::R; ++r) lds[W::pad(lp | wf_reg_part<W>(P, r))] = v[r];
}
template <class W, int P> PV_HD void wf_load(const cf *lds, cf (&v)[W::R], int lp) {
#pragma unroll
    for (int r = 0; r < W::R; ++r) v[r] = lds[W::pad(lp | wf_reg_part<W>(P, r))];
}

// Pass P of the FFT for one lane.  P == 0 expects v already loaded in pass-0 layout (element
// e = lane_part<0>(lane) | reg_part<0>(r)); passes 1 and 2 load their layout from LDS first.  Every pass
// leaves its results in LDS at pad(e); after pass 2 the LDS region holds the transform in natural order.
// The caller separates passes by a wave-level ordering point (all lanes finish pass P before pass P+1).
template <class W, int P, bool INV> PV_HD void wf_fft_pass(cf (&v)[W::R], int lane, cf *lds, const cf *__restrict__ tw) {
    const int lp = wf_lane_part<W>(P, lane);
    if (P > 0) wf_load<W, P>(lds, v, lp);
    wf_run_pass_stages<W, P, INV>(v, lp, tw);
    wf_store<W, P>(lds, v, lp);
}

// The same pass with the twiddles already in registers (wf_load_pass_tw<W, P> issued earlier by the caller).
template <class W, int P, bool INV> PV_HD void wf_fft_pass_tw(cf (&v)[W::R], int lane, cf *lds, const WfTw<W> &T) {
    const int lp = wf_lane_part<W>(P, lane);
    if (P > 0) wf_load<W, P>(lds, v, lp);
    wf_apply_pass_stages<W, P, INV>(v, T);
    wf_store<W, P>(lds, v, lp);
}

} // namespace pv
