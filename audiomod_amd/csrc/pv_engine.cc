// pv_engine.cc -- host engine + C ABI (include/audiomod_pv.h) of the MI355X phase-vocoder.
//
// The host stages overlapping STFT frames across channels / blocks / streams and drives the four
// gfx950 kernels of pv_kernels.hip chunk by chunk.  All data-independent decisions (hop sizes,
// per-slice shift increments, output counts, OLA tile geometry) come from the integer planner in
// pv_plan.cc; the device never needs a host round trip inside a chunk.
//
// There is NO CPU fallback: without a gfx950 device every constructor returns PV_ERR_NO_DEVICE.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "audiomod_pv.h"
#include "pv_kernels.h"
#include "pv_wavefft.h"
#include "pv_plan.h"

namespace pv {

static thread_local std::string g_last_error;

static int hip_fail(hipError_t e, const char *what, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed at pv_engine.cc:%d: %s", what, line, hipGetErrorString(e));
    g_last_error = buf;
    return PV_ERR_HIP;
}
// inside Core::launch_chunk (returns nothing): the first failing call of a launch sequence is remembered with its line
// and reported by the entry point that enqueued it (pv_batch_run / pv_feed)
#define HIPV(call)                                                           \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess && launch_err == hipSuccess) launch_err = e__, launch_err_line = __LINE__; \
    } while (0)
#define HIPC(call)                                                 \
    do {                                                           \
        hipError_t e__ = (call);                                   \
        if (e__ != hipSuccess) return hip_fail(e__, #call, __LINE__); \
    } while (0)

static int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static int next_pow2_i(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

static int count_gfx950() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count) {
        release();
        n = count;
        HIPC(hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)));
        // Every device buffer starts as zeros (a creation-time cost only).  The kernels read nothing they have not
        // written except slots whose values are provably unused (records and rotations of lanes beyond a step's peak
        // count), but "provably unused" deserves a belt: with this, what such a slot holds cannot depend on what the
        // memory held before -- round 2 saw one bit-identity test fail twice, with identical garbage in one output
        // sample, on what was probably one box of the pool, and never again.
#ifdef PV_POISON // debugging build: NaN / -1 patterns, so that relying on these zeros shows (pv_kernels.hip PV_POISON)
        HIPC(hipMemset(p, 0xFF, (count ? count : 1) * sizeof(T)));
#else
        HIPC(hipMemset(p, 0, (count ? count : 1) * sizeof(T)));
#endif
        return PV_OK;
    }
    int upload(const std::vector<T> &v) {
        int st = alloc(v.size());
        if (st != PV_OK) return st;
        if (!v.empty()) HIPC(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return PV_OK;
    }
};

template <typename T> struct PinBuf {
    T *p = nullptr;
    size_t n = 0;
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
    int alloc(size_t count) {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = count;
        HIPC(hipHostMalloc((void **)&p, (count ? count : 1) * sizeof(T), hipHostMallocDefault));
        return PV_OK;
    }
};

// ------------------------------------------------------------------------------------------
// Core: device-resident tables, intermediates and per-stream state for S streams x C channels.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Host plan of the fused synthesis + overlap-add kernel: for every slice, where its frame lands in the rings,
// the window-sum denominators of the samples it finalises and the resampler's positions for the outputs its wave
// emits.  All of it is data-independent and the same for every row, so it is computed once per slice here
// (64-bit divisions, float window sums in the reference's order) and the kernel only looks things up.
// ------------------------------------------------------------------------------------------
struct ChainBuilder {
    const Derived &d;
    int AR, smask;
    struct Live {
        int64_t P;
        int32_t flags;
    };
    std::deque<Live> live; // frames that may still cover samples not yet finalised, oldest first
    bool any_upper_skip = false;
    int64_t launch_k0 = 0; // first output of the current launch (begin_launch)
    std::vector<ChainSlice> launch_cs; // the current launch's slices, in order (end_launch turns them into run lists)
    std::vector<int64_t> launch_P;

    ChainBuilder(const Derived &dd, int ar, int mask) : d(dd), AR(ar), smask(mask) {}

    static int32_t pmod(int64_t v, int m) {
        int64_t r = v % m;
        return (int32_t)(r < 0 ? r + m : r);
    }
    // window-sum denominator of OLA sample n at the moment writeSlice divides (channelinfo.cc:108 seeds [0] with
    // 1; synthesiseSlice :1073 adds w[i] * float(area * 1.5) per frame, oldest frame first)
    float denominator(int64_t n, bool upper) const {
        float acc = n == 0 ? 1.f : 0.f;
        for (const Live &f : live) {
            if (upper && (f.flags & kSliceUpperChannelsSkip)) continue;
            const int64_t off = n - f.P;
            if (off >= 0 && off < d.N) acc += d.window[(size_t)off] * d.win_gain;
        }
        return acc;
    }
    // out_limit: outputs at or beyond it are not written (the CLI truncates to the input length).
    void add(const SliceRec &r, int64_t out_limit, std::vector<float> &wden, std::vector<float> &wden_hi) {
        ChainSlice c{};
        c.tl = (int32_t)launch_cs.size();
        c.acc_pos = pmod(r.P, AR);
        c.str_pos = (int32_t)(r.P & (int64_t)smask);
        c.adv = r.adv;
        c.flags = r.flags;
        if (r.flags & kSliceUpperChannelsSkip) any_upper_skip = true;
        live.push_back(Live{r.P, r.flags});
        // denominators by ring quads: entry 0 belongs to sample P - (P mod 4); entries outside [P, P + adv) are
        // never used (1.0); every slice starts on a 16-byte boundary
        while (wden.size() & 3) wden.push_back(1.f), wden_hi.push_back(1.f);
        c.wden_off = (int32_t)wden.size();
        const int lead = c.acc_pos & 3;
        for (int i = -lead; i < ((r.adv + lead + 3) & ~3) - lead; ++i) {
            const bool in = i >= 0 && i < r.adv;
            wden.push_back(in ? denominator(r.P + i, false) : 1.f);
            wden_hi.push_back(in && any_upper_skip ? denominator(r.P + i, true) : wden.back());
        }
        const int64_t Pn = r.P + r.adv;
        while (!live.empty() && live.front().P + d.N <= Pn) live.pop_front();
        if (!d.resample) { // the finalised samples are the outputs
            int64_t lo = r.K0, hi = r.K0 + r.cnt;
            if (hi > out_limit) hi = out_limit;
            if (lo > hi) lo = hi;
            c.k_off = (int32_t)(lo - launch_k0);
            c.kcnt = (int32_t)(hi - lo);
        }
        launch_cs.push_back(c);
        launch_P.push_back(r.P);
    }
    // the first output of the launch, relative to which k_off counts
    int64_t begin_launch(const SliceRec &first) {
        launch_k0 = first.K0;
        launch_cs.clear();
        launch_P.clear();
        return launch_k0;
    }
    // Ends the launch: its slices become the lists of up to `runs_wanted` runs (one workgroup each per row).  Run 0
    // continues from the row's carried accumulator.  A later run starts from an empty one and first re-adds the
    // frames before its range whose tails reach into it -- every frame j with P_j + N > P_start, copied in front of
    // its list with flag bit 1 (rebuilt, not emitted): from its first own sample on, its accumulator then holds the
    // same sums in the same order as the sequential walk.  The number of runs is cut down until that warm-up is at
    // most half a run (and falls back to one run where it would reach before the launch).
    // Appends the lists to `cs` and runs + 1 offsets (relative to the launch's first entry in `cs`) to `run_off`;
    // returns the number of runs.
    int end_launch(int runs_wanted, std::vector<ChainSlice> &cs, std::vector<int32_t> &run_off) {
        const int Tn = (int)launch_cs.size();
        int R = runs_wanted < 1 ? 1 : runs_wanted;
        std::vector<int> start, warm;
        for (; R > 1; R = R / 2) {
            const int L = (Tn + R - 1) / R;
            bool ok = L >= 2;
            start.clear();
            warm.clear();
            for (int r = 1; ok && r < R; ++r) {
                const int a = r * L;
                if (a >= Tn) {
                    ok = false;
                    break;
                }
                int w = a;
                while (w > 0 && launch_P[(size_t)(w - 1)] + d.N > launch_P[(size_t)a]) --w;
                if (w == 0 && launch_P[0] + d.N > launch_P[(size_t)a]) ok = false; // would need frames of the launch before
                if (2 * (a - w) > L) ok = false;
                start.push_back(a);
                warm.push_back(w);
            }
            if (ok) break;
        }
        const size_t base = cs.size();
        if (R <= 1) {
            run_off.push_back(0);
            cs.insert(cs.end(), launch_cs.begin(), launch_cs.end());
            run_off.push_back((int32_t)(cs.size() - base));
            return 1;
        }
        const int L = (Tn + R - 1) / R;
        for (int r = 0; r < R; ++r) {
            run_off.push_back((int32_t)(cs.size() - base));
            const int a = r * L, b = (r + 1) * L < Tn ? (r + 1) * L : Tn;
            if (r > 0)
                for (int j = warm[(size_t)(r - 1)]; j < a; ++j) {
                    ChainSlice w = launch_cs[(size_t)j];
                    w.flags |= 2;
                    cs.push_back(w);
                }
            for (int j = a; j < b; ++j) cs.push_back(launch_cs[(size_t)j]);
        }
        run_off.push_back((int32_t)(cs.size() - base));
        return R;
    }
};

// what one launch of the fused synthesis + overlap-add kernel needs from the host plan
struct ChainLaunch {
    const ChainSlice *slices; // the launch's run lists (device)
    const int32_t *run_off;   // [runs + 1]
    int runs;
    const float *wden, *wden_hi;
    float *out;               // the row-0 address of the launch's first output
    // resampling configurations: the second kernel's tiles for the outputs this launch completes
    const ResTile *res_tiles;
    const uint2 *res_otab;
    int res_ntiles;
    // Batch path: the resampling kernel (arithmetic-bound) runs on its own stream beside what the main stream does
    // next (the following chunk's analysis and match, memory- and latency-bound).  ev_fused: recorded behind this
    // launch's fused kernel; ev_res: recorded behind its resampling; ev_ring_free: the resampling that must have
    // finished before this launch's fused kernel may overwrite the stream ring (two launches back).
    hipStream_t res_stream;
    hipEvent_t ev_fused, ev_res, ev_ring_free;
    bool late_chain; // three-stage order: the rotation chain is handed over by a separate call (part 5)
};

struct Core {
    Derived d;
    int device = 0, S = 0, C = 0, rows = 0;
    int Tc = 0, TR = 0, FR = 0, HP = 0, pkmax = 0, PKP = 0, lookback = 0;
    bool pipelined_planes = false; // set before init() by the batch engine
    int ola_lds_floats = 0;
    int otab_off = 0, wacc_pitch = 0; // layout of one tile's row of host-planned values (pv_kernels.h OlaArgs)
    DevTables tb{};
    DevBuf<int32_t> perm, iperm;
    DevBuf<float2> tw_fwd, tw_inv, st_fwd, st_inv;
    DevBuf<float4> twl_fwd, twl_inv;
    DevBuf<float> window, window_sh, sinc;
    DevBuf<float4> tab4;
    DevBuf<float> mag, phase, outphase, frames, rot;
    DevBuf<float> cmag, cphase; // vocoder: carrier planes [TR][HP]
    DevBuf<uint16_t> peaks;
    DevBuf<int32_t> npk, modes;
    DevBuf<PeakRec> recs;
    // persistent per-row phase state
    DevBuf<float> st_pp, st_po, st_rot;
    DevBuf<int32_t> st_kind;
    // Fused synthesis + overlap-add ("chain", pv_kernels.h ChainArgs): the reference's accumulators live in LDS and
    // their images are carried here between launches.  On by default (AUDIOMOD_PV_FUSED=0 selects the frame ring +
    // tile kernel instead, which cannot represent dropped slices).
    bool use_chain = false;
    // The fused kernel runs one workgroup per (row, run); a batch with fewer than 256 rows splits each row's slices
    // of a launch into runs (ChainBuilder::end_launch) to fill the chip.  Very small batches stay on the tile path
    // (frames through HBM, thousands of small workgroups) unless their plan contains dropped slices (only the fused
    // path's accumulator can represent them) or AUDIOMOD_PV_FUSED=2 asks for it; the streaming engine always takes
    // the fused path (it must follow whatever the caller's call sizes lead to).
    bool chain_required = false; // set before init()
    // (measured, 128 / 64 / 32 rows: fused + runs 31.9 / 20.0 / 12.2 ms per step against the tile path's 31.2 / 17.2 /
    // 12.2 -- below a full chip's worth of rows the rotation chain's serial latency dominates either way, and the
    // tile path's kernels need no warm-up)
    static constexpr int kChainMinRows = 192;
    bool fast_arith = false;  // set before init() by the batch engine (audiomod_pv.h PV_ARITH_FAST); only the fused
                              // wave-FFT path has the fast kernels, everything else computes exactly either way
    bool fuse_phase = false;  // single-stream engine: match kernel and rotation chain in one launch (set by pv_create)
    bool split_analysis = false; // 4096-point frames: the analysis kernel with a frame on two waves (pv_analyze_split_kernel)
    bool ahead = false;       // three-stage order with the analysis one chunk further ahead (pv_batch_run): planes hold three chunks
    bool three_stage = false; // pipelined batch path: resampling of chunk i-2 between the front of i and the fused kernel of i-1
    int chain_AR = 0, chain_smask = 0, chain_waves = 0;
    int chain_max_adv = 0; // set before init(): the largest overlap-add advance the planner can emit
    DevBuf<float> st_acc, stream; // accumulator-ring images (two halves); normalised overlap-add stream rings (resampling only)
    mutable int acc_half = 0;     // which half of st_acc the next fused launch reads
#ifdef PV_DIAG
    static bool debug_stale_acc() {
        static const bool on = [] {
            const char *e = getenv("AUDIOMOD_PV_DEBUG_STALE_ACC");
            return e && atoi(e) != 0;
        }();
        return on;
    }
#endif
    mutable hipError_t launch_err = hipSuccess; // first HIP failure inside launch_chunk since take_launch_error()
    mutable int launch_err_line = 0;
    int take_launch_error() const {
        if (launch_err == hipSuccess) return PV_OK;
        const int rc = hip_fail(launch_err, "a launch / event call of Core::launch_chunk", launch_err_line);
        launch_err = hipSuccess;
        return rc;
    }
    // resample tiles (outputs [ka, kb)) of the fused path
    void build_res_tiles(int64_t ka, int64_t kb, std::vector<ResTile> &tiles, std::vector<uint2> &otab) const;
    static bool chain_wanted() {
        const char *e = getenv("AUDIOMOD_PV_FUSED");
        if (e && atoi(e) == 0) return false;
        const char *sl = getenv("AUDIOMOD_PV_STREAM_LAUNCHES"); // the opt-in one-workgroup streaming kernel chains
        return !(sl && strcmp(sl, "single") == 0);               // the separate stages' device functions
    }
    bool wave_fft() const { return d.fft.nc == 256 || d.fft.nc == 512 || d.fft.nc == 1024 || d.fft.nc == 2048; } // fft 512 ... 4096
    // PV_ARITH_FAST and a free-form fused kernel exists for this configuration (pv_kernels.hip launch_synth_chain):
    // the window-sum denominators are then uploaded as reciprocals
    bool fast_chain() const { return use_chain && fast_capable(); }
    // ... whatever the path: with PV_ARITH_FAST such a configuration ALWAYS takes the fused path (init), so that what
    // an engine computes does not depend on how many rows it was created for -- a batch, its host-staged groups and
    // the single-stream engine run the same kernels and agree bit for bit, as they do under PV_ARITH_EXACT
    bool fast_capable() const {
        if (!(fast_arith && wave_fft())) return false;
        SynthArgs sa{};
        sa.tb.nc = d.fft.nc;
        sa.do_freq_comp = d.do_freq_comp ? 1 : 0;
        sa.voc_band_len = d.vocoder ? d.voc_band_len : -1;
        sa.robotic = d.robotic ? 1 : 0;
        sa.passthru = d.constant ? 1 : 0;
        sa.whisper = d.whisper ? reinterpret_cast<const float *>(this) : nullptr; // (only tested against null)
        const bool bypass = d.robotic || d.whisper || d.constant || d.vocoder;
        sa.coremode = bypass ? 0 : ((d.cfg.coremode == 1 || d.cfg.coremode == 2) ? d.cfg.coremode : 0);
        return synth_chain_has_fast(sa);
    }

    int init(const pv_config &cfg, int dev, int nstreams, int chunk_slices);
    int reset_state(hipStream_t st);
    // tile geometry for outputs [ka, kb) given the slice table (P of slice t = slices[t - t_base].P)
    // The streaming path's single-launch kernel (every mode but the vocoders, wave-FFT sizes).  Measured SLOWER
    // than one launch per stage -- 93 vs 70 us per 480-frame stereo call: the five launches are asynchronous and
    // overlap the kernels, whereas one workgroup runs the stages' latencies back to back on one CU -- so it is
    // opt-in: AUDIOMOD_PV_STREAM_LAUNCHES=single.
    bool can_single_launch() const;
    int build_tiles(const std::vector<SliceRec> &slices, int64_t t_base, int64_t t_end, int64_t ka, int64_t kb,
                    int32_t p_index_base, std::vector<OlaTile> &tiles, std::vector<float> &wacc) const;
    // part: 0 = the whole chunk on `st`; 1 = its front (analysis, match, and the rotation chain handed to
    // st_chain); 2 = its back (synthesis, overlap-add, after waiting for the chain).  Parts 1 and 2 are the two
    // halves of the pipelined phase-locked batch path (pv_batch_run).
    void launch_chunk(const InAddr &ia, int64_t t0, int Tn, const int32_t *d_pinc, const OlaTile *d_tiles,
                      int ntiles, const int64_t *d_P, const float *d_wacc, const float *d_whisper,
                      const InAddr *carrier, float *out, int64_t out_stride_row, int64_t k_base,
                      hipStream_t st, hipEvent_t *ev /* 2*PV_NUM_KERNELS events or null */, int part = 0,
                      hipStream_t st_chain = nullptr /* the chain's own stream (with ev_match / ev_chain) */,
                      hipEvent_t ev_match = nullptr, hipEvent_t ev_chain = nullptr,
                      bool single_launch = false, // single_launch: the streaming path's one-workgroup kernel
                      const struct ChainLaunch *chain = nullptr) const; // non-null: fused synthesis + overlap-add
    // Phase-locked batch path: the rotation chain of chunk i (one workgroup per row, a few waves, pure latency:
    // it leaves 95 % of the chip idle) runs on a second HIP stream while the main stream synthesises and
    // overlap-adds chunk i-1 and analyses and matches chunk i+1.  The slice-indexed planes then hold two chunks.
    // AUDIOMOD_PV_PIPELINE=0 turns it off (one stream, chunk after chunk).
    static bool pipeline_wanted(const pv_config &cfg) {
        const char *e = getenv("AUDIOMOD_PV_PIPELINE");
        if (e && atoi(e) == 0) return false;
        const bool phase_stage = cfg.mode == PV_MODE_NORMAL_SHIFT || cfg.mode == PV_MODE_GENDER_CHANGE ||
                                 cfg.mode == PV_MODE_FORMANT_PRESERVE || cfg.mode == PV_MODE_NORMAL_STRETCH ||
                                 cfg.mode == PV_MODE_FORMANT_CEPSTRAL;
        // coremode 1 only: the coremode-0 kernel streams whole planes and merely trades places with synthesis
        // when it runs beside it (measured: 13.42 vs 13.47 Gsamples/s), coremode 2 has no phase kernel
        // ... and frames up to 2048 points: at 4096 the synthesis waves hold 240 VGPRs, two to a SIMD, and lose
        // more to the chain's waves beside them than the overlap returns (measured: 9.65 vs 9.77 Gsamples/s)
        // (AUDIOMOD_PV_PIPELINE=2 forces it for the larger frames too: for measurements)
        return phase_stage && cfg.coremode == 1 && (cfg.fftsize <= 2048 || (e && atoi(e) == 2));
    }
    bool can_overlap_chain() const { return pipelined_planes; }
};

int Core::init(const pv_config &cfg, int dev, int nstreams, int chunk_slices) {
    int st = derive(cfg, d);
    if (st != PV_OK) return st;
    if (d.N > 8192) {
        g_last_error = "fftsize above 8192 is not supported by the phase kernel";
        return PV_ERR_UNSUPPORTED;
    }
    if (count_gfx950() <= 0) {
        g_last_error = "no gfx950 (MI355X) device visible: this library has no CPU fallback";
        return PV_ERR_NO_DEVICE;
    }
    device = dev;
    HIPC(hipSetDevice(dev));
    {
        hipDeviceProp_t p;
        HIPC(hipGetDeviceProperties(&p, dev));
        if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
            g_last_error = std::string("device is ") + p.gcnArchName + ", kernels are built for gfx950 only";
            return PV_ERR_NO_DEVICE;
        }
    }
    if (!lds_starts_at_zero()) {
        g_last_error = "internal: an analysis kernel was built with static LDS (its atan2f table address assumes none)";
        return PV_ERR_HIP;
    }
    S = nstreams;
    C = cfg.channels;
    rows = S * C;
    if (rows > 65535) { // several kernels put the rows on grid.y
        g_last_error = "more than 65535 rows (streams x channels) in one batch: split it";
        return PV_ERR_UNSUPPORTED;
    }
    HP = d.hs + 8; // row pitch of the mag / phase planes (16-byte aligned rows)
    pkmax = d.hs / 3 + 2;
    PKP = (pkmax + 7) & ~7;
    // frames that can overlap one OLA tile / that must stay in the ring behind the newest slice
    const double step = d.resample ? (double)d.res_num / (double)d.res_den : 1.0;
    const int tile_span = (int)(kTileOut * step) + (d.resample ? d.filt_len : 0) + 4;
    ola_lds_floats = (tile_span + 4 + 3) & ~3; // a multiple of 4: the gather writes the tile four samples at a time
    otab_off = (ola_lds_floats + 3) & ~3;
    wacc_pitch = otab_off + 2 * kTileOut;
    lookback = (d.N + tile_span) / d.min_shift + 3;
    use_chain = chain_wanted() && (chain_required || rows >= kChainMinRows || fast_capable());
    if (const char *e = getenv("AUDIOMOD_PV_FUSED")) // =2: the fused path whatever the row count
        if (atoi(e) == 2) use_chain = true;
    if (!use_chain && (tile_span + d.N) / d.min_shift + 3 > kMaxTileFrames) {
        if (chain_wanted()) use_chain = true; // the tile path cannot hold that many frames per tile: fused after all
    }
    if (!use_chain && (tile_span + d.N) / d.min_shift + 3 > kMaxTileFrames) {
        g_last_error = "hop too small relative to the FFT size for the OLA tile";
        return PV_ERR_UNSUPPORTED;
    }
    if (use_chain) {
        // ring sizes and waves per workgroup of the chain kernel: as many waves (slices of a row in flight) as LDS
        // holds.  While a wave resamples slice t the others may finalise up to slice t + W - 1, and whole blocks of
        // 64 outputs are deferred by up to one slice (ChainBuilder), so the stream ring keeps W advances plus one
        // filter length plus the span of 64 outputs.
        if (chain_max_adv <= 0) {
            const bool fixed_shift = d.robotic || d.whisper || d.constant || d.vocoder;
            double m = fixed_shift ? (double)d.hop : (d.int_ratio ? (double)d.hop * d.hs_ratio : 2.0 * d.hop * d.hs_ratio + 1);
            chain_max_adv = (int)(m < d.N ? m + 1 : d.N);
        }
        chain_AR = d.N + 4;
        // (beside the rotation chain's kernel -- the pipelined batch path -- twelve: its six-wave workgroups then
        // find a CU's fourth wave slot and 8 KB of LDS free and run at their stand-alone speed; with fourteen or
        // sixteen they wait for a fused workgroup to finish and the overlap is gone: 54.1 vs 58.6 ms per step)
        {
            const char *e3 = getenv("AUDIOMOD_PV_THREE_STAGE");
            three_stage = pipelined_planes && d.resample && !(e3 && atoi(e3) == 0);
        }
        int wmax = d.fft.nc == 2048 ? 8 : ((pipelined_planes && !three_stage) ? 12 : 16);
        {
            // the all-modes synthesis variant is compiled for twelve waves (pv_kernels.hip pv_synth_chain_kernel)
            const char *g = getenv("AUDIOMOD_PV_SYNTH_GENERIC"); // (tests: every mode through the all-modes variant)
            const bool plain = !d.do_freq_comp && !d.vocoder && !d.robotic && !d.constant && !d.whisper &&
                               !(g && atoi(g) != 0);
            // (round 3: the FREE-FORM formant / gender kernel fits 128 registers, sixteen waves: pv_kernels.hip
            // chain_kernel_max_threads)
            const bool fc_fast = d.do_freq_comp && !d.vocoder && !d.robotic && !d.constant && !d.whisper && !(g && atoi(g) != 0) &&
                                 d.cfg.coremode == 1 && d.fft.nc == 1024 && fast_capable();
            if (!plain && !fc_fast && wmax > 12) wmax = 12;
        }
        if (const char *e = getenv("AUDIOMOD_PV_CHAIN_WAVES")) { // tuning knob: upper bound of waves per workgroup
            const int v = atoi(e);
            if (v >= 1 && v <= (d.fft.nc == 2048 ? 8 : 16)) wmax = v;
        }
        ChainArgs probe{};
        probe.AR = chain_AR;
        chain_waves = 0;
        for (int w = wmax; w >= 1; --w) {
            probe.waves = w;
            if (chain_lds_bytes(probe, wave_fft() ? d.fft.nc : 0) <= 160 * 1024 - 512) {
                chain_waves = w;
                break;
            }
        }
        if (chain_waves == 0) {
            g_last_error = "the overlap-add ring of this configuration does not fit the LDS";
            return PV_ERR_UNSUPPORTED;
        }
        // the stream ring keeps what one launch finalises plus the history the next launch's first windows reach
        // back into (one filter length), with room to spare
        chain_smask = next_pow2_i(2 * chunk_slices * chain_max_adv + 4 * d.N + 1024) - 1;
    }
    Tc = chunk_slices;
    // slice-indexed planes keep the last slice of the previous launch (pv_kernels.h); the pipelined batch path
    // has two chunks in flight (the front of chunk i+1 runs before the back of chunk i)
    {
        // Round 3: with the free-form resampling kernel the rotation chain of chunk i (0.3-0.4 ms beside it) outlasts
        // the resampling of chunk i-2 (0.28 ms) and delays the fused kernel, whose workgroups need whole CUs.  The
        // analysis of chunk i+1 therefore moves in between -- M(i) -> chain(i) | R(i-2), A(i+1), F(i-1) -- which needs
        // the planes of three chunks.  AUDIOMOD_PV_AHEAD=0: round 2's order.
        const char *ea = getenv("AUDIOMOD_PV_AHEAD"), *er = getenv("AUDIOMOD_PV_RES_STREAM");
        ahead = use_chain && three_stage && wave_fft() && !(ea && atoi(ea) == 0) && !(er && atoi(er) != 0);
    }
    TR = (nstreams > 0 && pipelined_planes) ? (ahead ? 3 : 2) * Tc + 1 : Tc + 1;
    FR = next_pow2_i(Tc + lookback + 1);

    // tables
    std::vector<int32_t> ip(d.fft.nc);
    for (int j = 0; j < d.fft.nc; ++j) ip[d.fft.perm[j]] = j;
    if ((st = perm.upload(d.fft.perm)) != PV_OK) return st;
    if ((st = iperm.upload(ip)) != PV_OK) return st;
    auto up2 = [&](DevBuf<float2> &b, const std::vector<cpx> &v) -> int {
        std::vector<float2> t(v.size());
        for (size_t i = 0; i < v.size(); ++i) t[i] = make_float2(v[i].r, v[i].i);
        return b.upload(t);
    };
    if ((st = up2(tw_fwd, d.fft.tw_fwd)) != PV_OK) return st;
    if ((st = up2(tw_inv, d.fft.tw_inv)) != PV_OK) return st;
    if ((st = up2(st_fwd, d.fft.st_fwd)) != PV_OK) return st;
    if (wave_fft()) { // lane-major twiddle tables of the wave-per-frame kernels
        auto lane_table = [&](const std::vector<cpx> &tw, DevBuf<float4> &dst) -> int {
            std::vector<cf> twc(tw.size());
            for (size_t i = 0; i < tw.size(); ++i) twc[i] = cf{tw[i].r, tw[i].i};
            const int entries = d.fft.nc == 256    ? wf_lane_table_entries<WF<256>>()
                                : d.fft.nc == 512  ? wf_lane_table_entries<WF<512>>()
                                : d.fft.nc == 1024 ? wf_lane_table_entries<WF<1024>>()
                                                   : wf_lane_table_entries<WF<2048>>();
            std::vector<cf> out(2 * (size_t)entries * 64);
            if (d.fft.nc == 256) wf_build_lane_table<WF<256>>(twc.data(), out.data());
            else if (d.fft.nc == 512) wf_build_lane_table<WF<512>>(twc.data(), out.data());
            else if (d.fft.nc == 1024) wf_build_lane_table<WF<1024>>(twc.data(), out.data());
            else wf_build_lane_table<WF<2048>>(twc.data(), out.data());
            std::vector<float4> o4((size_t)entries * 64);
            for (size_t i = 0; i < o4.size(); ++i) o4[i] = make_float4(out[2 * i].x, out[2 * i].y, out[2 * i + 1].x, out[2 * i + 1].y);
            return dst.upload(o4);
        };
        {
            // AUDIOMOD_PV_SPLIT_ANALYSIS=0: the one-wave kernel (also what the opt-in one-workgroup streaming kernel calls)
            const char *es = getenv("AUDIOMOD_PV_SPLIT_ANALYSIS");
            const char *sl = getenv("AUDIOMOD_PV_STREAM_LAUNCHES");
            split_analysis = d.fft.nc == 2048 && !(es && atoi(es) == 0) && !(sl && strcmp(sl, "single") == 0);
        }
        if (split_analysis) {
            std::vector<cf> twc(d.fft.tw_fwd.size());
            for (size_t i = 0; i < twc.size(); ++i) twc[i] = cf{d.fft.tw_fwd[i].r, d.fft.tw_fwd[i].i};
            const int entries = wf_lane_table_entries<WF2048S>();
            std::vector<cf> out(2 * (size_t)entries * WF2048S::LANES);
            wf_build_lane_table<WF2048S>(twc.data(), out.data());
            std::vector<float4> o4((size_t)entries * WF2048S::LANES);
            for (size_t i = 0; i < o4.size(); ++i) o4[i] = make_float4(out[2 * i].x, out[2 * i].y, out[2 * i + 1].x, out[2 * i + 1].y);
            if ((st = twl_fwd.upload(o4)) != PV_OK) return st;
        } else if ((st = lane_table(d.fft.tw_fwd, twl_fwd)) != PV_OK) return st;
        if ((st = lane_table(d.fft.tw_inv, twl_inv)) != PV_OK) return st;
    }
    if ((st = up2(st_inv, d.fft.st_inv)) != PV_OK) return st;
    if ((st = window.upload(d.window)) != PV_OK) return st;
    {
        // the window delayed by d = 0..3 samples: a frame that starts d floats past a 16-byte boundary is read
        // in aligned 16-byte pieces and multiplied by the copy that lines up with it (pv_analyze_wave_kernel)
        const size_t pitch = (size_t)d.N + 8;
        std::vector<float> sh(4 * pitch, 0.f);
        for (int dd = 0; dd < 4; ++dd)
            for (int j = 0; j < d.N; ++j) sh[dd * pitch + j + dd] = d.window[j];
        if ((st = window_sh.upload(sh)) != PV_OK) return st;
    }
    if ((st = sinc.upload(d.sinc)) != PV_OK) return st;
    {
        // interpolated-sinc coefficients expanded per sub-sample offset: row `off`, tap j = the four table
        // entries sinc[4 + (j+1)*ov - off + {-2,-1,0,1}] that resampler_basic_interpolate_single multiplies
        // into its four accumulators (resample.c:494-535); rows padded to filt_len + 1 (LDS bank spread)
        std::vector<float4> t4;
        if (d.resample && d.interp) {
            t4.assign((size_t)d.oversample * (d.filt_len + 1), make_float4(0, 0, 0, 0));
            for (int off = 0; off < d.oversample; ++off)
                for (int j = 0; j < d.filt_len; ++j) {
                    const float *sp = d.sinc.data() + 4 + (j + 1) * d.oversample - off - 2;
                    t4[(size_t)off * (d.filt_len + 1) + j] = make_float4(sp[0], sp[1], sp[2], sp[3]);
                }
        }
        if ((st = tab4.upload(t4)) != PV_OK) return st;
    }

    tb.N = d.N;
    tb.hs = d.hs;
    tb.H = d.H;
    tb.HP = HP;
    tb.nc = d.fft.nc;
    tb.log2nc = ilog2(d.fft.nc);
    tb.nstages = d.fft.nstages;
    for (int s = 0; s < d.fft.nstages; ++s) {
        tb.radix[s] = d.fft.radix[s];
        tb.log2m[s] = ilog2(d.fft.m[s]);
        tb.fstride[s] = d.fft.fstride[s];
    }
    tb.perm = perm.p;
    tb.iperm = iperm.p;
    tb.tw_fwd = tw_fwd.p;
    tb.twl_fwd = twl_fwd.p;
    tb.twl_inv = twl_inv.p;
    tb.tw_inv = tw_inv.p;
    tb.st_fwd = st_fwd.p;
    tb.st_inv = st_inv.p;
    tb.window = window.p;
    tb.window_sh = window_sh.p;

    const int cm = d.cfg.coremode;
    const size_t planes = (size_t)rows * TR;
    if ((st = mag.alloc(planes * HP)) != PV_OK) return st;
    if ((st = phase.alloc(planes * HP)) != PV_OK) return st;
    if (use_chain) FR = next_pow2_i(Tc + 1); // the chain reads a launch's own frames only (and none at wave-FFT sizes)
    if (!(use_chain && wave_fft()))
        if ((st = frames.alloc((size_t)rows * FR * d.N)) != PV_OK) return st;
    if (use_chain) {
        if ((st = st_acc.alloc(2 * (size_t)rows * chain_AR)) != PV_OK) return st; // read half + written half, swapped per launch
        if (d.resample)
            if ((st = stream.alloc((size_t)rows * ((size_t)chain_smask + 1))) != PV_OK) return st;
    }
    const bool bypass = d.robotic || d.whisper || d.constant || d.vocoder; // modes without a phase recurrence
    if (d.vocoder) {
        if ((st = cmag.alloc((size_t)TR * HP)) != PV_OK) return st;
        if ((st = cphase.alloc((size_t)TR * HP)) != PV_OK) return st;
    }
    if (!bypass && cm != 2) {
        if ((st = outphase.alloc(planes * HP)) != PV_OK) return st;
        if ((st = st_po.alloc((size_t)rows * d.hs)) != PV_OK) return st;
        if ((st = st_pp.alloc((size_t)rows * d.hs)) != PV_OK) return st;
    }
    if (!bypass && cm == 1) {
        if ((st = peaks.alloc(planes * PKP)) != PV_OK) return st;
        if ((st = npk.alloc(planes)) != PV_OK) return st;
        if ((st = modes.alloc(planes)) != PV_OK) return st;
        if ((st = recs.alloc(planes * PKP)) != PV_OK) return st;
        if ((st = rot.alloc(planes * PKP)) != PV_OK) return st;
        if ((st = st_rot.alloc((size_t)rows * PKP)) != PV_OK) return st;
        if ((st = st_kind.alloc((size_t)rows)) != PV_OK) return st;
    }
    return reset_state(nullptr);
}

int Core::reset_state(hipStream_t st) {
    if (st_pp.p) HIPC(hipMemsetAsync(st_pp.p, 0, st_pp.n * sizeof(float), st));
    if (st_po.p) HIPC(hipMemsetAsync(st_po.p, 0, st_po.n * sizeof(float), st));
    if (st_kind.p) HIPC(hipMemsetAsync(st_kind.p, 0, st_kind.n * sizeof(int32_t), st));
#ifdef PV_DIAG
    if (debug_stale_acc()) return PV_OK;
#endif
    if (st_acc.p) HIPC(hipMemsetAsync(st_acc.p, 0, st_acc.n * sizeof(float), st));
    acc_half = 0;
    return PV_OK;
}

bool Core::can_single_launch() const {
    static const bool on = [] {
        const char *e = getenv("AUDIOMOD_PV_STREAM_LAUNCHES");
        return e && strcmp(e, "single") == 0;
    }();
    if (!on || d.vocoder || use_chain) return false;
    StreamArgs probe{};
    probe.aa.tb = tb;
    probe.ma.hs = d.hs;
    probe.ma.PKP = PKP;
    probe.qa.hs = d.hs;
    probe.qa.PKP = PKP;
    probe.coremode = 1;
    probe.oa.tab_bytes = !d.resample ? 0
                         : d.interp  ? d.oversample * (d.filt_len + 1) * 16
                                     : (int)((d.sinc.size() * sizeof(float) + 15) & ~(size_t)15);
    probe.oa.lds_floats = ola_lds_floats;
    return stream_kernel_supported(probe);
}

int Core::build_tiles(const std::vector<SliceRec> &slices, int64_t t_base, int64_t t_end, int64_t ka, int64_t kb,
                      int32_t p_index_base, std::vector<OlaTile> &tiles, std::vector<float> &wacc) const {
    // slices[t - t_base] must exist for every t in [max(t_base, t_end - FR), t_end); a tile that needed an
    // older frame would be rejected below anyway (the frame ring no longer holds it)
    int64_t t_lo = t_end - FR > t_base ? t_end - FR : t_base; // monotone cursors
    int64_t t_hi = t_lo;
    for (int64_t k0 = ka; k0 < kb; k0 += kTileOut) {
        OlaTile tl{};
        tl.k0 = k0;
        tl.kcnt = (int32_t)((kb - k0) < kTileOut ? (kb - k0) : kTileOut);
        int64_t n_lo, n_hi;
        if (d.resample) {
            auto pos = [&](int64_t k) {
                return (int64_t)(d.filt_len / 2) + (int64_t)(((unsigned __int128)k * d.res_num) / d.res_den);
            };
            n_lo = pos(k0) - d.filt_len + 1;
            n_hi = pos(k0 + tl.kcnt - 1);
        } else {
            n_lo = k0;
            n_hi = k0 + tl.kcnt - 1;
        }
        tl.n_lo = n_lo;
        tl.n_cnt = (int32_t)(n_hi - n_lo + 1);
        if (tl.n_cnt > ola_lds_floats) {
            g_last_error = "internal: OLA tile larger than its LDS budget";
            return PV_ERR_UNSUPPORTED;
        }
        const int64_t n0 = n_lo < 0 ? 0 : n_lo;
        // t_first: smallest t with P_t + N > n0 ; t_last: largest t (< t_end) with P_t <= n_hi
        while (t_lo < t_end && slices[(size_t)(t_lo - t_base)].P + d.N <= n0) ++t_lo;
        if (t_hi < t_lo) t_hi = t_lo;
        while (t_hi + 1 < t_end && slices[(size_t)(t_hi + 1 - t_base)].P <= n_hi) ++t_hi;
        if (t_lo > t_base && t_lo == t_end - FR && slices[(size_t)(t_lo - 1 - t_base)].P + d.N > n0) {
            g_last_error = "internal: OLA tile needs a frame the ring no longer holds";
            return PV_ERR_UNSUPPORTED;
        }
        if (t_lo >= t_end) {
            g_last_error = "internal: OLA tile has no covering slice";
            return PV_ERR_UNSUPPORTED;
        }
        tl.t_first = (int32_t)t_lo;
        tl.t_cnt = (int32_t)(t_hi - t_lo + 1);
        if (tl.t_cnt > kMaxTileFrames || t_end - t_lo > FR) {
            g_last_error = "internal: OLA tile overlaps too many frames";
            return PV_ERR_UNSUPPORTED;
        }
        tl.p_off = (int32_t)(t_lo - p_index_base);
        tiles.push_back(tl);
        // window-sum denominator of every OLA sample of the tile: windowAccumulator at the moment writeSlice
        // divides (channelinfo.cc:108 seeds [0] with 1; synthesiseSlice :1073 adds w[i] * float(area*1.5) per
        // frame, ascending t).  Float arithmetic, evaluated exactly as written (-ffp-contract=off).
        const size_t wbase = wacc.size();
        wacc.resize(wbase + (size_t)wacc_pitch, 1.0f);
        for (int i = 0; i < tl.n_cnt; ++i) {
            const int64_t n = n_lo + i;
            float acc = n == 0 ? 1.f : 0.f;
            for (int64_t t = t_lo; t <= t_hi; ++t) {
                const int64_t off = n - slices[(size_t)(t - t_base)].P;
                if (off >= 0 && off < d.N) acc += d.window[(size_t)off] * d.win_gain;
            }
            wacc[wbase + (size_t)i] = acc;
        }
        // where each output of the tile sits in the OLA stream: last_sample = filt_len/2 + floor(k*num/den),
        // samp_frac_num = (k*num) mod den (closed form of resample.c:548-554 from skip_zeros :1225), and from
        // those the sub-sample offset and the interpolation fraction of resampler_basic_interpolate_single
        // (:494-500, float arithmetic as written there).  Data-independent and the same for every row, so the
        // 64-bit divisions happen here once instead of once per row in the kernel.
        if (d.resample) {
            uint32_t *ot = reinterpret_cast<uint32_t *>(wacc.data() + wbase + (size_t)otab_off);
            for (int o = 0; o < tl.kcnt; ++o) {
                const unsigned __int128 tot = (unsigned __int128)(k0 + o) * d.res_num;
                const int64_t pos = (int64_t)(d.filt_len / 2) + (int64_t)(tot / d.res_den);
                const uint32_t frac_num = (uint32_t)(tot % d.res_den);
                const uint32_t xoff = (uint32_t)(pos - d.filt_len + 1 - n_lo);
                uint32_t sub, fbits = 0;
                if (d.interp) {
                    const uint32_t ov = (uint32_t)d.oversample;
                    sub = frac_num * ov / d.res_den;
                    const float frac = ((float)((frac_num * ov) % d.res_den)) / d.res_den;
                    memcpy(&fbits, &frac, 4);
                } else {
                    sub = frac_num;
                }
                ot[2 * o] = xoff | (sub << 16);
                ot[2 * o + 1] = fbits;
            }
        }
    }
    return PV_OK;
}

void Core::build_res_tiles(int64_t ka, int64_t kb, std::vector<ResTile> &tiles, std::vector<uint2> &otab) const {
    for (int64_t k0 = ka; k0 < kb; k0 += kTileOut) {
        ResTile tl{};
        tl.k0 = k0;
        tl.kcnt = (int32_t)((kb - k0) < kTileOut ? (kb - k0) : kTileOut);
        auto pos = [&](int64_t k) {
            return (int64_t)(d.filt_len / 2) + (int64_t)(((unsigned __int128)k * d.res_num) / d.res_den);
        };
        tl.n_lo = pos(k0) - d.filt_len + 1;
        tl.n_cnt = (int32_t)(pos(k0 + tl.kcnt - 1) - tl.n_lo + 1);
        tiles.push_back(tl);
        // where each output of the tile sits in the stream: last_sample = filt_len/2 + floor(k*num/den),
        // samp_frac_num = (k*num) mod den (closed form of resample.c:548-554 from skip_zeros :1225), and from those
        // the sub-sample offset and the interpolation fraction of resampler_basic_interpolate_single (:494-500)
        for (int o = 0; o < kTileOut; ++o) {
            if (o >= tl.kcnt) {
                otab.push_back(make_uint2(0u, 0u));
                continue;
            }
            const unsigned __int128 tot = (unsigned __int128)(k0 + o) * d.res_num;
            const int64_t p = (int64_t)(d.filt_len / 2) + (int64_t)(tot / d.res_den);
            const uint32_t frac_num = (uint32_t)(tot % d.res_den);
            const uint32_t xoff = (uint32_t)(p - d.filt_len + 1 - tl.n_lo);
            uint32_t sub, fbits = 0;
            if (d.interp) {
                const uint32_t ov = (uint32_t)d.oversample;
                sub = frac_num * ov / d.res_den;
                const float frac = ((float)((frac_num * ov) % d.res_den)) / d.res_den;
                memcpy(&fbits, &frac, 4);
            } else {
                sub = frac_num;
            }
            otab.push_back(make_uint2(xoff | (sub << 16), fbits));
        }
    }
}

void Core::launch_chunk(const InAddr &ia, int64_t t0, int Tn, const int32_t *d_pinc, const OlaTile *d_tiles,
                        int ntiles, const int64_t *d_P, const float *d_wacc, const float *d_whisper,
                        const InAddr *carrier, float *out, int64_t out_stride_row, int64_t k_base,
                        hipStream_t st, hipEvent_t *ev, int part, hipStream_t st_chain, hipEvent_t ev_match,
                        hipEvent_t ev_chain, bool single_launch, const ChainLaunch *chain) const {
    // part 3 / 4: the back of a chunk in two pieces (fused path: 3 = everything up to the fused synthesis +
    // overlap-add kernel, 4 = the resampling kernel), for the three-stage order of pv_batch_run
    // part 5: only the rotation chain's hand-over to the second stream (for the order in which the chain starts behind
    // the previous chunk's fused kernel rather than right behind its own match kernel: ChainLaunch::late_chain)
    // part 6 / 7: the front of a chunk in two pieces (6 = the analysis kernel only, 7 = match kernel + the rotation
    // chain's hand-over), for the order in which the analysis runs one chunk further ahead (pv_batch_run, ahead order)
    const bool only_resample = part == 4, no_resample = part == 3, only_chain = part == 5;
    const bool only_analysis = part == 6, no_analysis = part == 7;
    const bool defer_chain = part == 1 && chain && chain->late_chain;
    if (part == 3 || part == 4) part = 2;
    if (part == 5 || part == 6 || part == 7) part = 1;
    const bool front = part != 2, back = part != 1;
    StreamArgs fused{};
    const bool bypass = d.robotic || d.whisper || d.constant || d.vocoder;
    const int cm = bypass ? -1 : ((d.cfg.coremode == 1 || d.cfg.coremode == 2) ? d.cfg.coremode : 0);
    auto rec = [&](int i) {
        if (ev) HIPV(hipEventRecord(ev[i], st));
    };
    // the phase stage's latency-bound kernel: on `st` (part 0), or handed to the second stream after what the
    // main stream has launched so far (part 1) and waited for before what follows (part 2)
    auto side_stream = [&](int k, auto &&launch_on) {
        if (defer_chain) return;
        if (part == 1) {
            HIPV(hipEventRecord(ev_match, st));
            HIPV(hipStreamWaitEvent(st_chain, ev_match, 0));
            if (ev) HIPV(hipEventRecord(ev[2 * k], st_chain));
            launch_on(st_chain);
            if (ev) HIPV(hipEventRecord(ev[2 * k + 1], st_chain));
            HIPV(hipEventRecord(ev_chain, st_chain));
        } else if (part == 2) {
            if (!only_resample) HIPV(hipStreamWaitEvent(st, ev_chain, 0));
        } else {
            rec(2 * k);
            launch_on(st);
            rec(2 * k + 1);
        }
    };
    AnalyzeArgs aa{};
    aa.tb = tb;
    aa.ia = ia;
    aa.hop = d.hop;
    const int s0 = (int)(t0 % TR);
    aa.t0 = t0;
    aa.s0 = s0;
    aa.Tn = Tn;
    aa.TR = TR;
    aa.rows = rows;
    aa.PKP = PKP;
    aa.find_peaks = cm == 1 ? 1 : 0;
    aa.split = split_analysis ? 1 : 0;
    aa.mag = mag.p;
    aa.phase = phase.p;
    aa.peaks = peaks.p;
    aa.npk = npk.p;
    const bool do_analysis = front && !only_chain && !no_analysis;
    if (do_analysis) rec(2 * PV_K_ANALYZE);
    if (single_launch) fused.aa = aa;
    else if (do_analysis) launch_analyze(aa, st); HIPV(hipGetLastError());
    if (do_analysis && d.vocoder && carrier) {
        // the carrier is one more (data-independent) row: same analysis, its own planes
        AnalyzeArgs ca = aa;
        ca.ia = *carrier;
        ca.rows = 1;
        ca.find_peaks = 0;
        ca.mag = cmag.p;
        ca.phase = cphase.p;
        launch_analyze(ca, st); HIPV(hipGetLastError());
    }
    if (do_analysis) rec(2 * PV_K_ANALYZE + 1);
    if (only_analysis) return;

    if (cm == 1) {
        MatchArgs ma{};
        ma.N = d.N;
        ma.hs = d.hs;
        ma.HP = HP;
        ma.PKP = PKP;
        ma.C = C;
        ma.hop = d.hop;
        ma.TR = TR;
        ma.rows = rows;
        ma.Tn = Tn;
        ma.two_pi_hop = d.two_pi_hop;
        ma.t0 = t0;
        ma.s0 = s0;
        ma.phase_inc = d_pinc;
        ma.phase = phase.p;
        ma.peaks = peaks.p;
        ma.npk = npk.p;
        ma.recs = recs.p;
        ma.modes = modes.p;
        const bool phase_fused = fuse_phase && part == 0 && !single_launch && !ev; // (filled in below: needs qa)
        if (front && !only_chain) rec(2 * PV_K_MATCH);
        if (single_launch) fused.ma = ma;
        else if (front && !only_chain && !phase_fused) launch_match(ma, st); HIPV(hipGetLastError());
        if (front && !only_chain) rec(2 * PV_K_MATCH + 1);
        SeqArgs qa{};
        qa.N = d.N;
        qa.hs = d.hs;
        qa.HP = HP;
        qa.PKP = PKP;
        qa.C = C;
        qa.hop = d.hop;
        qa.TR = TR;
        qa.rows = rows;
        qa.Tn = Tn;
        qa.two_pi_hop = d.two_pi_hop;
        qa.t0 = t0;
        qa.s0 = s0;
        qa.phase_inc = d_pinc;
        qa.phase = phase.p;
        qa.peaks = peaks.p;
        qa.npk = npk.p;
        qa.recs = recs.p;
        qa.modes = modes.p;
        qa.rot = rot.p;
        qa.outphase = outphase.p;
        qa.st_kind = st_kind.p;
        qa.st_rot = st_rot.p;
        qa.st_po = st_po.p;
        {
            static const int prio = [] {
                const char *e = getenv("AUDIOMOD_PV_SEQ_PRIO");
                return e ? atoi(e) : -1;
            }();
            qa.high_prio = prio >= 0 ? prio : 1; // (measured without: no difference, 56.6 vs 56.4 ms per step)
            static const int narrow = [] {
                const char *e = getenv("AUDIOMOD_PV_SEQ_NARROW");
                return e ? atoi(e) : 0;
            }();
            qa.narrow = narrow;
        }
        if (single_launch) fused.qa = qa;
        else if (phase_fused) {
            if (!launch_phase(ma, qa, st)) { // (does not fit one workgroup's LDS: the two kernels after all)
                launch_match(ma, st); HIPV(hipGetLastError());
                launch_seq(qa, st);
            }
            HIPV(hipGetLastError());
        } else side_stream(PV_K_SEQ, [&](hipStream_t s) { launch_seq(qa, s); HIPV(hipGetLastError()); });
    } else if (cm == 0) {
        PropArgs pa{};
        pa.N = d.N;
        pa.hs = d.hs;
        pa.HP = HP;
        pa.C = C;
        pa.hop = d.hop;
        pa.TR = TR;
        pa.rows = rows;
        pa.Tn = Tn;
        pa.two_pi_hop = d.two_pi_hop;
        pa.t0 = t0;
        pa.s0 = s0;
        pa.phase_inc = d_pinc;
        pa.phase = phase.p;
        pa.outphase = outphase.p;
        pa.st_pp = st_pp.p;
        pa.st_po = st_po.p;
        if (single_launch) fused.pa = pa;
        else side_stream(PV_K_PROP, [&](hipStream_t s) { launch_prop(pa, s); HIPV(hipGetLastError()); });
    }

    SynthArgs sa{};
    sa.tb = tb;
    sa.hop = d.hop;
    sa.C = C;
    sa.two_pi_hop = d.two_pi_hop;
    sa.do_freq_comp = d.do_freq_comp ? 1 : 0;
    sa.freq_comp = d.freq_comp;
    sa.fixed_gain = d.fixed_gain;
    sa.inv_n = d.inv_n;
    sa.robotic = d.robotic ? 1 : 0;
    sa.passthru = d.constant ? 1 : 0;
    sa.whisper = d.whisper ? d_whisper : nullptr;
    sa.voc_band_len = d.vocoder ? d.voc_band_len : -1;
    sa.cmag = cmag.p;
    sa.cphase = cphase.p;
    sa.coremode = cm < 0 ? 0 : cm;
    sa.t0 = t0;
    sa.s0 = s0;
    sa.Tn = Tn;
    sa.TR = TR;
    sa.rows = rows;
    sa.PKP = PKP;
    sa.phase_inc = d_pinc;
    sa.mag = mag.p;
    sa.phase = phase.p;
    sa.outphase = outphase.p;
    sa.peaks = peaks.p;
    sa.npk = npk.p;
    sa.modes = modes.p;
    sa.rot = rot.p;
    sa.frames = frames.p;
    sa.FR = FR;
    if (d.cepstral && !only_resample) {
        CepstralArgs ca{};
        ca.tb = tb;
        ca.Tn = Tn;
        ca.TR = TR;
        ca.rows = rows;
        ca.s0 = s0;
        ca.env_comp = d.env_comp;
        ca.inv_n = d.inv_n;
        ca.mag = mag.p;
        if (back) rec(2 * PV_K_CEPSTRAL);
        if (single_launch) fused.ca = ca;
        else if (back) launch_cepstral(ca, st); HIPV(hipGetLastError());
        if (back) rec(2 * PV_K_CEPSTRAL + 1);
    }
    if (chain && !single_launch) {
        if (!back) return;
        ChainArgs ca{};
        ca.N = d.N;
        ca.rows = rows;
        ca.C = C;
        ca.Tn = Tn;
        ca.AR = chain_AR;
        ca.smask = chain_smask;
        ca.waves = chain_waves;
        {
#ifdef PV_DIAG
            static const int diag = [] {
                const char *e = getenv("AUDIOMOD_PV_CHAIN_DIAG");
                return e ? atoi(e) : 0;
            }();
            ca.diag = diag;
#endif
        }
        ca.slices = chain->slices;
        ca.run_off = chain->run_off;
        ca.runs = chain->runs;
        ca.wden = chain->wden;
        ca.wden_hi = chain->wden_hi;
        // the ring images: this launch reads one half and writes the other (ChainArgs::st_acc_in); launches are
        // enqueued in slice order on one stream, so flipping at enqueue time is flipping in execution order
        ca.st_acc = st_acc.p;
        ca.acc_sel = acc_half | (t0 == 0 ? 2 : 0);
        if (!only_resample) acc_half ^= 1;
#ifdef PV_DIAG
        // AUDIOMOD_PV_DEBUG_STALE_ACC=1 (diagnostic builds): recreate round 2's hazard on purpose -- one buffer for both
        // directions and a first launch that reads it -- to see what a run 0 that starts from the previous pass's final
        // accumulator image puts out (reset_state leaves the image alone under the same switch)
        if (debug_stale_acc()) ca.acc_sel = 4; // bit 2 (diagnostic builds): read AND write half 0, never fresh
#endif
        ca.stream = stream.p;
        ca.resample = d.resample ? 1 : 0;
        ca.out = chain->out;
        ca.out_stride_row = out_stride_row;
        ca.frames = frames.p;
        ca.FR = FR;
        ca.t0 = t0;
        ca.fast = fast_chain() ? 1 : 0;
        ResArgs ra{};
        ra.rows = rows;
        ra.ntiles = chain->res_ntiles;
        ra.smask = chain_smask;
        ra.stream = stream.p;
        ra.tiles = chain->res_tiles;
        ra.otab = chain->res_otab;
        ra.interp = d.interp ? 1 : 0;
        ra.filt_len = d.filt_len;
        ra.oversample = d.oversample;
        ra.sinc = sinc.p;
        ra.sinc_len = d.resample ? (int)d.sinc.size() : 0;
        ra.tab4 = tab4.p;
        ra.lds_floats = ola_lds_floats;
        ra.tab_bytes = !d.resample ? 0
                       : d.interp  ? d.oversample * (d.filt_len + 1) * 16
                                   : (int)((d.sinc.size() * sizeof(float) + 15) & ~(size_t)15);
        ra.out = out;
        ra.out_stride_row = out_stride_row;
        ra.k_base = k_base;
        ra.fast = fast_chain() ? 1 : 0; // (the modes with a free-form fused kernel: none of them can put NaN into the stream)
        if (d.resample && chain->res_stream && chain->ev_ring_free && !only_resample)
            HIPV(hipStreamWaitEvent(st, chain->ev_ring_free, 0));
        if (only_resample) {
        } else if (wave_fft()) {
            // synthesis and overlap-add in one kernel: the frames stay in LDS
            rec(2 * PV_K_SYNTH_OLA);
            launch_synth_chain(sa, ca, st); HIPV(hipGetLastError());
            rec(2 * PV_K_SYNTH_OLA + 1);
        } else {
            rec(2 * PV_K_SYNTH);
            launch_synth(sa, st); HIPV(hipGetLastError());
            rec(2 * PV_K_SYNTH + 1);
            rec(2 * PV_K_OLA_RESAMPLE);
            launch_frames_chain(ca, st); HIPV(hipGetLastError());
            if (!d.resample || no_resample) rec(2 * PV_K_OLA_RESAMPLE + 1);
        }
        if (no_resample) return;
        if (d.resample && chain->res_stream) {
            HIPV(hipEventRecord(chain->ev_fused, st));
            HIPV(hipStreamWaitEvent(chain->res_stream, chain->ev_fused, 0));
            if (ev && wave_fft()) HIPV(hipEventRecord(ev[2 * PV_K_OLA_RESAMPLE], chain->res_stream));
            launch_resample(ra, chain->res_stream); HIPV(hipGetLastError());
            if (ev) HIPV(hipEventRecord(ev[2 * PV_K_OLA_RESAMPLE + 1], chain->res_stream));
            HIPV(hipEventRecord(chain->ev_res, chain->res_stream));
        } else if (d.resample) {
            if (wave_fft() || only_resample) rec(2 * PV_K_OLA_RESAMPLE);
            launch_resample(ra, st); HIPV(hipGetLastError());
            rec(2 * PV_K_OLA_RESAMPLE + 1);
        }
        return;
    }
    if (back) rec(2 * PV_K_SYNTH);
    if (single_launch) fused.sa = sa;
    else if (back) launch_synth(sa, st); HIPV(hipGetLastError());
    if (back) rec(2 * PV_K_SYNTH + 1);

    OlaArgs oa{};
    oa.N = d.N;
    oa.rows = rows;
    oa.FR = FR;
    oa.frames = frames.p;
    oa.tiles = d_tiles;
    oa.P = d_P;
    oa.ntiles = ntiles;
    oa.resample = d.resample ? 1 : 0;
    oa.interp = d.interp ? 1 : 0;
    oa.num = d.res_num;
    oa.den = d.res_den;
    oa.filt_len = d.filt_len;
    oa.oversample = d.oversample;
    oa.sinc = sinc.p;
    oa.tab4 = tab4.p;
    oa.wacc = d_wacc;
    oa.sinc_len = d.resample ? (int)d.sinc.size() : 0;
    oa.lds_floats = ola_lds_floats;
    oa.wacc_pitch = wacc_pitch;
    oa.otab_off = otab_off;
    oa.tab_bytes = !d.resample ? 0
                   : d.interp  ? d.oversample * (d.filt_len + 1) * 16
                               : (int)((d.sinc.size() * sizeof(float) + 15) & ~(size_t)15);
    oa.out = out;
    oa.out_stride_row = out_stride_row;
    oa.k_base = k_base;
    if (single_launch) {
        fused.oa = oa;
        fused.coremode = cm;
        fused.cepstral = d.cepstral ? 1 : 0;
        launch_stream(fused, st); HIPV(hipGetLastError());
    } else if (back && ntiles > 0) {
        rec(2 * PV_K_OLA_RESAMPLE);
        launch_ola(oa, st); HIPV(hipGetLastError());
        rec(2 * PV_K_OLA_RESAMPLE + 1);
    }
}

static void fill_info(const Derived &d, int64_t slices, pv_info *o) {
    memset(o, 0, sizeof(*o));
    o->fftsize = d.N;
    o->hop_in = d.hop;
    o->hop_out_nominal = d.hop_out_nominal;
    o->outbuf_capacity = d.outbuf_cap;
    o->pitch_scale = d.pitch_scale;
    o->hs_ratio = d.hs_ratio;
    o->int_ratio = d.int_ratio;
    o->resample = d.resample;
    if (d.resample) {
        o->res_num = d.res_num;
        o->res_den = d.res_den;
        o->res_filt_len = d.filt_len;
        o->res_oversample = d.oversample;
        o->res_interp = d.interp;
    }
    o->slices = slices;
    o->bytes_per_slice = bytes_per_slice(d);
}

} // namespace pv

using namespace pv;

// ------------------------------------------------------------------------------------------
// batch engine
// ------------------------------------------------------------------------------------------
static constexpr size_t kEvPerChunk = 2 * PV_NUM_KERNELS;

struct pv_batch {
    Core core;
    BatchPlan plan;
    int64_t frames = 0;
    struct Chunk {
        int64_t t0;
        int Tn;
        int tile_begin, ntiles;
        int64_t k0; // fused path: first output of the chunk
        int res_begin, res_ntiles; // ... and its tiles of the resampling kernel
        int64_t cs_begin;          // ... its run lists in d_cs
        int run_begin, runs;       // ... and their offsets in d_run_off
    };
    DevBuf<ChainSlice> d_cs; // fused path: the run lists of every chunk
    DevBuf<int32_t> d_run_off;
    DevBuf<float> d_wden, d_wden_hi;
    DevBuf<ResTile> d_res_tiles;
    DevBuf<uint2> d_res_otab;
    std::vector<Chunk> chunks;
    DevBuf<int32_t> d_pinc;
    DevBuf<int64_t> d_P;
    DevBuf<OlaTile> d_tiles;
    DevBuf<float> d_wacc;
    DevBuf<float> d_whisper; // WHISPER mode: [slices][C][HP] host-drawn phases, shared by all streams
    DevBuf<float> d_carrier; // vocoder modes: the carrier signal for every sample fed (incl. the zero flush)
    hipStream_t chain_stream = nullptr; // second HIP stream for the rotation chain (phase-locked mode)
    hipEvent_t ev_match[4] = {}, ev_chain[4] = {};
    hipStream_t res_stream = nullptr;   // fused path, resampling configurations: the resampling kernel's stream
    hipEvent_t ev_fused[4] = {}, ev_res[4] = {};
    int timing = 0; // 0 = off, n = instrument every n-th chunk
    std::vector<hipEvent_t> ev_pool; // kEvPerChunk per instrumented chunk
    std::vector<int> ev_chunk;       // chunk index of each used pool segment
    size_t ev_used = 0;
    double acc_ms[PV_NUM_KERNELS] = {};
    int64_t acc_n[PV_NUM_KERNELS] = {};
    ~pv_batch() {
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        for (int i = 0; i < 4; ++i) {
            if (ev_match[i]) (void)hipEventDestroy(ev_match[i]);
            if (ev_chain[i]) (void)hipEventDestroy(ev_chain[i]);
            if (ev_fused[i]) (void)hipEventDestroy(ev_fused[i]);
            if (ev_res[i]) (void)hipEventDestroy(ev_res[i]);
        }
        if (res_stream) (void)hipStreamDestroy(res_stream);
        if (chain_stream) (void)hipStreamDestroy(chain_stream);
    }
};

struct pv_engine {
    Core core;
    int poisoned = 0; // status of a failure that left device state and host bookkeeping out of step (sticky)
    std::string poison_reason;
    std::unique_ptr<Planner> planner;
    std::unique_ptr<ChainBuilder> chain; // fused path: the running host plan of the overlap-add rings
    std::vector<SliceRec> recent; // slice table window; recent[0] is slice t_base
    int64_t t_base = 0;
    int64_t fed = 0, uploaded = 0;
    hipStream_t stream = nullptr;
    int ring = 0; // device input ring length (power of two) per channel
    DevBuf<float> d_in, d_out, d_whisper, d_carrier;
    PinBuf<float> h_whisper, h_carrier;
    std::unique_ptr<WhisperRng> rng;
    std::unique_ptr<CarrierGen> cargen;
    DevBuf<char> d_desc;
    PinBuf<float> h_in, h_out;
    PinBuf<char> h_desc;
    // Round 3: a call's kernels write their output straight into page-locked host memory (h_out is mapped into the
    // device's address space: out_dev) and the stream then writes a sequence number into h_flag, which pv_feed spins
    // on -- no device-to-host copy and no hipStreamSynchronize per call (AUDIOMOD_PV_STREAM_SYNC=1: the copy + the
    // synchronisation, as before).
    PinBuf<uint32_t> h_flag;
    float *out_dev = nullptr;     // device-side address of h_out
    uint32_t *flag_dev = nullptr; // ... of h_flag
    uint32_t flag_seq = 0;
    bool direct_out = false;
    int out_cap = 0; // per-row capacity of d_out / h_out
    std::vector<std::vector<float>> outq; // per channel FIFO
    size_t outq_head = 0;
    ~pv_engine() {
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" {

const char *pv_strerror(int s) {
    switch (s) {
    case PV_OK: return "ok";
    case PV_ERR_INVALID_ARG: return "invalid argument";
    case PV_ERR_UNSUPPORTED: return "unsupported configuration";
    case PV_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback)";
    case PV_ERR_HIP: return "HIP runtime error";
    case PV_ERR_OUTPUT_OVERRUN: return "output ring overrun";
    default: return "unknown status";
    }
}

static int g_arith = [] {
    const char *e = getenv("AUDIOMOD_PV_EXACT");
    return (e && atoi(e) != 0) ? PV_ARITH_EXACT : PV_ARITH_FAST;
}();
int pv_set_arithmetic(int arith) {
    if (arith != PV_ARITH_FAST && arith != PV_ARITH_EXACT) return PV_ERR_INVALID_ARG;
    g_arith = arith;
    return PV_OK;
}
int pv_get_arithmetic(void) { return g_arith; }

const char *pv_last_error(void) { return g_last_error.empty() ? plan_reason() : g_last_error.c_str(); }

int pv_device_count(void) { return count_gfx950(); }

const char *pv_kernel_name(int k) {
    static const char *n[PV_NUM_KERNELS] = {"pv_analyze_kernel", "pv_match_kernel", "pv_seq_kernel",
                                            "pv_prop_kernel",    "pv_synth_kernel", "pv_ola_kernel",
                                            "pv_cepstral_kernel", "pv_synth_ola_kernel"};
    return (k >= 0 && k < PV_NUM_KERNELS) ? n[k] : "";
}

int pv_plan_simulate(const pv_config *cfg, const int32_t *n, int32_t ncalls, int32_t *avail, int32_t *shift,
                     int32_t *phase, int64_t max_slices, int64_t *nslices, pv_info *info) {
    g_last_error.clear();
    plan_reason_clear();
    if (!cfg || (ncalls > 0 && !n)) return PV_ERR_INVALID_ARG;
    Derived d;
    int st = derive(*cfg, d);
    if (st != PV_OK) return st;
    Planner pl(d);
    std::vector<SliceRec> sl;
    for (int i = 0; i < ncalls; ++i) {
        st = pl.feed(n[i], sl);
        if (st != PV_OK) return st;
        if (avail) avail[i] = pl.available();
        pl.retrieve(pl.available());
    }
    for (int64_t i = 0; i < (int64_t)sl.size() && i < max_slices; ++i) {
        if (shift) shift[i] = sl[(size_t)i].shift;
        if (phase) phase[i] = sl[(size_t)i].phase_inc;
    }
    if (nslices) *nslices = (int64_t)sl.size();
    if (info) fill_info(d, (int64_t)sl.size(), info);
    return PV_OK;
}

int64_t pv_plan_table(const pv_config *cfg, int which, float *out, int64_t max) {
    g_last_error.clear();
    plan_reason_clear();
    if (!cfg || max < 0 || (max > 0 && !out)) return -(int64_t)PV_ERR_INVALID_ARG;
    Derived d;
    const int st = derive(*cfg, d);
    if (st != PV_OK) return -(int64_t)st;
    if (which == PV_TABLE_CARRIER) {
        CarrierGen gen((float)cfg->sample_rate, cfg->mode == PV_MODE_VOCODER_CHORD);
        for (int64_t i = 0; i < max; ++i) out[i] = gen.next();
        return max;
    }
    const std::vector<float> *t = which == PV_TABLE_WINDOW ? &d.window : which == PV_TABLE_SINC ? &d.sinc : nullptr;
    if (!t) return -(int64_t)PV_ERR_INVALID_ARG;
    const int64_t n = (int64_t)t->size() < max ? (int64_t)t->size() : max;
    if (n > 0) memcpy(out, t->data(), (size_t)n * sizeof(float));
    return (int64_t)t->size();
}

int pv_plan_whisper_phases(int64_t n, float *out) {
    if (n < 0 || (n > 0 && !out)) return PV_ERR_INVALID_ARG;
    WhisperRng rng;
    for (int64_t i = 0; i < n; ++i) out[i] = rng.next_phase();
    return PV_OK;
}

// ---------------------------------------------------------------- batch
int pv_batch_create(const pv_config *cfg, int32_t nstreams, int64_t frames, int32_t block, int32_t flush, int device,
                    pv_batch **out) {
    g_last_error.clear();
    plan_reason_clear();
    if (!cfg || !out || nstreams < 1 || frames < 1 || block < 1) return PV_ERR_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<pv_batch> b(new pv_batch());
    const int rows = nstreams * cfg->channels;
    // slices per launch and row: 64 K slices per launch amortise the launch's fixed costs and tail (measured:
    // +8 % over 16 K with 256 rows; flat beyond), and the planes of such a chunk are a few GB of the 288
    // Round 2, fused path (192 rows and up): 128 K slices per launch, up to 512 per row -- half as many kernel
    // boundaries (each a drain and a refill of the chip): 52.0 vs 52.9 ms per bench step; 768 per row: no further gain.
    // Round 3: with PV_ARITH_FAST every wave-FFT configuration that has a free-form kernel takes the fused path at
    // any row count (Core::fast_capable), and the wide chunks with it (8-95 streams: +5...20 % over the tile path).
    const bool fast_wave = g_arith == PV_ARITH_FAST && (cfg->fftsize > 256 && cfg->fftsize <= 4096);
    const bool wide = rows >= 192 || fast_wave;
    int Tc = (wide ? 131072 : 65536) / (rows > 0 ? rows : 1);
    if (Tc < 16) Tc = 16;
    if (Tc > (wide ? 512 : 256)) Tc = wide ? 512 : 256;
    if (const char *env = getenv("AUDIOMOD_PV_CHUNK_SLICES")) { // tuning knob: slices per launch and row
        const int v = atoi(env);
        if (v >= 4 && v <= 1024) Tc = v;
    }
    b->core.pipelined_planes = Core::pipeline_wanted(*cfg);
    b->core.fast_arith = g_arith == PV_ARITH_FAST;
    int st;
    {
        // the plan first: the overlap-add rings are sized from the advances it really contains
        Derived dd;
        if ((st = derive(*cfg, dd)) != PV_OK) return st;
        if ((st = plan_batch(dd, frames, block, flush != 0, b->plan)) != PV_OK) return st;
        int mx = 1;
        bool drops = false;
        for (const SliceRec &r : b->plan.slices) {
            mx = r.adv > mx ? r.adv : mx;
            drops = drops || r.adv == 0;
        }
        b->core.chain_max_adv = mx;
        b->core.chain_required = drops;
    }
    st = b->core.init(*cfg, device, nstreams, Tc);
    if (st != PV_OK) return st;
    Core &c = b->core;
    b->frames = frames;
    const auto &sl = b->plan.slices;
    const int64_t T = (int64_t)sl.size();
    if (!c.use_chain)
        for (const SliceRec &r : sl)
            if (r.adv == 0) {
                g_last_error = "more output pending than the reference's output ring holds (the reference drops "
                               "slices there); only the fused overlap-add path reproduces that";
                return PV_ERR_OUTPUT_OVERRUN;
            }
    std::vector<int32_t> pinc((size_t)T);
    std::vector<int64_t> P((size_t)T);
    for (int64_t t = 0; t < T; ++t) {
        pinc[(size_t)t] = sl[(size_t)t].phase_inc;
        P[(size_t)t] = sl[(size_t)t].P;
    }
    std::vector<OlaTile> tiles;
    std::vector<float> wacc;
    std::vector<ChainSlice> cs;
    std::vector<int32_t> run_off;
    std::vector<float> wden, wden_hi;
    std::vector<ResTile> res_tiles;
    std::vector<uint2> res_otab;
    ChainBuilder cb(c.d, c.chain_AR, c.chain_smask);
    for (int64_t t0 = 0; t0 < T; t0 += Tc) {
        pv_batch::Chunk ch;
        ch.t0 = t0;
        ch.Tn = (int)((T - t0) < Tc ? (T - t0) : Tc);
        ch.k0 = 0;
        ch.res_begin = ch.res_ntiles = 0;
        ch.cs_begin = 0;
        ch.run_begin = 0;
        ch.runs = 1;
        const int64_t t1 = t0 + ch.Tn;
        int64_t ka = sl[(size_t)t0].K0;
        int64_t kb = sl[(size_t)(t1 - 1)].K0 + sl[(size_t)(t1 - 1)].cnt;
        if (ka > b->plan.out_frames) ka = b->plan.out_frames;
        if (kb > b->plan.out_frames) kb = b->plan.out_frames;
        ch.tile_begin = (int)tiles.size();
        if (c.use_chain) {
            ch.k0 = cb.begin_launch(sl[(size_t)t0]);
            for (int64_t t = t0; t < t1; ++t) cb.add(sl[(size_t)t], b->plan.out_frames, wden, wden_hi);
            // one workgroup per row leaves CUs idle below 256 rows: split the rows' slices into runs
            int runs_wanted = (256 + c.rows - 1) / c.rows;
            if (c.rows > 256) {
                // ... and a partly filled last round above them (320 rows: two rounds for 1.25 rounds of work): pick the
                // run count with the best product of round occupancy and useful share of a run (each later run
                // re-adds the frames that reach into its range).  320 rows: 4 runs, 11.8 -> 13.1 G samples/s.
                const double warm = (double)c.d.N / (double)(c.d.min_shift > 0 ? c.d.min_shift : 1) + 1.0;
                double best = 0.0;
                for (int r = 1; r <= 8; ++r) {
                    const double wgs = (double)c.rows * r / 256.0, len = (double)(t1 - t0) / r;
                    const double eff = wgs / std::ceil(wgs) * (r == 1 ? 1.0 : len / (len + warm));
                    if (eff > best + 1e-9) best = eff, runs_wanted = r;
                }
            }
            if (const char *e = getenv("AUDIOMOD_PV_CHAIN_RUNS")) runs_wanted = atoi(e);
            ch.cs_begin = (int64_t)cs.size();
            ch.run_begin = (int)run_off.size();
            ch.runs = cb.end_launch(runs_wanted > 32 ? 32 : runs_wanted, cs, run_off);
            if (c.d.resample && kb > ka) {
                ch.res_begin = (int)res_tiles.size();
                c.build_res_tiles(ka, kb, res_tiles, res_otab);
                ch.res_ntiles = (int)res_tiles.size() - ch.res_begin;
            }
        } else if (kb > ka) {
            st = c.build_tiles(sl, 0, t1, ka, kb, 0, tiles, wacc);
            if (st != PV_OK) return st;
        }
        ch.ntiles = (int)tiles.size() - ch.tile_begin;
        b->chunks.push_back(ch);
    }
    if ((st = b->d_pinc.upload(pinc)) != PV_OK) return st;
    if ((st = b->d_P.upload(P)) != PV_OK) return st;
    if ((st = b->d_tiles.upload(tiles)) != PV_OK) return st;
    if ((st = b->d_wacc.upload(wacc)) != PV_OK) return st;
    if (c.use_chain) {
        // (one spare entry each: a slice that finalises or emits nothing still prefetches its first entry)
        while (wden.size() & 3) wden.push_back(1.f), wden_hi.push_back(1.f);
        for (int i = 0; i < 4; ++i) wden.push_back(1.f), wden_hi.push_back(1.f);
        if (c.fast_chain()) { // the free-form kernel normalises by multiplying
            for (float &v : wden) v = 1.0f / v;
            for (float &v : wden_hi) v = 1.0f / v;
        }
        if ((st = b->d_cs.upload(cs)) != PV_OK) return st;
        if ((st = b->d_run_off.upload(run_off)) != PV_OK) return st;
        if ((st = b->d_wden.upload(wden)) != PV_OK) return st;
        if (cb.any_upper_skip) {
            if ((st = b->d_wden_hi.upload(wden_hi)) != PV_OK) return st;
        }
        if ((st = b->d_res_tiles.upload(res_tiles)) != PV_OK) return st;
        if ((st = b->d_res_otab.upload(res_otab)) != PV_OK) return st;
    }
    if (c.can_overlap_chain()) {
        // highest stream priority: the dispatcher must place the chain's few workgroups ahead of the thousands
        // of overlap-add tiles queued on the main stream, or the chain only starts when the tiles are done
        int prio_lo = 0, prio_hi = 0;
        HIPC(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIPC(hipStreamCreateWithPriority(&b->chain_stream, hipStreamNonBlocking, prio_hi));
        for (int i = 0; i < 4; ++i) {
            HIPC(hipEventCreateWithFlags(&b->ev_match[i], hipEventDisableTiming));
            HIPC(hipEventCreateWithFlags(&b->ev_chain[i], hipEventDisableTiming));
        }
    }
    if (c.use_chain && c.d.resample) {
        // AUDIOMOD_PV_RES_STREAM=1: the resampling kernel on a stream of its own, beside the next chunk's analysis.
        // Measured: no gain -- both kernels slow down by more than the overlap returns (63.2 vs 60.8 ms per step) --
        // so it is off unless asked for.
        const char *e = getenv("AUDIOMOD_PV_RES_STREAM");
        if (e && atoi(e) != 0) {
            HIPC(hipStreamCreateWithFlags(&b->res_stream, hipStreamNonBlocking));
            for (int i = 0; i < 4; ++i) {
                HIPC(hipEventCreateWithFlags(&b->ev_fused[i], hipEventDisableTiming));
                HIPC(hipEventCreateWithFlags(&b->ev_res[i], hipEventDisableTiming));
            }
        }
    }
    if (c.d.vocoder) {
        CarrierGen gen((float)c.d.cfg.sample_rate, c.d.chord);
        std::vector<float> car((size_t)b->plan.in_frames);
        for (auto &v : car) v = gen.next();
        if ((st = b->d_carrier.upload(car)) != PV_OK) return st;
    }
    if (c.d.whisper) {
        // every stream behaves like a fresh reference process, so all of them draw the same rand() sequence:
        // slice-major, channel ch0, ch1, ..., bins 0..N/2 (whisperSlice runs inside processSliceForChannel)
        WhisperRng rng;
        std::vector<float> wp((size_t)T * c.C * c.HP, 0.f);
        for (int64_t t = 0; t < T; ++t)
            for (int ch = 0; ch < c.C; ++ch)
                for (int k = 0; k <= c.d.hs; ++k) wp[((size_t)t * c.C + ch) * c.HP + k] = rng.next_phase();
        if ((st = b->d_whisper.upload(wp)) != PV_OK) return st;
    }
    *out = b.release();
    return PV_OK;
}

void pv_batch_destroy(pv_batch *b) { delete b; }

int64_t pv_batch_out_frames(const pv_batch *b) { return b ? b->plan.out_frames : -1; }
int64_t pv_batch_slices(const pv_batch *b) { return b ? (int64_t)b->plan.slices.size() : -1; }
int32_t pv_batch_launches(const pv_batch *b) { return b ? (int32_t)b->chunks.size() : -1; }
int32_t pv_batch_pipelined(const pv_batch *b) { return b ? (b->chain_stream != nullptr ? 1 : 0) : -1; }

int pv_batch_get_info(const pv_batch *b, pv_info *info) {
    if (!b || !info) return PV_ERR_INVALID_ARG;
    fill_info(b->core.d, (int64_t)b->plan.slices.size(), info);
    return PV_OK;
}

int pv_batch_enable_timing(pv_batch *b, int on) {
    if (!b) return PV_ERR_INVALID_ARG;
    b->timing = on < 0 ? 0 : on;
    b->ev_chunk.clear();
    for (int k = 0; k < PV_NUM_KERNELS; ++k) {
        b->acc_ms[k] = 0;
        b->acc_n[k] = 0;
    }
    b->ev_used = 0;
    return PV_OK;
}

int pv_batch_run(pv_batch *b, const float *d_in, float *d_out, void *hip_stream) {
    g_last_error.clear();
    plan_reason_clear();
    if (!b || !d_in || (!d_out && b->plan.out_frames > 0)) return PV_ERR_INVALID_ARG; // an empty output needs no buffer
    Core &c = b->core;
    hipStream_t st = (hipStream_t)hip_stream;
    HIPC(hipSetDevice(c.device));
    int rc = c.reset_state(st);
    if (rc != PV_OK) return rc;
    InAddr ia;
    ia.in = d_in;
    ia.stride_c = b->frames;
    ia.stride_s = b->frames * c.C;
    ia.mask = ~0ull;
    ia.len = b->frames;
    // Instrumentation: HIP events around each kernel of every `timing`-th chunk.  An event record costs a few
    // microseconds of stream time, so instrumenting every launch would slow the run it measures by ~10 %.
    InAddr car{};
    car.in = b->d_carrier.p;
    car.stride_c = 0;
    car.stride_s = 0;
    car.mask = ~0ull;
    car.len = (int64_t)b->d_carrier.n;
    // Software pipeline of the phase-locked path (two chunks in flight): the main stream runs the front of chunk i
    // (analysis, match), then the back of chunk i-1 (synthesis, overlap-add); the second stream walks chunk i's
    // rotation chain meanwhile, and has until the back of chunk i -- one more front later -- to finish.
    const bool piped = b->chain_stream != nullptr;
    // (indices first, pointers after the pool has stopped growing: a pointer into ev_pool taken before a later
    // push_back would dangle)
    std::vector<long> ev_index(b->chunks.size(), -1);
    for (size_t ci = 0; ci < b->chunks.size(); ++ci) {
        if (!(b->timing > 0 && (int)(ci % (size_t)b->timing) == (b->timing / 2) % b->timing)) continue;
        const size_t need = b->ev_used + kEvPerChunk;
        bool ok = true;
        while (ok && b->ev_pool.size() < need) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) ok = false;
            else b->ev_pool.push_back(e);
        }
        if (!ok) break;
        ev_index[ci] = (long)b->ev_used;
        b->ev_used = need;
        b->ev_chunk.push_back((int)ci);
    }
    auto events_for = [&](size_t ci) -> hipEvent_t * {
        return ev_index[ci] >= 0 ? &b->ev_pool[(size_t)ev_index[ci]] : nullptr;
    };
    static const bool late_env = [] {
        const char *e = getenv("AUDIOMOD_PV_LATE_CHAIN");
        return e && atoi(e) != 0;
    }();
    const bool late = late_env && b->chain_stream != nullptr && c.use_chain && c.d.resample && c.wave_fft() &&
                      b->res_stream == nullptr && c.three_stage;
    auto launch = [&](size_t ci, hipEvent_t *ev, int part) {
        const auto &ch = b->chunks[ci];
        ChainLaunch cl{};
        if (c.use_chain) {
            cl.slices = b->d_cs.p + ch.cs_begin;
            cl.run_off = b->d_run_off.p + ch.run_begin;
            cl.runs = ch.runs;
            cl.wden = b->d_wden.p;
            cl.wden_hi = b->d_wden_hi.p ? b->d_wden_hi.p : b->d_wden.p;
            cl.res_tiles = b->d_res_tiles.p + ch.res_begin;
            cl.res_otab = b->d_res_otab.p + (size_t)ch.res_begin * kTileOut;
            cl.res_ntiles = ch.res_ntiles;
            cl.res_stream = b->res_stream;
            cl.ev_fused = b->ev_fused[ci & 3];
            cl.ev_res = b->ev_res[ci & 3];
            cl.ev_ring_free = ci >= 2 ? b->ev_res[(ci - 2) & 3] : nullptr;
            cl.late_chain = late;
            cl.out = d_out + ch.k0;
        }
        c.launch_chunk(ia, ch.t0, ch.Tn, b->d_pinc.p + ch.t0, b->d_tiles.p + ch.tile_begin, ch.ntiles, b->d_P.p,
                       b->d_wacc.p + (size_t)ch.tile_begin * c.wacc_pitch,
                       b->d_whisper.p ? b->d_whisper.p + (size_t)ch.t0 * c.C * c.HP : nullptr,
                       c.d.vocoder ? &car : nullptr, d_out, b->plan.out_frames, 0, st, ev, part, b->chain_stream,
                       b->ev_match[ci & 3], b->ev_chain[ci & 3], false, c.use_chain ? &cl : nullptr);
    };
    const size_t nchunks = b->chunks.size();
    if (piped) {
        // the chain stream must not start before the caller's stream has reached this run (state reset, inputs)
        // Fused path with a resampling kernel: three stages -- front of chunk i, resampling of chunk i-2, fused
        // kernel of chunk i-1 -- so that the rotation chain of chunk i (started behind its match kernel) runs
        // beside the resampling kernel, whose workgroups leave it room on every CU, and is done when the fused
        // kernel, which fills the CUs' LDS, starts.
        const bool three = c.use_chain && c.d.resample && c.wave_fft() && b->res_stream == nullptr && c.three_stage;
        std::vector<hipEvent_t *> evs(nchunks, nullptr);
        const bool ahead = three && c.ahead && !late;
        if (ahead) {
            for (size_t ci = 0; ci < nchunks; ++ci) evs[ci] = events_for(ci);
            if (nchunks > 0) launch(0, evs[0], 6); // A(0)
            for (size_t ci = 0; ci < nchunks; ++ci) {
                launch(ci, evs[ci], 7);                                  // M(ci), chain(ci) handed to its stream
                if (ci > 1) launch(ci - 2, evs[ci - 2], 4);              // R(ci-2)   beside the chain
                if (ci + 1 < nchunks) launch(ci + 1, evs[ci + 1], 6);    // A(ci+1)   beside what is left of it
                if (ci > 0) launch(ci - 1, evs[ci - 1], 3);              // F(ci-1)
            }
            if (nchunks > 1) launch(nchunks - 2, evs[nchunks - 2], 4);
            if (nchunks > 0) launch(nchunks - 1, evs[nchunks - 1], 3), launch(nchunks - 1, evs[nchunks - 1], 4);
        } else {
        for (size_t ci = 0; ci < nchunks; ++ci) {
            evs[ci] = events_for(ci);
            launch(ci, evs[ci], 1);
            if (three) {
                if (ci > 1) launch(ci - 2, evs[ci - 2], 4);
                if (ci > 0) launch(ci - 1, evs[ci - 1], 3);
                if (late) launch(ci, evs[ci], 5); // the chain of chunk i starts behind the fused kernel of chunk i-1
            } else if (ci > 0) {
                launch(ci - 1, evs[ci - 1], 2);
            }
        }
        if (three) {
            if (nchunks > 1) launch(nchunks - 2, evs[nchunks - 2], 4);
            if (nchunks > 0) launch(nchunks - 1, evs[nchunks - 1], 3), launch(nchunks - 1, evs[nchunks - 1], 4);
        } else if (nchunks > 0) {
            launch(nchunks - 1, evs[nchunks - 1], 2);
        }
        }
    } else {
        for (size_t ci = 0; ci < nchunks; ++ci) launch(ci, events_for(ci), 0);
    }
    if (b->res_stream && c.use_chain) // the caller synchronises `st`: it has to cover the resampling stream too
        for (size_t ci = nchunks > 2 ? nchunks - 2 : 0; ci < nchunks; ++ci)
            HIPC(hipStreamWaitEvent(st, b->ev_res[ci & 3], 0));
    if ((rc = c.take_launch_error()) != PV_OK) return rc;
    HIPC(hipGetLastError());
    return PV_OK;
}

int pv_batch_kernel_times(pv_batch *b, double ms[PV_NUM_KERNELS], int64_t launches[PV_NUM_KERNELS]) {
    if (!b) return PV_ERR_INVALID_ARG;
    // fold finished event pairs into the accumulators
    const Derived &d = b->core.d;
    const bool bypass = d.robotic || d.whisper || d.constant || d.vocoder;
    const int cm = bypass ? -1 : ((d.cfg.coremode == 1 || d.cfg.coremode == 2) ? d.cfg.coremode : 0);
    for (size_t i = 0; i + kEvPerChunk <= b->ev_used; i += kEvPerChunk) {
        for (int k = 0; k < PV_NUM_KERNELS; ++k) {
            if ((k == PV_K_MATCH || k == PV_K_SEQ) && cm != 1) continue;
            if (k == PV_K_PROP && cm != 0) continue;
            if (k == PV_K_CEPSTRAL && !d.cepstral) continue;
            const bool fused = b->core.use_chain && b->core.wave_fft();
            if (k == PV_K_SYNTH_OLA && !fused) continue;
            if (k == PV_K_SYNTH && fused) continue;
            if (k == PV_K_OLA_RESAMPLE && fused && !d.resample) continue; // the fused kernel emits the output itself
            if (k == PV_K_OLA_RESAMPLE && !b->core.use_chain) {
                const int ci2 = b->ev_chunk[i / kEvPerChunk];
                if (b->chunks[(size_t)ci2].ntiles == 0) continue;
            }
            float t = 0;
            if (hipEventElapsedTime(&t, b->ev_pool[i + 2 * k], b->ev_pool[i + 2 * k + 1]) == hipSuccess) {
                b->acc_ms[k] += t;
                b->acc_n[k] += 1;
            }
        }
    }
    b->ev_used = 0;
    b->ev_chunk.clear();
    for (int k = 0; k < PV_NUM_KERNELS; ++k) {
        if (ms) ms[k] = b->acc_ms[k];
        if (launches) launches[k] = b->acc_n[k];
    }
    return PV_OK;
}

// ---------------------------------------------------------------- streaming
static constexpr int kStreamChunk = 16; // slices per launch group in streaming mode

static int map_out(pv_engine *e);

int pv_create(const pv_config *cfg, int device, pv_engine **out) {
    g_last_error.clear();
    plan_reason_clear();
    if (!cfg || !out) return PV_ERR_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<pv_engine> e(new pv_engine());
    e->core.chain_required = true;
    e->core.fast_arith = g_arith == PV_ARITH_FAST; // (same kernels as the batch engine: see Core::fast_capable)
    {
        const char *ef = getenv("AUDIOMOD_PV_STREAM_FUSE_PHASE"); // =0: match kernel and rotation chain as two launches
        e->core.fuse_phase = !(ef && atoi(ef) == 0);
    }
    int st = e->core.init(*cfg, device, 1, kStreamChunk);
    if (st != PV_OK) return st;
    Core &c = e->core;
    e->planner.reset(new Planner(c.d));
    if (c.use_chain) e->chain.reset(new ChainBuilder(c.d, c.chain_AR, c.chain_smask));
    HIPC(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->ring = next_pow2_i(3 * c.d.N + kStreamChunk * c.d.hop + 16);
    if ((st = e->d_in.alloc((size_t)c.C * e->ring)) != PV_OK) return st;
    HIPC(hipMemset(e->d_in.p, 0, e->d_in.n * sizeof(float)));
    if ((st = e->h_in.alloc((size_t)c.C * e->ring)) != PV_OK) return st;
    // most outputs one group of slices can emit
    // (largest shift increment -- the hop itself in the modes that do not stretch, else the upper clamp
    // lrint(2 * hop * ratio), phasevocoderprocess.cc:394-395 -- through the resampler where there is one)
    const bool fixed_shift = c.d.robotic || c.d.whisper || c.d.constant || c.d.vocoder;
    const double max_shift = fixed_shift ? (double)c.d.hop : 2.0 * c.d.hop * c.d.hs_ratio + 1;
    const double per_slice = c.d.resample ? max_shift * c.d.res_den / c.d.res_num + 2 : max_shift + 2;
    e->out_cap = (int)(kStreamChunk * per_slice) + 64;
    if ((st = e->d_out.alloc((size_t)c.C * e->out_cap)) != PV_OK) return st;
    if ((st = e->h_out.alloc((size_t)c.C * e->out_cap)) != PV_OK) return st;
    if ((st = e->h_flag.alloc(16)) != PV_OK) return st;
    e->h_flag.p[0] = 0;
    {
        const char *es = getenv("AUDIOMOD_PV_STREAM_SYNC");
        e->direct_out = !(es && atoi(es) != 0);
    }
    if ((st = map_out(e.get())) != PV_OK) return st;
    const size_t desc_bytes = 256 * 1024; // grows on demand (ensure_desc)
    if ((st = e->d_desc.alloc(desc_bytes)) != PV_OK) return st;
    if ((st = e->h_desc.alloc(desc_bytes)) != PV_OK) return st;
    if (c.d.vocoder) {
        e->cargen.reset(new CarrierGen((float)cfg->sample_rate, c.d.chord));
        if ((st = e->d_carrier.alloc((size_t)e->ring)) != PV_OK) return st;
        HIPC(hipMemset(e->d_carrier.p, 0, (size_t)e->ring * sizeof(float)));
        if ((st = e->h_carrier.alloc((size_t)e->ring)) != PV_OK) return st;
    }
    if (c.d.whisper) {
        e->rng.reset(new WhisperRng());
        if ((st = e->d_whisper.alloc((size_t)kStreamChunk * c.C * c.HP)) != PV_OK) return st;
        if ((st = e->h_whisper.alloc((size_t)kStreamChunk * c.C * c.HP)) != PV_OK) return st;
    }
    e->outq.resize(c.C);
    *out = e.release();
    return PV_OK;
}

void pv_destroy(pv_engine *e) { delete e; }

int32_t pv_available(const pv_engine *e) { return e ? e->planner->available() : -1; }

int pv_get_info(const pv_engine *e, pv_info *info) {
    if (!e || !info) return PV_ERR_INVALID_ARG;
    fill_info(e->core.d, e->planner->slices(), info);
    return PV_OK;
}

// upload input samples [e->uploaded, upto) of the current call into the device ring
static int upload_until(pv_engine *e, const float *const *in, int64_t call_base, int64_t upto) {
    Core &c = e->core;
    while (e->uploaded < upto) {
        const int64_t pos = e->uploaded;
        const int64_t roff = pos & (e->ring - 1);
        int64_t n = upto - pos;
        if (n > e->ring - roff) n = e->ring - roff;
        for (int ch = 0; ch < c.C; ++ch)
            memcpy(e->h_in.p + (size_t)ch * e->ring + roff, in[ch] + (pos - call_base), (size_t)n * sizeof(float));
        if (e->cargen) { // the carrier advances in lock-step with the input samples
            float *hc = e->h_carrier.p + roff;
            for (int64_t i = 0; i < n; ++i) hc[i] = e->cargen->next();
            HIPC(hipMemcpyAsync(e->d_carrier.p + roff, hc, (size_t)n * sizeof(float), hipMemcpyHostToDevice, e->stream));
        }
        // one strided copy for all channels (rows of n floats, pitch = ring)
        HIPC(hipMemcpy2DAsync(e->d_in.p + roff, (size_t)e->ring * sizeof(float), e->h_in.p + roff,
                              (size_t)e->ring * sizeof(float), (size_t)n * sizeof(float), (size_t)c.C,
                              hipMemcpyHostToDevice, e->stream));
        e->uploaded += n;
    }
    return PV_OK;
}

// (re)size the pinned + device descriptor staging; contents are per launch group, nothing to preserve
static int ensure_desc(pv_engine *e, size_t bytes) {
    if (bytes <= e->h_desc.n) return PV_OK;
    size_t cap = e->h_desc.n ? e->h_desc.n : 64 * 1024;
    while (cap < bytes) cap *= 2;
    HIPC(hipStreamSynchronize(e->stream)); // an earlier group's copy may still read the old buffer
    int st = e->h_desc.alloc(cap);
    if (st != PV_OK) return st;
    return e->d_desc.alloc(cap);
}
static int map_out(pv_engine *e) { // (after h_out / h_flag were allocated)
    if (!e->direct_out) return PV_OK;
    void *p = nullptr;
    HIPC(hipHostGetDevicePointer(&p, e->h_out.p, 0));
    e->out_dev = static_cast<float *>(p);
    HIPC(hipHostGetDevicePointer(&p, e->h_flag.p, 0));
    e->flag_dev = static_cast<uint32_t *>(p);
    return PV_OK;
}
static int ensure_out(pv_engine *e, int64_t cnt) {
    if (cnt <= e->out_cap) return PV_OK;
    int cap = e->out_cap;
    while (cap < cnt) cap *= 2;
    HIPC(hipStreamSynchronize(e->stream));
    int st = e->d_out.alloc((size_t)e->core.C * cap);
    if (st != PV_OK) return st;
    if ((st = e->h_out.alloc((size_t)e->core.C * cap)) != PV_OK) return st;
    e->out_cap = cap;
    return map_out(e);
}

int pv_feed(pv_engine *e, const float *const *in, int32_t n) {
    g_last_error.clear();
    plan_reason_clear();
    if (!e || n < 0 || (n > 0 && !in)) return PV_ERR_INVALID_ARG;
    Core &c = e->core;
    if (e->poisoned) {
        g_last_error = "engine unusable after an earlier failure inside pv_feed: " + e->poison_reason;
        return e->poisoned;
    }
    HIPC(hipSetDevice(c.device));
    const int64_t call_base = e->fed;
    std::vector<SliceRec> fresh;
    // Planning is transactional: a call the planner refuses leaves the engine exactly where it was -- nothing fed,
    // nothing pending -- so the caller may retrieve and try again.
    const Planner::State before = e->planner->save();
    int st = e->planner->feed(n, fresh);
    if (st == PV_OK && !c.use_chain)
        for (const SliceRec &r : fresh)
            if (r.adv == 0) {
                g_last_error = "more output pending than the reference's output ring holds (the reference drops "
                               "slices there): retrieve between calls, or use the fused overlap-add path";
                st = PV_ERR_OUTPUT_OVERRUN;
                break;
            }
    if (st != PV_OK) {
        e->planner->restore(before);
        return st;
    }
    // From here on device state and host bookkeeping move together; a failure in between (a HIP error) cannot be
    // rolled back, so it makes the engine unusable instead of leaving it inconsistent.
    auto fail = [&](int code) {
        e->poisoned = code;
        e->poison_reason = g_last_error.empty() ? pv_strerror(code) : g_last_error;
        return code;
    };
    e->fed += n;
    const int64_t t_new0 = e->t_base + (int64_t)e->recent.size();
    e->recent.insert(e->recent.end(), fresh.begin(), fresh.end());
    const int64_t t_new1 = t_new0 + (int64_t)fresh.size();

    for (int64_t ta = t_new0; ta < t_new1; ta += kStreamChunk) {
        const int64_t tb = (t_new1 - ta) < kStreamChunk ? t_new1 : ta + kStreamChunk;
        const int Tn = (int)(tb - ta);
        // input needed by slices [ta, tb): up to (tb-1)*hop + N.  The pinned staging ring must not be
        // overwritten while a previous async copy still reads it: groups are synchronised below.
        int64_t need = (tb - 1) * (int64_t)c.d.hop + c.d.N;
        if (need > e->fed) need = e->fed;
        if ((st = upload_until(e, in, call_base, need)) != PV_OK) return fail(st);

        const SliceRec &first = e->recent[(size_t)(ta - e->t_base)];
        const SliceRec &last = e->recent[(size_t)(tb - 1 - e->t_base)];
        const int64_t ka = first.K0, kb = last.K0 + last.cnt;
        if ((st = ensure_out(e, kb - ka)) != PV_OK) return fail(st);
        // descriptors of the group, one upload: [pinc Tn int32] then either the tile path's [P list int64][tiles]
        // [window sums + output tables] or the fused path's [ChainSlice Tn][denominators][output table]
        std::vector<OlaTile> tiles;
        std::vector<float> wacc, wden, wden_hi;
        std::vector<ChainSlice> cs;
        std::vector<ResTile> res_tiles;
        std::vector<uint2> res_otab;
        if (c.use_chain) {
            e->chain->begin_launch(first);
            for (int64_t t = ta; t < tb; ++t)
                e->chain->add(e->recent[(size_t)(t - e->t_base)], INT64_MAX, wden, wden_hi);
            std::vector<int32_t> ro;
            e->chain->end_launch(1, cs, ro); // a call's few slices: one run
            if (c.d.resample && kb > ka) c.build_res_tiles(ka, kb, res_tiles, res_otab);
            while (wden.size() & 3) wden.push_back(1.f), wden_hi.push_back(1.f);
            for (int i = 0; i < 4; ++i) wden.push_back(1.f), wden_hi.push_back(1.f);
            if (c.fast_chain()) { // the free-form kernel normalises by multiplying
                for (float &v : wden) v = 1.0f / v;
                for (float &v : wden_hi) v = 1.0f / v;
            }
        } else if (kb > ka) {
            st = c.build_tiles(e->recent, e->t_base, tb, ka, kb, (int32_t)e->t_base, tiles, wacc);
            if (st != PV_OK) return fail(st);
        }
        auto pad16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t nP = c.use_chain ? 0 : (size_t)(tb - e->t_base);
        const bool hi = c.use_chain && e->chain->any_upper_skip;
        const size_t o_pinc = 0, o_ro = pad16((size_t)Tn * 4), o_a = o_ro + 16; // (o_ro: the run offsets {0, Tn})
        const size_t o_b = o_a + (c.use_chain ? pad16(cs.size() * sizeof(ChainSlice)) : pad16(nP * 8));
        const size_t o_c = o_b + (c.use_chain ? pad16(wden.size() * 4) : pad16(tiles.size() * sizeof(OlaTile)));
        const size_t o_d = o_c + (c.use_chain ? (hi ? pad16(wden_hi.size() * 4) : 0) : pad16(wacc.size() * 4));
        const size_t o_e = o_d + (c.use_chain ? pad16(res_tiles.size() * sizeof(ResTile)) : 0);
        const size_t total = o_e + (c.use_chain ? pad16(res_otab.size() * sizeof(uint2)) : 0);
        if ((st = ensure_desc(e, total)) != PV_OK) return fail(st);
        char *hd = e->h_desc.p;
        int32_t *h_pinc = reinterpret_cast<int32_t *>(hd + o_pinc);
        for (int i = 0; i < Tn; ++i) h_pinc[i] = e->recent[(size_t)(ta + i - e->t_base)].phase_inc;
        {
            int32_t *h_ro = reinterpret_cast<int32_t *>(hd + o_ro);
            h_ro[0] = 0, h_ro[1] = (int32_t)cs.size(), h_ro[2] = h_ro[3] = 0;
        }
        if (c.use_chain) {
            memcpy(hd + o_a, cs.data(), cs.size() * sizeof(ChainSlice));
            memcpy(hd + o_b, wden.data(), wden.size() * 4);
            if (hi) memcpy(hd + o_c, wden_hi.data(), wden_hi.size() * 4);
            memcpy(hd + o_d, res_tiles.data(), res_tiles.size() * sizeof(ResTile));
            memcpy(hd + o_e, res_otab.data(), res_otab.size() * sizeof(uint2));
        } else {
            int64_t *h_P = reinterpret_cast<int64_t *>(hd + o_a);
            for (size_t i = 0; i < nP; ++i) h_P[i] = e->recent[i].P;
            memcpy(hd + o_b, tiles.data(), tiles.size() * sizeof(OlaTile));
            memcpy(hd + o_c, wacc.data(), wacc.size() * sizeof(float));
        }
        if (hipMemcpyAsync(e->d_desc.p, hd, total, hipMemcpyHostToDevice, e->stream) != hipSuccess)
            return fail(hip_fail(hipGetLastError(), "descriptor upload", __LINE__));

        if (c.d.whisper) {
            for (int i = 0; i < Tn; ++i)
                for (int ch = 0; ch < c.C; ++ch)
                    for (int k = 0; k <= c.d.hs; ++k)
                        e->h_whisper.p[((size_t)i * c.C + ch) * c.HP + k] = e->rng->next_phase();
            if (hipMemcpyAsync(e->d_whisper.p, e->h_whisper.p, (size_t)Tn * c.C * c.HP * sizeof(float),
                               hipMemcpyHostToDevice, e->stream) != hipSuccess)
                return fail(hip_fail(hipGetLastError(), "whisper upload", __LINE__));
        }
        InAddr ia;
        ia.in = e->d_in.p;
        ia.stride_c = e->ring;
        ia.stride_s = (int64_t)e->ring * c.C;
        ia.mask = (uint64_t)(e->ring - 1);
        ia.len = INT64_MAX;
        InAddr car = ia;
        car.in = e->d_carrier.p;
        car.stride_c = 0;
        car.stride_s = 0;
        ChainLaunch cl{};
        if (c.use_chain) {
            cl.slices = reinterpret_cast<const ChainSlice *>(e->d_desc.p + o_a);
            cl.run_off = reinterpret_cast<const int32_t *>(e->d_desc.p + o_ro);
            cl.runs = 1;
            cl.wden = reinterpret_cast<const float *>(e->d_desc.p + o_b);
            cl.wden_hi = hi ? reinterpret_cast<const float *>(e->d_desc.p + o_c) : cl.wden;
            cl.res_tiles = reinterpret_cast<const ResTile *>(e->d_desc.p + o_d);
            cl.res_otab = reinterpret_cast<const uint2 *>(e->d_desc.p + o_e);
            cl.res_ntiles = (int)res_tiles.size();
            cl.out = e->direct_out ? e->out_dev : e->d_out.p;
        }
        c.launch_chunk(ia, ta, Tn, reinterpret_cast<const int32_t *>(e->d_desc.p + o_pinc),
                       reinterpret_cast<const OlaTile *>(e->d_desc.p + o_b), (int)tiles.size(),
                       reinterpret_cast<const int64_t *>(e->d_desc.p + o_a),
                       reinterpret_cast<const float *>(e->d_desc.p + o_c), e->d_whisper.p,
                       c.d.vocoder ? &car : nullptr, e->direct_out ? e->out_dev : e->d_out.p, e->out_cap, ka, e->stream,
                       nullptr, 0, nullptr, nullptr,
                       nullptr, c.can_single_launch(), c.use_chain ? &cl : nullptr);
        if ((st = c.take_launch_error()) != PV_OK) return fail(st);
        const int64_t cnt = kb - ka;
        hipError_t he = hipSuccess;
        const bool last_group = tb == t_new1;
        if (e->direct_out) {
            // what the call leaves unconsumed goes to the device ring behind the last group's kernels, in front of
            // the flag: one wait per call
            if (last_group && (st = upload_until(e, in, call_base, e->fed)) != PV_OK) return fail(st);
            const uint32_t seq = ++e->flag_seq;
            he = hipStreamWriteValue32(e->stream, e->flag_dev, seq, 0);
            if (he == hipSuccess) {
                // spin on the host copy of the flag; look at the stream every ~4096 polls so that a failed launch ends the
                // wait instead of hanging it
                volatile uint32_t *fl = e->h_flag.p;
                uint32_t spins = 0;
                while (__atomic_load_n(fl, __ATOMIC_ACQUIRE) != seq) {
                    if ((++spins & 0xfffu) == 0) {
                        const hipError_t q = hipStreamQuery(e->stream);
                        if (q != hipSuccess && q != hipErrorNotReady) {
                            he = q;
                            break;
                        }
                        if (q == hipSuccess && __atomic_load_n(fl, __ATOMIC_ACQUIRE) != seq && spins > (1u << 26)) {
                            he = hipErrorUnknown; // the stream is idle and the flag never arrived
                            break;
                        }
                    }
                }
            }
        } else {
            if (cnt > 0)
                he = hipMemcpy2DAsync(e->h_out.p, (size_t)e->out_cap * sizeof(float), e->d_out.p,
                                      (size_t)e->out_cap * sizeof(float), (size_t)cnt * sizeof(float), (size_t)c.C,
                                      hipMemcpyDeviceToHost, e->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        }
        if (he == hipSuccess) he = hipGetLastError();
        if (he != hipSuccess) return fail(hip_fail(he, "streaming launch group", __LINE__));
        for (int ch = 0; ch < c.C && cnt > 0; ++ch) {
            const float *src = e->h_out.p + (size_t)ch * e->out_cap;
            e->outq[(size_t)ch].insert(e->outq[(size_t)ch].end(), src, src + cnt);
        }
        // drop slice records no later group can reference
        const int64_t keep_from = tb - (c.lookback + kStreamChunk + 2);
        if (keep_from > e->t_base) {
            e->recent.erase(e->recent.begin(), e->recent.begin() + (size_t)(keep_from - e->t_base));
            e->t_base = keep_from;
        }
    }
    // everything fed stays needed by later slices (at most 2N unconsumed): park it in the device ring
    if (e->uploaded < e->fed || !e->direct_out) { // (direct output: the last group's flag already covered the upload)
        if ((st = upload_until(e, in, call_base, e->fed)) != PV_OK) return fail(st);
        if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(hip_fail(hipGetLastError(), "stream sync", __LINE__));
    }
    return PV_OK;
}

int32_t pv_retrieve(pv_engine *e, float *const *out, int32_t n) {
    if (!e || n < 0 || (n > 0 && !out)) return -1;
    // never hand out more than the FIFO holds (the planner's count and the FIFO move together; this is the guard
    // against that invariant ever breaking, not a normal path)
    const size_t held = e->outq.empty() ? 0 : e->outq[0].size() - e->outq_head;
    if ((size_t)n > held) n = (int32_t)held;
    const int32_t got = e->planner->retrieve(n);
    for (int ch = 0; ch < e->core.C; ++ch) {
        const std::vector<float> &q = e->outq[(size_t)ch];
        memcpy(out[ch], q.data() + e->outq_head, (size_t)got * sizeof(float));
    }
    e->outq_head += (size_t)got;
    if (e->outq_head > (1u << 16)) {
        for (auto &q : e->outq) q.erase(q.begin(), q.begin() + (long)e->outq_head);
        e->outq_head = 0;
    }
    return got;
}

} // extern "C"
