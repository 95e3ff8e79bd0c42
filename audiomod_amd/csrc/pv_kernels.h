// pv_kernels.h -- launch interface between the host engine (pv_engine.cc) and the gfx950 kernels
// (pv_kernels.hip).  Plain structs; everything is sized and validated on the host before launch.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace pv {

constexpr int kFftThreads = 256;   // workgroup size of the analysis / synthesis kernels (4 waves)
constexpr int kTileOut = 256;      // outputs (= threads) per workgroup of the OLA+resample kernel
constexpr int kMaxTileFrames = 128; // frames that can overlap one OLA tile (checked on the host; the small FFT
                                    // sizes with a large pitch scale reach ~70 at the automatic hop)

struct DevTables {
    int N, hs, H, HP, nc, log2nc;
    int nstages;
    int radix[16], log2m[16], fstride[16];
    const int32_t *perm;  // [nc] out -> in
    const int32_t *iperm; // [nc] in -> out
    const float2 *tw_fwd, *tw_inv, *st_fwd, *st_inv;
    const float4 *twl_fwd, *twl_inv; // lane-major twiddles of the wave-FFT passes 1 and 2 (pv_wavefft.h), nc 1024 / 2048
    const float *window; // [N]
    const float *window_sh; // [4][N + 8]: window_sh[d][j] = window[j - d] (0 outside), for 16-byte aligned frame loads
};

// Input addressing: sample `a` (absolute frame index of the stream) of stream s, channel c lives at
//   in[s*stride_s + c*stride_c + (a & mask)]  and reads as 0 when a >= len (zero flush).
struct InAddr {
    const float *in;
    int64_t stride_s, stride_c;
    uint64_t mask;
    int64_t len;
};

// Slice-indexed planes (mag, phase, peaks, ...) are rings of TR = Tc + 1 slots per (stream, channel) row:
// slot(t) = t % TR, so the last slice of the previous launch stays readable (the phase recurrences look
// one slice back).
enum : int { kModeInit = 0, kModeProp = 1, kModeLock = 2 };

struct AnalyzeArgs {
    DevTables tb;
    InAddr ia;
    int hop;
    int64_t t0; // first slice of this launch
    int s0;     // ring slot of t0 (= t0 % TR, computed on the host: no 64-bit modulo in the kernels)
    int Tn;     // slices in this launch
    int TR;     // ring slots
    int rows;   // S*C
    int PKP;    // pitch of the peak lists
    int find_peaks;
    int split;       // 4096-point frames on two waves each (tb.twl_fwd then holds WF2048S's lane table)
    float *mag;      // [rows][TR][HP]
    float *phase;    // [rows][TR][HP] analysis phase
    uint16_t *peaks; // [rows][TR][PKP]
    int32_t *npk;    // [rows][TR]
};

// what the sequential kernel needs for one spectral peak of a phase-locked step
struct PeakRec {
    float adv;     // (peak_delta_phi * phaseIncrement) / hop
    float a2;      // analysis phase at the peak bin
    float a1;      // previous analysis phase (same channel) at the matched previous peak bin
    uint32_t p1r1; // p1 | (region index of p1 in the previous same-channel step) << 16
};

struct MatchArgs {
    int N, hs, HP, PKP, C, hop, TR, rows, Tn, s0;
    double two_pi_hop;
    int64_t t0;
    const int32_t *phase_inc; // [Tn]
    const float *phase;
    const uint16_t *peaks;
    const int32_t *npk;
    PeakRec *recs;  // [rows][TR][PKP]
    int32_t *modes; // [rows][TR]
};

struct SeqArgs {
    int N, hs, HP, PKP, C, hop, TR, rows, Tn, s0;
    int high_prio; // s_setprio(3): win every issue arbitration (when the chain's result is what everybody waits for)
    int narrow;    // three peaks per lane: a third of the waves per row (to share a CU with the fused kernel)
    double two_pi_hop;
    int64_t t0;
    const int32_t *phase_inc;
    const float *phase;
    const uint16_t *peaks;
    const int32_t *npk;
    const PeakRec *recs;
    const int32_t *modes;
    float *rot;      // [rows][TR][PKP] rotation of each peak region (LOCK steps)
    float *outphase; // [rows][TR][HP]  per-bin output phase (INIT / PROP steps)
    // persistent per-row state
    int32_t *st_kind; // [rows] 0 = zero state, 1 = po_full valid, 2 = lazily locked (st_rot valid)
    float *st_rot;    // [rows][PKP]
    float *st_po;     // [rows][hs]
};

// coremode 0: per-bin recurrence, one thread per bin, streaming over the slices of the launch
struct PropArgs {
    int N, hs, HP, C, hop, TR, rows, Tn, s0;
    double two_pi_hop;
    int64_t t0;
    const int32_t *phase_inc;
    const float *phase;
    float *outphase;
    float *st_pp; // [rows][hs]
    float *st_po; // [rows][hs]
};

struct SynthArgs {
    DevTables tb;
    int hop, C;
    double two_pi_hop;
    int do_freq_comp;
    float freq_comp, fixed_gain, inv_n;
    int robotic;  // output phase = 0
    int passthru; // CONSTANT mode: output phase = analysis phase
    const float *whisper; // WHISPER mode: [Tn][C][HP] phases drawn on the host (else null)
    // channel vocoder: carrier planes [TR][HP] (one row, shared by every stream and channel); band_len < 0 = off
    int voc_band_len;
    const float *cmag;
    const float *cphase;
    int coremode; // 0: phases from outphase; 1: per-step mode (rot / outphase); 2: phase * inc / hop
    int64_t t0;
    int s0;
    int Tn, TR, rows, PKP;
    const int32_t *phase_inc;
    const float *mag;
    const float *phase;
    const float *outphase;
    const uint16_t *peaks;
    const int32_t *npk;
    const int32_t *modes;
    const float *rot;
    float *frames; // [rows][FR][N] ring of windowed synthesis frames
    int FR;        // power of two
};

// One workgroup of the OLA+resample kernel produces outputs [k0, k0+kcnt) of every (stream, channel).
// cepstral formant shift of the magnitude planes (extension mode PV_MODE_FORMANT_CEPSTRAL)
struct CepstralArgs {
    DevTables tb;
    int Tn, TR, rows, s0;
    float env_comp; // the envelope is read at bin lrint(k * env_comp)
    float inv_n;    // float(1.0 / N)
    float *mag;     // [rows][TR][HP], modified in place
};

struct OlaTile {
    int64_t k0;      // first output index (resampled domain, or OLA domain when not resampling)
    int64_t n_lo;    // first OLA-stream sample the tile needs (may be negative: zero history)
    int32_t kcnt;    // outputs in the tile (<= kTileOut)
    int32_t n_cnt;   // OLA samples needed
    int32_t t_first; // first slice whose frame may overlap [n_lo, n_lo+n_cnt)
    int32_t t_cnt;   // number of candidate slices (<= kMaxTileFrames)
    int32_t p_off;   // offset of t_first's entry in the launch's P list
    int32_t pad;
};

struct OlaArgs {
    int N, rows, FR;
    const float *frames;
    const OlaTile *tiles;
    const int64_t *P; // OLA positions of candidate slices, indexed tile.p_off + j
    int ntiles;
    // resampler
    int resample, interp;
    uint32_t num, den;
    int filt_len, oversample;
    const float *sinc;
    int sinc_len;
    const float4 *tab4; // interpolated mode: [oversample][filt_len + 1] expanded coefficient rows
    const float *wacc;  // [ntiles][wacc_pitch]: window-sum denominators of each tile's OLA samples, then (at
                        // float offset otab_off) the tile's output table: per output two 32-bit words,
                        // { x offset in the tile | sub-sample offset << 16, bits of the interpolation fraction }
    int wacc_pitch, otab_off;
    int lds_floats; // capacity of the OLA tile in LDS
    int tab_bytes;  // LDS bytes of the resampler coefficient table (16-byte multiple)
    // output
    float *out;
    int64_t out_stride_row; // floats between consecutive (stream,channel) rows
    int64_t k_base;         // out index = k - k_base
};

// ---------------------------------------------------------------------------------------------------------------
// Fused synthesis + overlap-add ("chain" kernel): one workgroup per (stream, channel) row walks the launch's slices
// in order with the reference's own streaming state -- the output accumulator (synthesiseSlice / writeSlice,
// phasevocoderprocess.cc:1057-1073,1140-1194) as a ring in LDS -- so the synthesis frames never travel through HBM.
// Wave w synthesises slices w, w + W, w + 2W, ... of the row; frames are added into the accumulator strictly in
// slice order (a turn counter in LDS), which keeps the reference's summation order.  What a slice completes,
// [P_t, P_t + adv), is divided by its window sum and is the output (no resampling) or goes to a small per-row ring
// in HBM from which pv_resample_kernel produces the output.  Everything about a slice that does not depend on the
// data is planned on the host, once for all rows.
// ---------------------------------------------------------------------------------------------------------------
struct ChainSlice {
    int32_t acc_pos;  // (P_t mod AR): accumulator-ring position of the frame's first sample
    int32_t str_pos;  // (P_t & smask): stream-ring position of the first sample this slice finalises
    int32_t adv;      // samples finalised after this slice's frame = its shift increment; 0 when the reference
                      // drops the slice (output ring full, phasevocoderprocess.cc:344-364): the frame stays in the
                      // accumulator and the next frame lands on the same position
    int32_t wden_off; // offset of their window-sum denominators in ChainArgs::wden (laid out by ring quads: entry 0
                      // belongs to sample P_t - (P_t mod 4))
    int32_t k_off;    // not resampling: first output (= finalised sample) of the slice, relative to the launch's
    int32_t kcnt;     // ... and how many of its samples are outputs (the CLI truncates the last ones)
    int32_t flags;    // bit 0: channels > 0 do not add this frame (CONSTANT-mode overrun: processOneSliceConstant
                      // returns after channel 0, phasevocoderprocess.cc:139-150); bit 1: warm-up slice of a later
                      // run -- added and finalised like any other, but its samples are not emitted
    int32_t tl;       // which slice of the launch this is (planes, frame ring)
};

struct ChainArgs {
    int N, rows, C, Tn;
    int AR;     // accumulator ring (floats, multiple of 4, >= N + 4)
    int smask;  // stream ring size - 1 (power of two minus one)
    int waves;  // W: waves per workgroup (= slices of a row in flight)
    int diag;   // always 0 in the product library (a value the compiler cannot see through: pv_kernels.hip
                // PV_CHAIN_DIAG); -DPV_DIAG builds set it from AUDIOMOD_PV_CHAIN_DIAG for elimination timings
    // A row's slices are processed by `runs` workgroups: run r walks slices[run_off[r] .. run_off[r + 1]), a list
    // that begins with the frames before its range whose tails reach into it (flag bit 1: rebuilt, not emitted)
    int runs;
    const int32_t *run_off;   // [runs + 1]
    const ChainSlice *slices; // run lists, concatenated
    const float *wden;        // denominators, indexed wden_off + quad-relative sample
    const float *wden_hi;     // ... for channels > 0 (differs from wden only after a CONSTANT-mode overrun)
    // Ring images carried between launches: two halves of [rows][AR] each.  Run 0 of a row READS half (acc_sel & 1),
    // the row's LAST run WRITES the other half, and the engine flips the bit per launch: with one buffer the last
    // run's workgroup could overwrite the image before a late-starting run 0 of the same launch had read it (nothing
    // orders the workgroups of one launch; seen in round 2 as a rare, deterministic wrong value).  acc_sel bit 1: the
    // launch starts the stream (slice 0) -- the accumulator is all zeros by definition (channelinfo.cc:93-115) and
    // nothing is read.  (One pointer and one int rather than two pointers and a flag: the fused kernel is short of
    // scalar registers, and three more of them cost it 400 v_readlane reloads.)
    float *st_acc;
    int acc_sel;
    float *stream;            // [rows][smask + 1] normalised overlap-add stream (resampling configurations)
    int resample;
    // output (not resampling)
    float *out;
    int64_t out_stride_row;
    // frame source when the synthesis runs as its own kernel (FFT sizes without a wave-per-frame transform)
    const float *frames; // [rows][FR][N]
    int FR;
    int64_t t0;
    int fast; // host-side only: pick the free-form (PV_ARITH_FAST) kernel where one exists; wden then holds reciprocals
};
bool synth_chain_has_fast(const SynthArgs &s);

// pv_resample_kernel: one workgroup = 256 outputs of two rows
struct ResTile {
    int64_t k0;   // first output
    int64_t n_lo; // first stream sample its windows need (negative: zero history)
    int32_t kcnt; // outputs in the tile (<= kTileOut)
    int32_t n_cnt; // stream samples it needs
};
struct ResArgs {
    int rows, ntiles, smask;
    const float *stream;
    const ResTile *tiles;
    const uint2 *otab; // [ntiles][kTileOut]: { first tap's offset in the tile | sub-sample offset << 16, fraction bits }
    int interp, filt_len, oversample, sinc_len;
    const float *sinc;
    const float4 *tab4;
    int lds_floats, tab_bytes;
    float *out;
    int64_t out_stride_row, k_base;
    int fast; // arithmetic free within the 1e-4 RMS contract (pv_resample_fast_kernel) instead of the reference's order
};
void launch_resample(const ResArgs &a, hipStream_t st);
bool launch_phase(const MatchArgs &m, const SeqArgs &a, hipStream_t st); // single-stream engine: match + chain in one launch
size_t chain_lds_bytes(const ChainArgs &a, int nc_wave /* 0: frames from HBM */);
void launch_synth_chain(const SynthArgs &s, const ChainArgs &c, hipStream_t st); // nc 1024 / 2048
void launch_frames_chain(const ChainArgs &c, hipStream_t st);                    // any size, frames from HBM

// Everything one streaming call needs, for the single-launch kernel of the drop-in path (pv_stream_kernel): the
// argument blocks of the stages it chains.  coremode: 1 = match + chain, 0 = per-bin propagation, anything else =
// no phase stage (coremode 2 and the modes that bypass it).
struct StreamArgs {
    AnalyzeArgs aa;
    MatchArgs ma;
    SeqArgs qa;
    PropArgs pa;
    CepstralArgs ca;
    SynthArgs sa;
    OlaArgs oa;
    int coremode, cepstral;
};
bool stream_kernel_supported(const StreamArgs &s);  // wave-FFT sizes only (nc 1024 / 2048)
void launch_stream(const StreamArgs &s, hipStream_t st);

bool lds_starts_at_zero(); // build invariant of the wave-per-frame analysis kernels (see pv_kernels.hip)
void launch_analyze(const AnalyzeArgs &a, hipStream_t st);
void launch_match(const MatchArgs &a, hipStream_t st);
void launch_seq(const SeqArgs &a, hipStream_t st);
void launch_prop(const PropArgs &a, hipStream_t st);
void launch_synth(const SynthArgs &a, hipStream_t st);
void launch_cepstral(const CepstralArgs &a, hipStream_t st); // nc 1024 / 2048 only
void launch_ola(const OlaArgs &a, hipStream_t st);

} // namespace pv
