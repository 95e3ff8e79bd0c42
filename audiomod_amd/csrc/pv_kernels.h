// pv_kernels.h -- launch interface between the host engine (pv_engine.cc) and the gfx950 kernels
// (pv_kernels.hip).  Plain structs; everything is sized and validated on the host before launch.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace pv {

constexpr int kFftThreads = 256;   // workgroup size of the analysis / synthesis kernels (4 waves)
constexpr int kTileOut = 256;      // outputs per workgroup of the OLA+resample kernel
constexpr int kMaxTileFrames = 64; // frames that can overlap one OLA tile (checked on the host)

struct DevTables {
    int N, hs, H, HP, nc, log2nc;
    int nstages;
    int radix[16], log2m[16], fstride[16];
    const int32_t *perm;  // [nc] out -> in
    const int32_t *iperm; // [nc] in -> out
    const float2 *tw_fwd, *tw_inv, *st_fwd, *st_inv;
    const float *window; // [N]
};

// Input addressing: sample `a` (absolute frame index of the stream) of stream s, channel c lives at
//   in[s*stride_s + c*stride_c + (a & mask)]  and reads as 0 when a >= len (zero flush).
struct InAddr {
    const float *in;
    int64_t stride_s, stride_c;
    uint64_t mask;
    int64_t len;
};

struct AnalyzeArgs {
    DevTables tb;
    InAddr ia;
    int hop;
    int64_t t0;  // first slice of this launch
    int Tn;      // slices in this launch
    int Tc;      // slice pitch of the mag/phase buffers
    int rows;    // S*C
    float *mag;  // [rows][Tc][HP]
    float *phase;
};

struct PhaseArgs {
    int N, hs, HP, C, hop, coremode;
    double two_pi_hop;
    int64_t t0;
    int Tn, Tc;
    const int32_t *phase_inc; // [Tn] phaseIncrement of each slice of this launch
    float *mag;               // [S*C][Tc][HP]
    float *phase;             // in: analysis phase, out: synthesis phase (in place)
    // persistent per-stream state (global memory, loaded to LDS at kernel start, stored at end)
    float *st_prev_phase; // [S][C][hs]
    float *st_prev_out;   // [S][C][hs]
    int32_t *st_peaks;    // [S][pkmax]
    int32_t *st_npeaks;   // [S]
    int pkmax;
};

struct SynthArgs {
    DevTables tb;
    int hop;
    double two_pi_hop;
    int do_freq_comp;
    float freq_comp, fixed_gain, inv_n;
    int robotic;
    int64_t t0;
    int Tn, Tc, rows;
    const float *mag;
    const float *phase;
    float *frames; // [rows][FR][N] ring of windowed synthesis frames
    int FR;        // power of two
};

// One workgroup of the OLA+resample kernel produces outputs [k0, k0+kcnt) of every (stream, channel).
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
    const float *window;
    float win_gain;
    const OlaTile *tiles;
    const int64_t *P; // OLA positions of candidate slices, indexed tile.p_off + j
    int ntiles;
    // resampler
    int resample, interp;
    uint32_t num, den;
    int filt_len, oversample;
    const float *sinc;
    int sinc_len;
    int lds_floats; // capacity of the OLA tile in LDS
    // output
    float *out;
    int64_t out_stride_row; // floats between consecutive (stream,channel) rows
    int64_t k_base;         // out index = k - k_base
};

void launch_analyze(const AnalyzeArgs &a, hipStream_t st);
void launch_phase(const PhaseArgs &a, int nstreams, hipStream_t st);
void launch_synth(const SynthArgs &a, hipStream_t st);
void launch_ola(const OlaArgs &a, hipStream_t st);
size_t phase_lds_bytes(int hs, int C, int pkmax);
int phase_threads(int hs);

} // namespace pv
