/* audiomod_pv.h -- C ABI of the MI355X-native phase-vocoder engine (libaudiomod_pv.so).
 *
 * This is the drop-in boundary for the phase-vocoder hot path of tangkk/audiomod.  The
 * reference has no C ABI or plugin loader: its callers use the C++
 * class audiomod::phasevocoder (reference include/dafx/phasevocoder.h:42-117) through
 * modbase / modbase_offline (reference include/dafx/modbase.h:26-66,75-126).  The entry
 * points below are what that class binds to underneath (see include/dafx/phasevocoder.h in
 * this repository for the source-compatible class, and INTEGRATION.md for how a reference
 * maintainer swaps it in).  Plain pointers and sizes only; no C++ or torch types; no
 * exceptions cross this boundary -- every call returns a pv_status.
 *
 * Semantics: one engine instance == one FRESH reference process.  The reference keeps
 * DSP state in process-global statics (phasevocoderprocess.cc:380-384,602,716;
 * phasevocoderimpl.cc:46-62, phasevocoderimpl.h:236-238); here that state is per instance.
 */
#ifndef AUDIOMOD_PV_H
#define AUDIOMOD_PV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mode / coremode values: reference include/dafx/phasevocoder.h:22-36 */
#define PV_MODE_CONSTANT (-1)
#define PV_MODE_NORMAL_SHIFT 0
#define PV_MODE_GENDER_CHANGE 1
#define PV_MODE_FORMANT_PRESERVE 2
#define PV_MODE_VOCODER_ROSENBERG 3
#define PV_MODE_VOCODER_CHORD 4
#define PV_MODE_NORMAL_STRETCH 5
#define PV_MODE_ROBOTIC 6
#define PV_MODE_WHISPER 7
/* Extension, not a value of the reference's enum: pitch shift whose formants are restored by the reference's
 * cepstral formant shift (formantShiftSlice, phasevocoderprocess.cc:925-999, with env_comp = the pitch scale) --
 * code the reference carries but never calls (its call in formantPreserveSlice is commented out, :838).
 * Any fftsize from 128 up (its lifter keeps 60 quefrencies). */
#define PV_MODE_FORMANT_CEPSTRAL 8
#define PV_CORE_NORMAL_PV 0
#define PV_CORE_PHASE_LOCKED 1
#define PV_CORE_INT_RATIO 2

typedef enum pv_status {
    PV_OK = 0,
    PV_ERR_INVALID_ARG = 1,
    PV_ERR_UNSUPPORTED = 2,   /* unknown mode, fftsize above 8192, or a resample ratio whose
                                 per-slice output cap would bind (reference resampler.cc:783) */
    PV_ERR_NO_DEVICE = 3,     /* no usable MI355X / HIP runtime: the product has NO CPU fallback */
    PV_ERR_HIP = 4,           /* a HIP call failed; pv_last_error() has the text */
    PV_ERR_OUTPUT_OVERRUN = 5 /* caller let more than the reference's output ring capacity pile up
                                 (reference phasevocoderprocess.cc:344-364 drops the slice there) */
} pv_status;

/* Constructor arguments of audiomod::phasevocoder (reference include/dafx/phasevocoder.h:54). */
typedef struct pv_config {
    int32_t sample_rate;
    int32_t channels;
    float time_ratio;
    float pitch_semitones;
    int32_t mode;     /* PV_MODE_* */
    int32_t coremode; /* PV_CORE_* */
    int32_t fftsize;  /* rounded up to a power of two like the reference (phasevocoderimpl.cc:177-181) */
    int32_t hopsize;  /* 0 = auto, the only value the reference CLI passes */
} pv_config;

/* Derived constants (reference Impl::calculateSizes, phasevocoderimpl.cc:169-263, and the Speex set-up,
 * resampler.cc:740-770 + resample.c:661-913). */
typedef struct pv_info {
    int32_t fftsize, hop_in, hop_out_nominal, outbuf_capacity;
    float pitch_scale, hs_ratio;
    int32_t int_ratio, resample;
    uint32_t res_num, res_den;
    int32_t res_filt_len, res_oversample, res_interp;
    int64_t slices;        /* slices processed so far */
    int64_t bytes_per_slice; /* algorithmic HBM bytes per slice, SURVEY.md section 8(d): 4*(3N+7H+2s+h) */
} pv_info;

/* Arithmetic of the synthesis side (process-wide setting, read when an engine is created; default PV_ARITH_FAST,
 * or PV_ARITH_EXACT when the environment has AUDIOMOD_PV_EXACT=1).
 *   Everything up to and including the phase propagation -- window, forward FFT, magnitudes, atan2f, peak picking and
 *   matching, the float / double phase chain -- always follows the reference's x86 build operation for operation
 *   (separate multiplies and adds, its libm's atan2f): the propagation is discontinuous in those values, so nothing
 *   short of the same bits is safe there.  Behind it the output is a continuous function of its inputs, and
 *   BASELINE.json's contract is 1e-4 RMS:
 *   PV_ARITH_EXACT  resynthesis, overlap-add, normalisation and resampling also in the reference's operation order
 *                   (bit-identical to the reference except for the sine / cosine of the resynthesis; ROBOTIC mode
 *                   bit-identical end to end);
 *   PV_ARITH_FAST   where a free-form kernel exists -- the plain pitch-shift / stretch modes in every core mode at fft
 *                   512 ... 4096, the formant / gender modes at fft 2048 -- resynthesis, normalisation and resampling may
 *                   fuse multiply-adds, regroup sums, skip phase wraps and use the hardware's sine / cosine
 *                   (measured: 1e-8 ... 5e-8 RMS against the reference).  Such a configuration then always takes the
 *                   fused overlap-add path, in the single-stream engine and in batches of any size alike, so every
 *                   engine runs the same kernels and they agree with each other bit for bit under either setting.
 *                   Every other mode and size computes as PV_ARITH_EXACT. */
#define PV_ARITH_FAST 0
#define PV_ARITH_EXACT 1
int pv_set_arithmetic(int arith);
int pv_get_arithmetic(void);

const char *pv_strerror(int status);
const char *pv_last_error(void);
/* number of visible HIP devices that are gfx950; <= 0 means the library cannot run */
int pv_device_count(void);

/* ----------------------------------------------------------------------------------------------
 * Host planner only (no GPU needed): the reference's data-independent integer behaviour.
 * Feeds `ncalls` blocks of sizes n[i] through the scheduling logic of Impl::processNormal
 * (phasevocoderimpl.cc:340-369), retrieving everything available after each call like the
 * reference CLI (main/main.cc:484-491); writes the per-call availability to avail[i].  If shift /
 * phase are non-NULL they receive the per-slice increments of calculateIncrements
 * (phasevocoderprocess.cc:412-489), up to max_slices; *nslices gets the slice count.
 * -------------------------------------------------------------------------------------------- */
int pv_plan_simulate(const pv_config *cfg, const int32_t *n, int32_t ncalls, int32_t *avail, int32_t *shift,
                     int32_t *phase, int64_t max_slices, int64_t *nslices, pv_info *info);
/* The first n phases WHISPER mode assigns in a fresh reference process (glibc rand() from its default seed;
 * reference phasevocoderprocess.cc:814-822), in draw order: slice-major, channel, bin 0..N/2. */
int pv_plan_whisper_phases(int64_t n, float *out);
/* The data-independent float tables the planner derives for a configuration, for tests and inspection (no GPU
 * needed): PV_TABLE_WINDOW = the Hann window (windowfunc.h:159-169); PV_TABLE_SINC = the Speex Q4 filter table
 * (resample.c:661-775; empty when the configuration does not resample); PV_TABLE_CARRIER = the first `max` samples of
 * the vocoder carrier (gen/rosenberg.cc, rosenbergchord.cc).  Writes min(len, max) floats, returns the table's
 * length (PV_TABLE_CARRIER: max), or a negative pv_status. */
#define PV_TABLE_WINDOW 0
#define PV_TABLE_SINC 1
#define PV_TABLE_CARRIER 2
int64_t pv_plan_table(const pv_config *cfg, int which, float *out, int64_t max);

/* ----------------------------------------------------------------------------------------------
 * Streaming engine: ONE stream of cfg->channels planar channels, host buffers in and out.
 * Replaces phasevocodercore::{processNormal, numsamples_available, retrieve}
 * (reference src/phasevocoder/phasevocoderinterface.h:24-172; phasevocoderimpl.cc:340-369;
 * phasevocoderprocess.cc:1240-1284), i.e. what audiomod::phasevocoder::processInData /
 * getOutData / processBlock call (reference src/phasevocoder/phasevocoder.cc:87-183).
 * -------------------------------------------------------------------------------------------- */
typedef struct pv_engine pv_engine;

int pv_create(const pv_config *cfg, int device, pv_engine **out);
void pv_destroy(pv_engine *e);
/* == processNormal(in, n): in[c] points at n host floats of channel c.  Synchronous. */
int pv_feed(pv_engine *e, const float *const *in, int32_t n);
/* == numsamples_available() */
int32_t pv_available(const pv_engine *e);
/* == retrieve(out, n): copies min(n, available) frames per channel; returns the count (>= 0) */
int32_t pv_retrieve(pv_engine *e, float *const *out, int32_t n);
int pv_get_info(const pv_engine *e, pv_info *info);

/* ----------------------------------------------------------------------------------------------
 * Batch engine: `nstreams` independent streams of identical configuration and length, input and
 * output resident in device memory (HBM).  Equivalent, per stream, to driving the reference CLI
 * loop (main/main.cc:471-510) with `block`-frame calls: flush != 0 -> feed zeros until `frames`
 * output frames exist and truncate to `frames` (pitch-shift modes); flush == 0 -> no flush
 * (time_stretch).  This is the throughput path bench.py measures.
 *   d_in  : [nstreams][channels][frames]      float32, device
 *   d_out : [nstreams][channels][out_frames]  float32, device (out_frames = pv_batch_out_frames)
 * pv_batch_run enqueues all work on `hip_stream` (a hipStream_t passed as void*; NULL = the
 * default stream) and returns without synchronising.
 * -------------------------------------------------------------------------------------------- */
typedef struct pv_batch pv_batch;

int pv_batch_create(const pv_config *cfg, int32_t nstreams, int64_t frames, int32_t block, int32_t flush,
                    int device, pv_batch **out);
void pv_batch_destroy(pv_batch *b);
int64_t pv_batch_out_frames(const pv_batch *b);
int64_t pv_batch_slices(const pv_batch *b); /* slices per channel per stream */
int32_t pv_batch_launches(const pv_batch *b); /* launches of each kernel per pv_batch_run (= chunks of slices) */
/* 1 when pv_batch_run software-pipelines the phase-locked path: the rotation chain (PV_K_SEQ) of chunk i then runs on a
 * second HIP stream beside the synthesis / overlap-add of chunk i-1 and the analysis / match of chunk i+1, so its
 * measured duration overlaps the other kernels' (environment AUDIOMOD_PV_PIPELINE=0 turns it off) */
int32_t pv_batch_pipelined(const pv_batch *b);
int pv_batch_get_info(const pv_batch *b, pv_info *info);
int pv_batch_run(pv_batch *b, const float *d_in, float *d_out, void *hip_stream);
/* Optional per-kernel timing of the NEXT pv_batch_run calls (HIP events on the run's stream).
 * pv_batch_enable_timing(b, n): n = 0 off; n >= 1 instruments every n-th chunk (n = 1: every launch; an event
 * record costs stream time, so full instrumentation slows the run by about 10 %).  After synchronising the
 * stream, pv_batch_kernel_times returns, for each of PV_NUM_KERNELS kernels, the summed device time in ms and
 * the number of instrumented launches since timing was enabled, and resets the accumulation window. */
#define PV_NUM_KERNELS 8
#define PV_K_ANALYZE 0      /* window + forward real FFT + polar (+ peak picking) */
#define PV_K_MATCH 1        /* phase-locked: peak matching, parallel part */
#define PV_K_SEQ 2          /* phase-locked: per-peak rotation chain, sequential over slices */
#define PV_K_PROP 3         /* coremode 0: per-bin phase recurrence */
#define PV_K_SYNTH 4        /* phase application + freqComp + inverse real FFT + window */
#define PV_K_OLA_RESAMPLE 5 /* overlap-add + normalise + resample */
#define PV_K_CEPSTRAL 6     /* PV_MODE_FORMANT_CEPSTRAL: cepstral envelope shift of the magnitudes */
#define PV_K_SYNTH_OLA 7    /* PV_K_SYNTH and PV_K_OLA_RESAMPLE fused: synthesis frames overlap-added in LDS (fft 512 ...
                               4096; the default -- AUDIOMOD_PV_FUSED=0 brings the two separate kernels back) */
int pv_batch_enable_timing(pv_batch *b, int on);
int pv_batch_kernel_times(pv_batch *b, double ms[PV_NUM_KERNELS], int64_t launches[PV_NUM_KERNELS]);
const char *pv_kernel_name(int k);

/* ----------------------------------------------------------------------------------------------
 * Host-staged many-stream job.  The reference's callers hold their audio in host memory (planar
 * float buffers filled from 16-bit WAV data: main/main.cc:152-162,484-491; main/wavfile.cc:733-755,
 * 1295-1306,1334-1342), so this is the batch engine with the staging included: `nstreams` streams in
 * host memory, processed in groups of `streams_per_group` whose host-to-device copy, kernels and
 * device-to-host copy overlap (three groups in flight on three HIP streams).  On the wire the
 * samples are float32, or int16 exactly as the reference's WAV reader / writer convert them
 * (in: int16 * 1/32768; out: saturate(x * 32768, -32768, 32767) truncated toward zero).
 *   host_in  : [nstreams][channels][frames]      float32 or int16
 *   host_out : [nstreams][channels][out_frames]  same type
 * Both should be page-locked (pv_host_alloc) for the copies to run at PCIe rate and asynchronously.
 * pv_hostio_run is synchronous: it returns when host_out is complete.
 * -------------------------------------------------------------------------------------------- */
/* Diagnostics: the analysis kernels' atan2f (libm's algorithm restated, audiomod_amd/csrc/pv_atan2f.h, device build)
 * evaluated on host arrays of FINITE values -- tests compare it with the C library's atan2f bit for bit. */
int pv_debug_atan2f(const float *y, const float *x, float *out, int64_t n, int device);
/* ... and the polar conversion of the wave-per-frame analysis kernels (FFT.cc:2623-2630 mag = sqrtf(re^2 + im^2),
 * phase = atan2f(im, re): table-driven atan2f, range-tested short division and square root) on FINITE values. */
int pv_debug_polar(const float *im, const float *re, float *phase, float *mag, int64_t n, int device);
/* ... and its short square root against the correctly rounded one on EVERY float whose bit pattern lies in
 * [first_bits, first_bits + count): the number of mismatches and the first offending pattern. */
int pv_debug_sqrt_sweep(uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint32_t *first_bad, int device);

#define PV_WIRE_F32 0
#define PV_WIRE_I16 1
typedef struct pv_hostio pv_hostio;
int pv_hostio_create(const pv_config *cfg, int32_t nstreams, int64_t frames, int32_t block, int32_t flush, int device,
                     int32_t streams_per_group, int32_t wire, pv_hostio **out);
void pv_hostio_destroy(pv_hostio *h);
int64_t pv_hostio_out_frames(const pv_hostio *h);
int pv_hostio_run(pv_hostio *h, const void *host_in, void *host_out);
void *pv_host_alloc(size_t bytes); /* page-locked host memory (NULL on failure) */
void pv_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
