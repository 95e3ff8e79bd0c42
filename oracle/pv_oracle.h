/* oracle/pv_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's phase-vocoder hot path
 * (tangkk/audiomod, /root/reference/src/phasevocoder/ and the FFT / window /
 * resampler primitives under it).  It is the parity
 * checker for the HIP engine and the "port" CPU baseline.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product library (audiomod_amd/lib/libaudiomod_pv.so) never links it.
 *
 * Pinning: bit-exact against the real reference compiled into oracle/_ref/ (oracle/ref.mk)
 * and against the golden vectors under tests/golden/ (tools/make_golden.py).
 */
#ifndef PV_ORACLE_H
#define PV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* modes / coremodes: /root/reference/include/dafx/phasevocoder.h:22-36 */
enum { PVO_CONSTANT = -1, PVO_NORMAL_SHIFT = 0, PVO_GENDER_CHANGE = 1, PVO_FORMANT_PRESERVE = 2,
       PVO_VOCODER_ROSENBERG = 3, PVO_VOCODER_CHORD = 4, PVO_NORMAL_STRETCH = 5, PVO_ROBOTIC = 6, PVO_WHISPER = 7,
       /* extension (not a reference mode): pitch shift with the reference's unreachable cepstral formant shift
        * (formantShiftSlice, env_comp = pitch scale) in the place formantPreserveSlice's comment marks for it */
       PVO_FORMANT_CEPSTRAL = 8 };
enum { PVO_NORMAL_PV = 0, PVO_PHASE_LOCKED = 1, PVO_INT_RATIO = 2 };

typedef struct pvo_config {
    int sample_rate;
    int channels;
    float time_ratio;
    float pitch_semitones;
    int mode;
    int coremode;
    int fftsize;
    int hopsize; /* 0 = auto (the only value the reference CLI ever passes) */
} pvo_config;

typedef struct pvo_info {
    int fftsize, hop_in, hop_out_nominal, outbuf_capacity;
    float pitch_scale, hs_ratio;
    int int_ratio;            /* isIntRatio() */
    int resample;             /* 1 if writeSlice resamples (pitch_scale != 1) */
    unsigned res_num, res_den;/* speex num_rate / den_rate after gcd (0 until first use) */
    int res_filt_len, res_oversample, res_interp; /* res_interp: 1 interpolated-sinc, 0 direct table */
    long slices;              /* slices processed so far (per channel) */
} pvo_info;

typedef struct pvo pvo;

pvo *pvo_create(const pvo_config *cfg);
void pvo_destroy(pvo *h);
/* == phasevocodercore::Impl::processNormal (phasevocoderimpl.cc:340-369); returns numsamples_available(), or
 * PVO_UNDEFINED once a slice's shift increment exceeded the FFT size (the reference's writeSlice then overflows
 * its accumulators: undefined behaviour, reachable only with a caller-chosen hop -- nothing to be on par with) */
#define PVO_UNDEFINED (-2)
int pvo_process(pvo *h, const float *const *in, int n);
int pvo_available(const pvo *h);
/* == Impl::retrieve (phasevocoderprocess.cc:1266-1284); returns frames read per channel */
int pvo_retrieve(pvo *h, float *const *out, int n);
void pvo_get_info(const pvo *h, pvo_info *info);
/* per-slice shift increments recorded so far (for pinning the host planner); returns count copied */
long pvo_get_increments(const pvo *h, int *shift, int *phase, long max);

/* unit-level entry points (pinned by tests/golden KATs) */
void pvo_hann(int n, float *w, float *area);
void pvo_forward_polar(int n, const float *in, float *mag, float *phase);
void pvo_inverse_polar(int n, const float *mag, const float *phase, float *out);
/* the reference's (unreachable) cepstral formant shift of one slice's magnitudes, in place:
 * phasevocoderprocess.cc:925-999 */
void pvo_formant_shift(int n, float *mag, float env_comp);
double pvo_princarg(double a);

typedef struct pvo_resampler pvo_resampler;
pvo_resampler *pvo_res_create(void);
void pvo_res_destroy(pvo_resampler *r);
/* == RS_Speex::doresample (resampler.cc:772-817) for one mono channel; returns frames written */
int pvo_res_process(pvo_resampler *r, const float *in, int incount, float ratio, float *out);
void pvo_res_info(const pvo_resampler *r, unsigned *num, unsigned *den, int *filt_len, int *oversample, int *interp);
/* copies the sinc table (length returned) */
int pvo_res_table(const pvo_resampler *r, float *dst, int max);

#ifdef __cplusplus
}
#endif
void pvo_atan2f_array(const float *y, const float *x, float *out, long n);

#endif
