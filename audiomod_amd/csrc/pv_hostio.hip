// pv_hostio.hip -- the batch engine with the host staging included (include/audiomod_pv.h, pv_hostio_*).
//
// The reference's callers keep their audio in host memory: planar float buffers per channel, filled from and
// written to 16-bit WAV data (main/main.cc:152-162,484-491; main/wavfile.cc:733-755,1334-1342).  Here `nstreams`
// such streams are processed in groups; a group's host-to-device copy, its kernels (pv_batch_run) and its
// device-to-host copy run on three HIP streams, so that while group g computes, group g+1 arrives and group g-1
// leaves.  With int16 on the wire the conversions of the reference's WAV reader and writer run on the device.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "audiomod_pv.h"
#include "pv_atan2f.h"

namespace {

// main/wavfile.cc:733-755: float = (float)(int16 * (1.0 / 32768.0)) -- exact, so one float multiply gives the same
__global__ void pv_i16_to_f32(const int16_t *__restrict__ in, float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i] * (1.0f / 32768.0f);
}
// main/wavfile.cc:1295-1306,1334-1342: (short)saturate(x * 32768.0f, -32768.0f, 32767.0f), truncation toward zero
__global__ void pv_f32_to_i16(const float *__restrict__ in, int16_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = in[i] * 32768.0f;
    if (v > 32767.0f) v = 32767.0f;
    else if (v < -32768.0f) v = -32768.0f;
    out[i] = (int16_t)(int)v;
}

// the analysis kernels' atan2f (pv_atan2f.h, device build: short division) on arrays, for pv_debug_atan2f
__global__ void pv_atan2f_probe(const float *__restrict__ y, const float *__restrict__ x, float *__restrict__ out,
                                int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pv_atan2f_fd_finite(y[i], x[i]);
}

// ... and the wave-per-frame kernels' polar conversion (table-driven atan2f, range-tested short division and square
// root, IEEE operations otherwise: pv_kernels.hip analyze_wave_role), for pv_debug_polar
__device__ const PvAtanBlob pv_atan_blob_probe = pv_atan_make_blob();
__global__ void pv_polar_probe(const float *__restrict__ im, const float *__restrict__ re, float *__restrict__ ph,
                               float *__restrict__ mag, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint32_t tab[PV_ATAN_BLOB_WORDS];
    if (threadIdx.x < PV_ATAN_BLOB_WORDS) tab[threadIdx.x] = pv_atan_blob_probe.w[threadIdx.x];
    __syncthreads();
    typedef __attribute__((address_space(3))) const unsigned char lds_u8;
    const unsigned char *atab = reinterpret_cast<const unsigned char *>((uintptr_t)(lds_u8 *)tab); // LDS address as a number
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float y = im[i], x = re[i];
    const float a = x * x + y * y;
    const float mx = fmaxf(fabsf(x), fabsf(y)), mn = fminf(fabsf(x), fabsf(y));
    if (mx < 0x1p63f && mn >= 0x1p-48f) {
        mag[i] = pv_sqrt_safe(a);
        ph[i] = pv_atan2f_fd_tab<true>(y, x, atab);
    } else {
        mag[i] = sqrtf(a);
        ph[i] = pv_atan2f_fd_tab<false>(y, x, atab);
    }
}

// every float with bit pattern in [first, first + count): pv_sqrt_safe against the compiler's correctly rounded sqrtf
__global__ void pv_sqrt_sweep_kernel(uint32_t first, uint64_t count, unsigned long long *res) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    uint32_t first_bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t u = first + (uint32_t)i;
        const float a = pv_u2f(u);
        if (pv_f2u(pv_sqrt_safe(a)) != pv_f2u(sqrtf(a))) {
            if (!bad) first_bad = u;
            ++bad;
        }
    }
    if (bad) {
        atomicAdd(&res[0], bad);
        atomicMin(&res[1], (unsigned long long)first_bad);
    }
}

constexpr int kSlots = 3; // groups in flight

} // namespace

struct pv_hostio {
    pv_batch *batch = nullptr;
    int device = 0, channels = 0, wire = 0;
    int32_t nstreams = 0, per_group = 0;
    int64_t frames = 0, out_frames = 0;
    float *d_in[kSlots] = {}, *d_out[kSlots] = {};
    int16_t *d_in16[kSlots] = {}, *d_out16[kSlots] = {};
    hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
    hipEvent_t ev_up[kSlots] = {}, ev_run[kSlots] = {}, ev_down[kSlots] = {};
};

extern "C" {

int pv_debug_atan2f(const float *y, const float *x, float *out, int64_t n, int device) {
    if (n < 0 || (n > 0 && (!y || !x || !out))) return PV_ERR_INVALID_ARG;
    if (n == 0) return PV_OK;
    float *d = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&d, (size_t)n * 3 * sizeof(float)) != hipSuccess)
        return PV_ERR_HIP;
    bool ok = hipMemcpy(d, y, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d + n, x, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(pv_atan2f_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d, d + n, d + 2 * n, n);
        ok = hipMemcpy(out, d + 2 * n, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    return ok ? PV_OK : PV_ERR_HIP;
}

int pv_debug_polar(const float *im, const float *re, float *phase, float *mag, int64_t n, int device) {
    if (n < 0 || (n > 0 && (!im || !re || !phase || !mag))) return PV_ERR_INVALID_ARG;
    if (n == 0) return PV_OK;
    float *d = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&d, (size_t)n * 4 * sizeof(float)) != hipSuccess)
        return PV_ERR_HIP;
    bool ok = hipMemcpy(d, im, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d + n, re, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(pv_polar_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d, d + n, d + 2 * n,
                           d + 3 * n, n);
        ok = hipMemcpy(phase, d + 2 * n, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(mag, d + 3 * n, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    return ok ? PV_OK : PV_ERR_HIP;
}

int pv_debug_sqrt_sweep(uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint32_t *first_bad, int device) {
    if (!mismatches || !first_bad) return PV_ERR_INVALID_ARG;
    unsigned long long *d = nullptr, h[2] = {0ull, ~0ull};
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&d, sizeof(h)) != hipSuccess) return PV_ERR_HIP;
    bool ok = hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && count) {
        hipLaunchKernelGGL(pv_sqrt_sweep_kernel, dim3(4096), dim3(256), 0, nullptr, first_bits, (uint64_t)count, d);
        ok = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    *mismatches = h[0];
    *first_bad = (uint32_t)h[1];
    return ok ? PV_OK : PV_ERR_HIP;
}

void *pv_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void pv_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

void pv_hostio_destroy(pv_hostio *h) {
    if (!h) return;
    for (int i = 0; i < kSlots; ++i) {
        if (h->d_in[i]) (void)hipFree(h->d_in[i]);
        if (h->d_out[i]) (void)hipFree(h->d_out[i]);
        if (h->d_in16[i]) (void)hipFree(h->d_in16[i]);
        if (h->d_out16[i]) (void)hipFree(h->d_out16[i]);
        if (h->ev_up[i]) (void)hipEventDestroy(h->ev_up[i]);
        if (h->ev_run[i]) (void)hipEventDestroy(h->ev_run[i]);
        if (h->ev_down[i]) (void)hipEventDestroy(h->ev_down[i]);
    }
    if (h->s_up) (void)hipStreamDestroy(h->s_up);
    if (h->s_run) (void)hipStreamDestroy(h->s_run);
    if (h->s_down) (void)hipStreamDestroy(h->s_down);
    pv_batch_destroy(h->batch);
    delete h;
}

int64_t pv_hostio_out_frames(const pv_hostio *h) { return h ? h->out_frames : -1; }

#define HIO(call)                         \
    do {                                  \
        if ((call) != hipSuccess) {       \
            pv_hostio_destroy(h);         \
            return PV_ERR_HIP;            \
        }                                 \
    } while (0)

int pv_hostio_create(const pv_config *cfg, int32_t nstreams, int64_t frames, int32_t block, int32_t flush, int device,
                     int32_t streams_per_group, int32_t wire, pv_hostio **out) {
    if (!cfg || !out || nstreams < 1 || frames < 1 || streams_per_group < 1 || (wire != PV_WIRE_F32 && wire != PV_WIRE_I16))
        return PV_ERR_INVALID_ARG;
    *out = nullptr;
    pv_hostio *h = new pv_hostio();
    h->device = device;
    h->channels = cfg->channels;
    h->wire = wire;
    h->nstreams = nstreams;
    h->per_group = streams_per_group < nstreams ? streams_per_group : nstreams;
    h->frames = frames;
    // one batch engine of a group's size serves every group in turn (the last group may be smaller: it runs as a
    // full group whose surplus streams read and write the slot's own buffer and are not copied back)
    int st = pv_batch_create(cfg, h->per_group, frames, block, flush, device, &h->batch);
    if (st != PV_OK) {
        delete h;
        return st;
    }
    h->out_frames = pv_batch_out_frames(h->batch);
    const size_t n_in = (size_t)h->per_group * h->channels * (size_t)frames;
    const size_t n_out = (size_t)h->per_group * h->channels * (size_t)(h->out_frames > 0 ? h->out_frames : 1);
    HIO(hipSetDevice(device));
    for (int i = 0; i < kSlots; ++i) {
        HIO(hipMalloc((void **)&h->d_in[i], n_in * sizeof(float)));
        HIO(hipMemset(h->d_in[i], 0, n_in * sizeof(float)));
        HIO(hipMalloc((void **)&h->d_out[i], n_out * sizeof(float)));
        if (wire == PV_WIRE_I16) {
            HIO(hipMalloc((void **)&h->d_in16[i], n_in * sizeof(int16_t)));
            HIO(hipMalloc((void **)&h->d_out16[i], n_out * sizeof(int16_t)));
        }
        HIO(hipEventCreateWithFlags(&h->ev_up[i], hipEventDisableTiming));
        HIO(hipEventCreateWithFlags(&h->ev_run[i], hipEventDisableTiming));
        HIO(hipEventCreateWithFlags(&h->ev_down[i], hipEventDisableTiming));
    }
    HIO(hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking));
    HIO(hipStreamCreateWithFlags(&h->s_run, hipStreamNonBlocking));
    HIO(hipStreamCreateWithFlags(&h->s_down, hipStreamNonBlocking));
    *out = h;
    return PV_OK;
}
#undef HIO

int pv_hostio_run(pv_hostio *h, const void *host_in, void *host_out) {
    if (!h || !host_in || (!host_out && h->out_frames > 0)) return PV_ERR_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return PV_ERR_HIP;
    const size_t esz = h->wire == PV_WIRE_I16 ? sizeof(int16_t) : sizeof(float);
    const size_t row_in = (size_t)h->channels * (size_t)h->frames, row_out = (size_t)h->channels * (size_t)h->out_frames;
    const int groups = (h->nstreams + h->per_group - 1) / h->per_group;
    for (int g = 0; g < groups; ++g) {
        const int slot = g % kSlots;
        const int s0 = g * h->per_group;
        const int ns = h->nstreams - s0 < h->per_group ? h->nstreams - s0 : h->per_group;
        const size_t n_in = (size_t)ns * row_in, n_out = (size_t)ns * row_out;
        // up: the slot's input buffer is free once the run that read it (three groups ago) has finished
        if (g >= kSlots && hipStreamWaitEvent(h->s_up, h->ev_run[slot], 0) != hipSuccess) return PV_ERR_HIP;
        void *d_up = h->wire == PV_WIRE_I16 ? (void *)h->d_in16[slot] : (void *)h->d_in[slot];
        if (hipMemcpyAsync(d_up, (const char *)host_in + (size_t)s0 * row_in * esz, n_in * esz, hipMemcpyHostToDevice,
                           h->s_up) != hipSuccess)
            return PV_ERR_HIP;
        if (hipEventRecord(h->ev_up[slot], h->s_up) != hipSuccess) return PV_ERR_HIP;
        // run: after the input has arrived and the slot's output buffer has left (three groups ago)
        if (hipStreamWaitEvent(h->s_run, h->ev_up[slot], 0) != hipSuccess) return PV_ERR_HIP;
        if (g >= kSlots && hipStreamWaitEvent(h->s_run, h->ev_down[slot], 0) != hipSuccess) return PV_ERR_HIP;
        if (h->wire == PV_WIRE_I16)
            hipLaunchKernelGGL(pv_i16_to_f32, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, h->s_run,
                               h->d_in16[slot], h->d_in[slot], (int64_t)n_in);
        const int st = pv_batch_run(h->batch, h->d_in[slot], h->d_out[slot], (void *)h->s_run);
        if (st != PV_OK) return st;
        if (h->wire == PV_WIRE_I16 && n_out > 0)
            hipLaunchKernelGGL(pv_f32_to_i16, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, h->s_run,
                               h->d_out[slot], h->d_out16[slot], (int64_t)n_out);
        if (hipEventRecord(h->ev_run[slot], h->s_run) != hipSuccess) return PV_ERR_HIP;
        // down
        if (hipStreamWaitEvent(h->s_down, h->ev_run[slot], 0) != hipSuccess) return PV_ERR_HIP;
        const void *d_dn = h->wire == PV_WIRE_I16 ? (const void *)h->d_out16[slot] : (const void *)h->d_out[slot];
        if (n_out > 0 && hipMemcpyAsync((char *)host_out + (size_t)s0 * row_out * esz, d_dn, n_out * esz,
                                        hipMemcpyDeviceToHost, h->s_down) != hipSuccess)
            return PV_ERR_HIP;
        if (hipEventRecord(h->ev_down[slot], h->s_down) != hipSuccess) return PV_ERR_HIP;
    }
    if (hipStreamSynchronize(h->s_down) != hipSuccess || hipStreamSynchronize(h->s_run) != hipSuccess ||
        hipStreamSynchronize(h->s_up) != hipSuccess)
        return PV_ERR_HIP;
    return hipGetLastError() == hipSuccess ? PV_OK : PV_ERR_HIP;
}

} // extern "C"
