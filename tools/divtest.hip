#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float div_fast(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    float e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    return q;
}
__device__ __forceinline__ float div_b(float a, float b) { // refined reciprocal, one correction
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    float e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ float div_c(float a, float b) { // raw reciprocal, two corrections
    float r = __builtin_amdgcn_rcpf(b);
    float q = a * r;
    float e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
}
__device__ uint32_t rng(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
__global__ void k(unsigned long long *bad, unsigned long long *firstbad, int iters, int mode, int variant) {
    uint64_t s = (blockIdx.x * 1024ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    unsigned long long nb = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t ua = rng(s), ub = rng(s);
        // exponents limited to [64, 190]: normal operands, quotient far from overflow / underflow
        uint32_t ea = 64 + (ua >> 23 & 0xff) % 127, eb = 64 + (ub >> 23 & 0xff) % 127;
        if (mode == 1) { ea = 120 + (ua >> 23 & 7); eb = 120 + (ub >> 23 & 7); }
        ua = (ua & 0x807fffffu) | (ea << 23); ub = (ub & 0x807fffffu) | (eb << 23);
        float a = __uint_as_float(ua), b = __uint_as_float(ub);
        float q0 = a / b, q1 = variant == 0 ? div_fast(a, b) : variant == 1 ? div_b(a, b) : div_c(a, b);
        if (__float_as_uint(q0) != __float_as_uint(q1)) { if (!nb) *firstbad = ((unsigned long long)ua << 32) | ub; ++nb; }
    }
    if (nb) atomicAdd(bad, nb);
}
int main() {
    unsigned long long *d, h[2] = {0, 0};
    hipMalloc(&d, 16); 
    for (int variant = 1; variant < 3; ++variant)
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, 16);
        printf("variant %d ", variant);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, d + 1, 4000, mode, variant);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatches of %llu (first %016llx)\n", mode, h[0], 4096ull * 256 * 4000, h[1]);
    }
    return 0;
}
