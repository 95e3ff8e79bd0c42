// Accuracy of gfx950's v_sin_f32 / v_cos_f32 (argument in turns) as the fast synthesis variant uses them
// (pv_kernels.hip synth_wave_role<..., kFast>): sin / cos of p for p in [-2 pi, 2 pi] (and a wider range with the
// two-term reduction the frequency-compression modes use), against double precision on the host.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/sincos_hw_probe.hip -o /tmp/sincos_probe && /tmp/sincos_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *p, float *s, float *c, int n, int reduce) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = p[i];
    if (reduce) {
        const float q = __builtin_rintf(x * 0.15915494309189535f);
        x = __builtin_fmaf(q, -6.2831854820251465f, x);
        x = __builtin_fmaf(q, 1.7484555e-07f, x);
    }
    const float rev = x * 0.15915494309189535f;
    s[i] = __builtin_amdgcn_sinf(rev);
    c[i] = __builtin_amdgcn_cosf(rev);
}
int main() {
    const int n = 1 << 22;
    for (int reduce = 0; reduce < 2; ++reduce) {
        const double range = reduce ? 700.0 : 2.0 * M_PI;
        std::vector<float> p(n), s(n), c(n);
        for (int i = 0; i < n; ++i) p[i] = (float)(-range + 2.0 * range * (i + 0.37) / n);
        float *dp, *ds, *dc;
        hipMalloc(&dp, n * 4), hipMalloc(&ds, n * 4), hipMalloc(&dc, n * 4);
        hipMemcpy(dp, p.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dp, ds, dc, n, reduce);
        hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        double es = 0, ec = 0, rs = 0;
        for (int i = 0; i < n; ++i) {
            const double a = std::fabs((double)s[i] - std::sin((double)p[i])), b = std::fabs((double)c[i] - std::cos((double)p[i]));
            es = a > es ? a : es, ec = b > ec ? b : ec, rs += a * a + b * b;
        }
        printf("range +-%.1f rad%s: max |sin err| %.3e  max |cos err| %.3e  rms %.3e\n", range,
               reduce ? " (two-term reduction)" : "", es, ec, std::sqrt(rs / (2.0 * n)));
    }
    return 0;
}
