// Probe: what does a HIP graph buy for a launch-bound call of 3 small copies + 5 small kernels (the shape of one
// streaming call of the engine)?  direct launches vs hipGraphLaunch with per-call node parameter updates.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Args { float *p; int n; int k; float pad[40]; };
__global__ void kern(Args a) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < a.n) a.p[i] = a.p[i] * 1.0001f + (float)a.k; }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 4096;
    float *d, *d2, *h, *h2, *hd;
    CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&d2, 8192));
    CK(hipHostMalloc(&h, n * 4)); CK(hipHostMalloc(&h2, n * 4)); CK(hipHostMalloc(&hd, 8192));
    const int iters = 3000;
    std::vector<double> t_direct, t_graph;
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int it = 0; it < iters; ++it) {
        auto t0 = now();
        CK(hipMemcpyAsync(d, h, 1920 * 4, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(d2, hd, 5000, hipMemcpyHostToDevice, st));
        for (int k = 0; k < 5; ++k) { Args a{d, n, it + k}; hipLaunchKernelGGL(kern, dim3(6), dim3(256), 0, st, a); }
        CK(hipMemcpyAsync(h2, d, 2000 * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        t_direct.push_back(std::chrono::duration<double, std::micro>(now() - t0).count());
    }
    // explicit graph, same chain
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    hipGraphNode_t nodes[8];
    CK(hipGraphAddMemcpyNode1D(&nodes[0], g, nullptr, 0, d, h, 1920 * 4, hipMemcpyHostToDevice));
    CK(hipGraphAddMemcpyNode1D(&nodes[1], g, &nodes[0], 1, d2, hd, 5000, hipMemcpyHostToDevice));
    Args a[5]; void *argp[5][1]; hipKernelNodeParams kp[5];
    for (int k = 0; k < 5; ++k) {
        a[k] = Args{d, n, k}; argp[k][0] = &a[k];
        kp[k] = hipKernelNodeParams{}; kp[k].func = (void *)kern; kp[k].gridDim = dim3(6); kp[k].blockDim = dim3(256);
        kp[k].sharedMemBytes = 0; kp[k].kernelParams = argp[k]; kp[k].extra = nullptr;
        CK(hipGraphAddKernelNode(&nodes[2 + k], g, &nodes[1 + k], 1, &kp[k]));
    }
    CK(hipGraphAddMemcpyNode1D(&nodes[7], g, &nodes[6], 1, h2, d, 2000 * 4, hipMemcpyDeviceToHost));
    hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int it = 0; it < iters; ++it) {
        auto t0 = now();
        CK(hipGraphExecMemcpyNodeSetParams1D(ex, nodes[0], d + (it & 7), h, (1900 + (it & 15)) * 4, hipMemcpyHostToDevice));
        CK(hipGraphExecMemcpyNodeSetParams1D(ex, nodes[1], d2, hd, 4000 + (it & 255), hipMemcpyHostToDevice));
        for (int k = 0; k < 5; ++k) {
            a[k].k = it + k; kp[k].gridDim = dim3(4 + (it & 3));
            CK(hipGraphExecKernelNodeSetParams(ex, nodes[2 + k], &kp[k]));
        }
        CK(hipGraphExecMemcpyNodeSetParams1D(ex, nodes[7], h2, d, (1990 + (it & 7)) * 4, hipMemcpyDeviceToHost));
        CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        t_graph.push_back(std::chrono::duration<double, std::micro>(now() - t0).count());
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto p99 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() * 99 / 100]; };
    printf("direct: median %.1f us p99 %.1f   graph(update+launch): median %.1f us p99 %.1f\n", med(t_direct), p99(t_direct), med(t_graph), p99(t_graph));
    return 0;
}
