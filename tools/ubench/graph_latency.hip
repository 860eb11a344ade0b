// Micro-benchmark: does a captured hipGraph beat eager submission for the shape of a single-query call?
//   (pinned H2D of 512 B, three short dependent kernels, pinned D2H of 16 B, one synchronisation)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/graph_latency.hip -o /tmp/graph_latency && /tmp/graph_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k1(const float *q, float *t) { t[threadIdx.x] = q[threadIdx.x & 127] * 2.f; }
__global__ void k2(const float *t, float *u, int n)
{
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += t[i & 255];
    u[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k3(const float *u, float *out) { if (threadIdx.x < 4) out[threadIdx.x] = u[threadIdx.x]; }

int main()
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float *h_in, *h_out, *d_q, *d_t, *d_u, *d_o;
    hipHostMalloc(&h_in, 512); hipHostMalloc(&h_out, 16);
    hipMalloc(&d_q, 512); hipMalloc(&d_t, 1024); hipMalloc(&d_u, 256 * 256 * 4); hipMalloc(&d_o, 16);
    for (int i = 0; i < 128; ++i) h_in[i] = (float) i;
    auto enqueue = [&]() {
        hipMemcpyAsync(d_q, h_in, 512, hipMemcpyHostToDevice, st);
        hipLaunchKernelGGL(k1, dim3(1), dim3(256), 0, st, d_q, d_t);
        hipLaunchKernelGGL(k2, dim3(256), dim3(256), 0, st, d_t, d_u, 4096);
        hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, st, d_u, d_o);
        hipMemcpyAsync(h_out, d_o, 16, hipMemcpyDeviceToHost, st);
    };
    auto run = [&](const char *name, auto &&fn) {
        std::vector<double> us;
        for (int i = 0; i < 2200; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            fn();
            hipStreamSynchronize(st);
            const auto t1 = std::chrono::steady_clock::now();
            if (i >= 200) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(us.begin(), us.end());
        printf("%-28s p50 %.1f us   p99 %.1f us   min %.1f us\n", name, us[us.size() / 2], us[us.size() * 99 / 100], us[0]);
    };
    run("eager (5 submissions)", enqueue);
    hipGraph_t graph; hipGraphExec_t exec;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    enqueue();
    hipStreamEndCapture(st, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    run("graph replay", [&]() { hipGraphLaunch(exec, st); });
    // kernels only (device-resident call: no copies)
    auto kernels = [&]() {
        hipLaunchKernelGGL(k1, dim3(1), dim3(256), 0, st, d_q, d_t);
        hipLaunchKernelGGL(k2, dim3(256), dim3(256), 0, st, d_t, d_u, 4096);
        hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, st, d_u, d_o);
    };
    run("eager, 3 kernels only", kernels);
    hipGraph_t g2; hipGraphExec_t e2;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    kernels();
    hipStreamEndCapture(st, &g2);
    hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0);
    run("graph replay, 3 kernels", [&]() { hipGraphLaunch(e2, st); });
    printf("hip status: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
