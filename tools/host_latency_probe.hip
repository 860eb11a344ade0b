// host_latency_probe.hip -- what one synchronous host call costs around its kernels on this box (round 3, latency of the README pattern).
//   a) pinned H2D copy + K empty kernels + pinned D2H copy + hipStreamSynchronize           (what rii_query_linear does today)
//   b) K empty kernels + hipStreamSynchronize
//   c) K empty kernels, the last one reads its input from and writes its result + a sequence flag to coherent host memory;
//      the host spins on the flag                                                             (zero-copy + flag)
// build: hipcc --offload-arch=gfx950 -O2 tools/host_latency_probe.hip -o /tmp/hlp ; run: /tmp/hlp [K]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty(float *p) { if (p && threadIdx.x == 12345) p[0] = 1.f; }
__global__ void k_publish(const float *in, float *out, volatile unsigned int *flag, unsigned int seq)
{
    float v = in[threadIdx.x];
    out[threadIdx.x] = v + 1.f;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store(const_cast<unsigned int *>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void report(const char *what, std::vector<double> &t)
{
    std::sort(t.begin(), t.end());
    printf("%-58s p10 %.1f  p50 %.1f  p90 %.1f us\n", what, t[t.size() / 10], t[t.size() / 2], t[t.size() * 9 / 10]);
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 2, iters = 2000;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *pin = nullptr, *dev = nullptr;
    CK(hipHostMalloc(&pin, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    CK(hipMalloc(&dev, 4096));
    memset(pin, 0, 4096);
    volatile unsigned int *flag = reinterpret_cast<volatile unsigned int *>(pin + 512);
    std::vector<double> t;
    for (int mode = 0; mode < 3; ++mode) {
        t.clear();
        for (int it = 0; it < iters + 100; ++it) {
            const double t0 = now_us();
            if (mode == 0) {
                CK(hipMemcpyAsync(dev, pin, 512, hipMemcpyHostToDevice, st));
                for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, dev);
                CK(hipMemcpyAsync(pin + 256, dev, 64, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
            } else if (mode == 1) {
                for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, dev);
                CK(hipStreamSynchronize(st));
            } else {
                for (int k = 0; k + 1 < K; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, dev);
                const unsigned int seq = (unsigned int) it + 1u;
                hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, pin, pin + 256, flag, seq);
                while (*flag != seq) { __builtin_ia32_pause(); }
            }
            if (it >= 100) t.push_back(now_us() - t0);
        }
        char what[128];
        snprintf(what, sizeof what, mode == 0 ? "a) H2D + %d kernels + D2H + hipStreamSynchronize" : mode == 1 ? "b) %d kernels + hipStreamSynchronize"
                 : "c) %d kernels, zero-copy in/out, host spins on a flag", K);
        report(what, t);
        CK(hipStreamSynchronize(st));
    }
    return 0;
}
