// Probe (round 4): how fast are divergent 4-byte gathers from a small L1-resident table through the vector memory path, next to
// the LDS gathers of the exact one-query scan?  NL lookups of each "code" go to LDS, NG to a 16 KiB global table.  No code stream: the
// indices come from a per-lane LCG, so only the two lookup engines are measured.   build: hipcc --offload-arch=gfx950 -O3 gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NL, int NG>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ gtab, float *out, int iters)
{
    __shared__ float lds[16 * 256];
    for (int i = threadIdx.x; i < 16 * 256; i += 256) lds[i] = gtab[i];
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint32_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; r[j] = s; }
        float v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const uint32_t code = (r[m >> 2] >> (8 * (m & 3))) & 255u;
            if (m < NL) v[m] = lds[m * 256 + code];
            else if (m < NL + NG) v[m] = gtab[m * 256 + code];
            else v[m] = 0.f;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) acc += v[m];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int NL, int NG> float run(const float *tab, float *out, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<NL, NG>), dim3(256 * 8), dim3(256), 0, 0, tab, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<NL, NG>), dim3(256 * 8), dim3(256), 0, 0, tab, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double items = 256.0 * 8 * 256 * iters;
    printf("LDS %2d + L1 %2d lookups per item: %.3f ms  %.2f G items/s  (as 16-byte codes: %.2f TB/s)\n", NL, NG, ms, items / ms / 1e6, items * 16 / ms / 1e9);
    return ms;
}
int main()
{
    float *tab, *out;
    std::vector<float> h(16 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float) (i % 97);
    hipMalloc(&tab, h.size() * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 4000;
    run<16, 0>(tab, out, iters);
    run<15, 1>(tab, out, iters);
    run<14, 2>(tab, out, iters);
    run<13, 3>(tab, out, iters);
    run<12, 4>(tab, out, iters);
    run<10, 6>(tab, out, iters);
    run<8, 8>(tab, out, iters);
    run<0, 16>(tab, out, iters);
    run<12, 0>(tab, out, iters);
    run<8, 0>(tab, out, iters);
    return 0;
}
