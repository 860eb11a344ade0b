// Probe (round 4): what does a one-query exact scan of a 2 GB shard cost in parts?  125 M rows of 16 bytes streamed by `blocks` x 1024
// threads, U rows per thread and trip; WORK = 0: xor of the words only, 1: the 16 table lookups + sequential adds + running minimum.
// build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int U, int WORK>
__global__ __launch_bounds__(1024) void probe(const uint4 *__restrict__ rows, int64_t n, int64_t chunk, const float *__restrict__ gtab, float *out)
{
    __shared__ float lds[16 * 256];
    for (int i = threadIdx.x; i < 16 * 256; i += 1024) lds[i] = gtab[i];
    __syncthreads();
    const int64_t b = (int64_t) blockIdx.x * chunk;
    int64_t e = b + chunk; if (e > n) e = n;
    float best = 3.4e38f; uint32_t x = 0;
    for (int64_t n0 = b + threadIdx.x; n0 < e; n0 += 1024 * U) {
        uint4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = n0 + u * 1024; w[u] = rows[i < e ? i : b]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (WORK == 0) { x ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w; }
            else {
                const uint32_t ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
                float v[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m] = lds[m * 256 + ((ww[m >> 2] >> (8 * (m & 3))) & 255u)];
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < 16; ++m) acc += v[m];
                if (acc < best) { best = acc; x = (uint32_t) (n0 + u * 1024); }
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = best + (float) x;
}
__global__ void fill(uint32_t *p, int64_t nwords)
{
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t) gridDim.x * blockDim.x) {
        uint32_t s = (uint32_t) i * 2654435761u + 12345u;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        p[i] = s;
    }
}
template <int U, int WORK> void run(const uint4 *rows, int64_t n, int blocks, const float *tab, float *out)
{
    const int64_t chunk = (n + blocks - 1) / blocks;
    hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void) hipEventRecord(a);
        hipLaunchKernelGGL((probe<U, WORK>), dim3(blocks), dim3(1024), 0, 0, rows, n, chunk, tab, out);
        (void) hipEventRecord(b); (void) hipEventSynchronize(b);
        float ms; (void) hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("U=%d work=%d blocks=%4d: %.3f ms  %.2f TB/s\n", U, WORK, blocks, best, n * 16.0 / best / 1e9);
}
int main()
{
    const int64_t n = 125000000;
    uint4 *rows; float *tab, *out;
    (void) hipMalloc(&rows, n * 16); hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *) rows, n * 4); (void) hipDeviceSynchronize();
    std::vector<float> h(16 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float) (i % 97);
    (void) hipMalloc(&tab, h.size() * 4); (void) hipMalloc(&out, 4096 * 1024 * 4);
    (void) hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<4, 0>(rows, n, 512, tab, out);
    run<4, 0>(rows, n, 1024, tab, out);
    run<4, 0>(rows, n, 2048, tab, out);
    run<8, 0>(rows, n, 512, tab, out);
    run<1, 0>(rows, n, 512, tab, out);
    run<4, 1>(rows, n, 512, tab, out);
    run<4, 1>(rows, n, 1024, tab, out);
    run<4, 1>(rows, n, 2048, tab, out);
    run<8, 1>(rows, n, 512, tab, out);
    run<2, 1>(rows, n, 512, tab, out);
    run<1, 1>(rows, n, 512, tab, out);
    return 0;
}
