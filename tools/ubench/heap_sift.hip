// Micro-benchmark (round 5): one sift of std::partial_sort's heap, k = 100 (a tied top-100 query replays ~1100 of
// them, 570 ns each with the heap in LDS).  A: wh_adjust_top (rii_device.h: heap in LDS, ballots for the directions, one LDS round
// trip for the path).  B: the heap in REGISTERS, two entries per lane (j and j + 64), libstdc++'s __adjust_heap / __push_heap as
// wave-uniform scalar code over v_readlane + compare-and-select writes.  Both replay the same pseudo-random sequence and must end in
// the same heap.   hipcc --offload-arch=gfx950 -O3 -I rii_amd/csrc -I include tools/ubench/heap_sift.hip -o tools/ubench/heap_sift
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "rii_device.h"
using namespace riiamd;

__device__ __forceinline__ pq64_t whr_get(pq64_t hv, int j)          // entry of lane j (wave-uniform index)
{
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (hv & 0xffffffffu), j);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (hv >> 32), j);
    return ((pq64_t) hi << 32) | lo;
}

__device__ __forceinline__ pq64_t r2_get(pq64_t h0, pq64_t h1, int j) { return j < 64 ? whr_get(h0, j) : whr_get(h1, j - 64); }
__device__ __forceinline__ void r2_set(pq64_t &h0, pq64_t &h1, int j, pq64_t v, int lane)
{
    if (j < 64) { if (lane == j) h0 = v; } else { if (lane == j - 64) h1 = v; }
}
__device__ __forceinline__ void r2_adjust(pq64_t &h0, pq64_t &h1, int hole, int len, pq64_t v, int lane)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        pq64_t a = r2_get(h0, h1, child);
        const pq64_t b = r2_get(h0, h1, child - 1);
        if (pq64_less(a, b)) { child--; a = b; }
        r2_set(h0, h1, hole, a, lane);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        r2_set(h0, h1, hole, r2_get(h0, h1, child - 1), lane);
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top) {
        const pq64_t pv = r2_get(h0, h1, parent);
        if (!pq64_less(pv, v)) break;
        r2_set(h0, h1, hole, pv, lane);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    r2_set(h0, h1, hole, v, lane);
}

__global__ __launch_bounds__(64) void sift_kernel(int mode, int k, int nsift, const uint32_t *__restrict__ vals, unsigned long long *out, long long *cycles)
{
    __shared__ pq64_t heap[128];
    const int lane = threadIdx.x;
    // initial heap: a max-heap by construction (descending array)
    for (int i = lane; i < k; i += 64) heap[i] = ((pq64_t) (0xf0000000u - (uint32_t) i * 1000u) << 32) | (uint32_t) i;
    __syncthreads();
    pq64_t h0 = lane < k ? heap[lane] : 0ull, h1 = lane + 64 < k ? heap[lane + 64] : 0ull;
    const long long t0 = clock64();
    if (mode == 0) {
        pq64_t topv = wh_uniform(heap[0]);
        for (int s = 0; s < nsift; ++s) {
            const pq64_t v = ((pq64_t) vals[s] << 32) | (uint32_t) (1000 + s);
            if (pq64_less(v, topv)) topv = wh_adjust_top(heap, k, v, lane);
        }
    } else {
        pq64_t topv = whr_get(h0, 0);
        for (int s = 0; s < nsift; ++s) {
            const pq64_t v = ((pq64_t) vals[s] << 32) | (uint32_t) (1000 + s);
            if (pq64_less(v, topv)) { r2_adjust(h0, h1, 0, k, v, lane); topv = whr_get(h0, 0); }
        }
        if (lane < k) heap[lane] = h0;
        if (lane + 64 < k) heap[lane + 64] = h1;
    }
    const long long t1 = clock64();
    __syncthreads();
    for (int i = lane; i < k; i += 64) out[i] = heap[i];
    if (lane == 0) *cycles = t1 - t0;
}

int main()
{
    const int k = 100, nsift = 1100;
    std::vector<uint32_t> v(nsift);
    uint32_t s = 12345u, cur = 0xefff0000u;
    for (int i = 0; i < nsift; ++i) { s = s * 1664525u + 1013904223u; cur -= (s >> 20); v[i] = cur - (s & 0xffffu) * 50u; }   // mostly below the top: every one sifts
    uint32_t *dv; unsigned long long *dout; long long *dc;
    (void) hipMalloc(&dv, nsift * 4); (void) hipMalloc(&dout, 2 * 128 * 8); (void) hipMalloc(&dc, 16);
    (void) hipMemcpy(dv, v.data(), nsift * 4, hipMemcpyHostToDevice);
    unsigned long long h[2][128]; long long cyc[2];
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(sift_kernel, dim3(1), dim3(64), 0, 0, mode, k, nsift, dv, dout + mode * 128, dc + mode);
        (void) hipDeviceSynchronize();
        (void) hipMemcpy(h[mode], dout + mode * 128, k * 8, hipMemcpyDeviceToHost);
        (void) hipMemcpy(&cyc[mode], dc + mode, 8, hipMemcpyDeviceToHost);
    }
    bool same = true;
    for (int i = 0; i < k; ++i) same = same && h[0][i] == h[1][i];
    printf("{\"k\": %d, \"sifts\": %d, \"lds_heap_cycles_per_sift\": %.1f, \"register_heap_cycles_per_sift\": %.1f, \"same_heap\": %s}\n", k, nsift,
           (double) cyc[0] / nsift, (double) cyc[1] / nsift, same ? "true" : "false");
    return same ? 0 : 1;
}
