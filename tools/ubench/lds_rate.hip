// Micro-benchmark (round 6): how many conflict-free ds_read_b32 wave-instructions a CU sustains, alone and with 1 .. 3 VALU
// instructions per read (the rotated table gather of tools/ubench/gather_rot.hip levelled off at one read per ~6.9 cycles per CU with
// the LDS array 30 % and the VALU issue 50 % busy: which unit is the ceiling?).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_rate.hip -o tools/ubench/lds_rate && tools/ubench/lds_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: reads only (addresses fixed per lane, 8 reads in flight, results summed by independent adds afterwards: 1 VALU per read)
// MODE 1: + one v_perm_b32 per read forming the address (independent)
// MODE 2: + one dependent v_add_f32 chain (1 perm + 1 add per read)
// MODE 3: perm + dependent v_pk_fma_f32 chain
// MODE 4: perm + two alternating dependent v_pk_fma_f32 chains
// MODE 5: reads whose addresses are random per lane over the whole table, [m][ks] layout (the direct gather's conflicts), 1 add each
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float *out, int iters, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += NT) lds[i] = (float) (i & 1023);
    __syncthreads();
    uint32_t w = (tid * 2654435761u) ^ seed;                      // four "code bytes"
    const uint32_t laneoff = (uint32_t) ((lane & 31) * 4);
    float s0 = 0.f, s1 = 0.f;
    f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
    const f2 sel = {1.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        float t[8];
        uint32_t a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) a[u] = ((w >> (8 * (u & 3))) & 0xffu) * 256u + laneoff;                  // hoisted by the compiler: no VALU in the loop
            else if (MODE == 5) a[u] = (((w >> (8 * (u & 3))) & 0xffu) + 256u * (uint32_t) u) * 4u; // random bank per lane
            else a[u] = __builtin_amdgcn_perm(w, laneoff, 0x0c0c0000u | ((4u + (u & 3)) << 8));     // byte 1 = code byte, byte 0 = lane column
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const __attribute__((address_space(3))) float *>(a[u] + 4 * u * (MODE == 5 ? 0 : 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE <= 1 || MODE == 5) { if (u & 1) s1 += t[u]; else s0 += t[u]; }
            else if (MODE == 2) s0 += t[u];
            else if (MODE == 3) {
                f2 tt = {t[u], 0.f};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc0) : "v"(tt), "v"(sel));
            } else if (MODE == 4) {
                f2 tt = {t[u], 0.f};
                if (u & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc1) : "v"(tt), "v"(sel));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc0) : "v"(tt), "v"(sel));
            }
        }
        if (MODE != 0) w = w * 1664525u + 1013904223u;            // (2 VALU per 8 reads)
    }
    out[blockIdx.x * NT + tid] = s0 + s1 + acc0.x + acc0.y + acc1.x + acc1.y;
}

template <int MODE, int NT> static int run(const char *name, int blocks_per_cu)
{
    float *d; CK(hipMalloc(&d, 256 * 8 * 1024 * 4));
    const size_t smem = 65536;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(NT), smem, 0, d, 10, 1u);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(NT), smem, 0, d, iters, 1u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double reads_per_cu = (double) blocks_per_cu * (NT / 64) * iters * 8;
    printf("{\"mode\": \"%s\", \"threads\": %d, \"blocks_per_cu\": %d, \"waves_per_cu\": %d, \"ms\": %.3f, \"ns_per_read_per_cu\": %.3f, \"cycles_at_2.4GHz\": %.2f}\n",
           name, NT, blocks_per_cu, blocks_per_cu * NT / 64, ms, ms * 1e6 / reads_per_cu, ms * 1e6 / reads_per_cu * 2.4);
    CK(hipFree(d));
    return 0;
}

int main()
{
    run<0, 256>("reads + 1 independent add", 2);
    run<0, 512>("reads + 1 independent add", 2);
    run<1, 256>("perm + read + independent add", 2);
    run<1, 512>("perm + read + independent add", 2);
    run<2, 256>("perm + read + dependent add", 2);
    run<2, 512>("perm + read + dependent add", 2);
    run<3, 256>("perm + read + dependent pk_fma", 2);
    run<3, 512>("perm + read + dependent pk_fma", 2);
    run<4, 256>("perm + read + two pk_fma chains", 2);
    run<4, 512>("perm + read + two pk_fma chains", 2);
    run<5, 256>("random banks (direct gather) + independent add", 2);
    run<5, 512>("random banks (direct gather) + independent add", 2);
    return 0;
}
