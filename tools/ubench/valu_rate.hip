// Micro-benchmark: issue rate of a few VALU ops on gfx950 (cycles per wave64 instruction per SIMD, 4 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP> __global__ __launch_bounds__(1024) void k(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b0 = 11u, b1 = 13u, b2 = 17u, b3 = 19u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) {       // v_add_u32 x4
                asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (OP == 1) { // v_add3_u32 x4
                asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %6\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %7, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (OP == 2) { // v_lshl_add_u64 x2 (covers 4 dwords)
                unsigned long long x = ((unsigned long long) a1 << 32) | a0, y = ((unsigned long long) a3 << 32) | a2;
                unsigned long long p = ((unsigned long long) b1 << 32) | b0, q = ((unsigned long long) b3 << 32) | b2;
                asm volatile("v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %3" : "+v"(x), "+v"(y) : "v"(p), "v"(q));
                a0 = (uint32_t) x; a1 = (uint32_t) (x >> 32); a2 = (uint32_t) y; a3 = (uint32_t) (y >> 32);
            } else if (OP == 3) { // v_perm_b32 x4
                asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %5, %6\n v_perm_b32 %2, %2, %6, %7\n v_perm_b32 %3, %3, %7, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (OP == 4) { // v_pk_add_u16 x4
                asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %6\n v_pk_add_u16 %3, %3, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (OP == 5) { // v_lshlrev_b32_sdwa x4
                asm volatile("v_lshlrev_b32_sdwa %0, 4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                             "v_lshlrev_b32_sdwa %1, 4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_lshlrev_b32_sdwa %2, 4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                             "v_lshlrev_b32_sdwa %3, 4, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else if (OP == 6) { // v_pk_add_f32 x2 (two dwords each)
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 x = {__uint_as_float(a0), __uint_as_float(a1)}, y = {__uint_as_float(a2), __uint_as_float(a3)};
                f2 p = {__uint_as_float(b0), __uint_as_float(b1)}, q = {__uint_as_float(b2), __uint_as_float(b3)};
                asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3" : "+v"(x), "+v"(y) : "v"(p), "v"(q));
                a0 = __float_as_uint(x.x); a1 = __float_as_uint(x.y); a2 = __float_as_uint(y.x); a3 = __float_as_uint(y.y);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

template <int OP> static void run(const char *name, int n_instr_per_u)
{
    uint32_t *d; hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: 4 waves x iters x 16 x n_instr instructions
    const double instr = 4.0 * iters * 16 * n_instr_per_u;
    printf("%-22s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
    hipFree(d);
}

int main()
{
    run<0>("v_add_u32", 4); run<1>("v_add3_u32", 4); run<2>("v_lshl_add_u64", 2); run<3>("v_perm_b32", 4);
    run<4>("v_pk_add_u16", 4); run<5>("v_lshlrev_b32_sdwa", 4); run<6>("v_pk_add_f32", 2);
    return 0;
}
