// Micro-benchmark / probe for the MFMA-reduced filter scan (fastscan.hip: fscan_mx_kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_reduce.hip -o /tmp/mfma_reduce && /tmp/mfma_reduce
// 1. layout probe: v_mfma_i32_16x16x64_i8 with A = 16 data bytes per lane and B = a one-hot byte per lane sums the four
//    lanes (g, i), g = 0..3, of a column and transposes: D[i][n] = sum_g dataA(lane 16 g + i)[byte n], found in lane
//    16 (i / 4) + n, register i % 4.
// 2. rate: the filter loop reduced to its skeleton -- per wave and group of 16 codes: one 16-byte load of eight formatted
//    lookups per lane, eight ds_read_b128 of table rows, eight chained MFMAs, four compares -- against the same loop
//    without the MFMAs, without the LDS reads, and with a bank-conflicting row order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

__global__ void probe(const uint8_t *data, int *out)
{
    const int l = threadIdx.x;
    v4i a, b = {0, 0, 0, 0}, c = {0, 0, 0, 0};
    const uint32_t *dw = reinterpret_cast<const uint32_t *>(data + l * 16);
    a[0] = dw[0]; a[1] = dw[1]; a[2] = dw[2]; a[3] = dw[3];
    const int n = l & 15;
    b[n >> 2] = 1 << (8 * (n & 3));
    v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}

// sparse form: v_smfmac_i32_16x16x128_i8 takes TWO rows per lane as its dense operand (32 bytes); the compressed operand
// keeps, for output row i, stored bytes 2 (i / 4) and 8 + 2 (i / 4) = 1 with 2-bit index i % 4 (tools/ubench/smfmac_probe.hip):
// D[i][n] = sum over the four lanes 16 g + n of (row0[i] + row1[i]), in lane 16 (i / 4) + n, register i % 4
__device__ __forceinline__ void sparse_pattern(int i, v4i &a, int &idx)
{
    a = v4i{0, 0, 0, 0};
    const int s1 = 2 * (i >> 2), s2 = 8 + s1;
    a[s1 >> 2] |= 1 << (8 * (s1 & 3));
    a[s2 >> 2] |= 1 << (8 * (s2 & 3));
    unsigned x = 0;
    for (int f = 0; f < 16; ++f) {
        const int v = (f == s1 || f == s2) ? (i & 3) : ((f ^ 1) == s1 || (f ^ 1) == s2) ? ((i + 2) & 3) : (f & 1);
        x |= (unsigned) v << (2 * f);
    }
    idx = (int) x;
}
__global__ void probe_sparse(const uint8_t *data, int *out)
{
    const int l = threadIdx.x;
    v4i a; int idx;
    sparse_pattern(l & 15, a, idx);
    const uint32_t *dw = reinterpret_cast<const uint32_t *>(data + l * 32);
    v8i b = {(int) dw[0], (int) dw[1], (int) dw[2], (int) dw[3], (int) dw[4], (int) dw[5], (int) dw[6], (int) dw[7]};
    v4i c = {0, 0, 0, 0};
    v4i d = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a, b, c, idx, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}

// MODE 0: every row to the MFMA, 1: pairs pre-added, 2: no LDS reads (MFMA rate alone), 3: fours pre-added
template <int MODE> __global__ __launch_bounds__(1024) void rate(const uint4 *__restrict__ fc, int64_t groups_per_block,
                                                                 const uint4 *__restrict__ table, int *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 8192; i += 1024) reinterpret_cast<uint4 *>(smem)[i] = table[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i onehot = {0, 0, 0, 0};
    onehot[(lane & 15) >> 2] = 1 << (8 * (lane & 3));
    const int thr = 40 + (lane & 15);
    v4i spa; int spidx;
    sparse_pattern(lane & 15, spa, spidx);
    int hits = 0;
    // a wave takes NG consecutive groups (NG KB of formatted lookups) per trip, the block 16 * NG of them
    constexpr int NG = 4;
    const uint4 *p = fc + ((size_t) (blockIdx.x & 3) * groups_per_block + wave * NG) * 64 + lane;   // 4 chunks, 64 blocks each
    uint4 w[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) w[j] = p[j * 64];
    auto issue = [&](v4i (&rows)[8], const uint4 &c) {
        const uint32_t ws[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            uint32_t ad;
            if (t & 1) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(ad) : "v"(ws[t >> 1]));
            else asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(ad) : "v"(ws[t >> 1]));
            if (MODE == 2 || MODE == 5 || MODE == 7 || MODE == 8 || MODE == 9) rows[t] = v4i{(int) ad, (int) ad, (int) ad, (int) ad};
            else rows[t] = *reinterpret_cast<const v4i *>(smem + ad);
        }
    };
    // MODE 0: every row through the matrix core.  MODE 1: rows added in PAIRS as packed bytes first (2 x 63 < 128 stays a
    // positive int8), one MFMA per pair.  MODE 3: in fours (only meaningful as a rate: 4 x 63 overflows int8).
    auto reduce = [&](const v4i (&rows)[8]) {
        v4i acc = {0, 0, 0, 0};
        if (MODE == 10) {                      // two chains of two, joined by four adds
            v4i acc2 = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const v8i b = {rows[2 * t][0], rows[2 * t][1], rows[2 * t][2], rows[2 * t][3],
                               rows[2 * t + 1][0], rows[2 * t + 1][1], rows[2 * t + 1][2], rows[2 * t + 1][3]};
                if (t & 1) acc2 = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc2, spidx, 0, 0);
                else acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
            }
            acc += acc2;
        } else if (MODE == 6 || MODE == 7 || MODE == 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const v8i b = {rows[2 * t][0], rows[2 * t][1], rows[2 * t][2], rows[2 * t][3],
                               rows[2 * t + 1][0], rows[2 * t + 1][1], rows[2 * t + 1][2], rows[2 * t + 1][3]};
                acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
            }
        } else if (MODE == 0 || MODE == 2 || MODE == 9) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(rows[t], onehot, acc, 0, 0, 0);
        } else if (MODE == 4 || MODE == 5) {          // no MFMA, one VALU op per row: the ceiling of the loads alone
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t & 3] ^= rows[t][t & 3];
        } else if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(rows[2 * t] + rows[2 * t + 1], onehot, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(rows[4 * t] + rows[4 * t + 1] + rows[4 * t + 2] + rows[4 * t + 3], onehot, acc, 0, 0, 0);
        }
        if (acc[0] < thr || acc[1] < thr || acc[2] < thr || acc[3] < thr) ++hits;
    };
    // two groups' rows in flight: the reads of group j + 1 are issued before the MFMAs of group j (16 DS operations per wave)
    v4i ra[8], rb[8];
    issue(ra, w[0]);
    if (MODE == 8 || MODE == 9) issue(rb, w[1]);
    for (int64_t g = wave * NG; g < groups_per_block; g += 16 * NG) {
        const uint4 *pn = p + 16 * NG * 64;
        uint4 cur[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) cur[j] = w[j];
        const bool more = g + 16 * NG < groups_per_block;
        if (more) {
#pragma unroll
            for (int j = 0; j < NG; ++j) w[j] = pn[j * 64];
        }
        p = pn;
        static_assert(NG == 4, "unrolled by hand");
        if (MODE == 8 || MODE == 9) {          // matrix pipe alone: constant operands, nothing else in the loop
            reduce(ra); reduce(rb); reduce(ra); reduce(rb);
        } else {
            issue(rb, cur[1]); reduce(ra);
            issue(ra, cur[2]); reduce(rb);
            issue(rb, cur[3]); reduce(ra);
            if (more) issue(ra, w[0]);
            reduce(rb);
        }
    }
    if (hits == 0x7fffffff) out[0] = hits;
    out[1 + blockIdx.x * 1024 + threadIdx.x] = hits;
}

// byte lookups: 8 bytes per lane and group (half the stream); address = per-lane constant (half, slot) with ks spliced in
// as byte 1 by one v_perm_b32 per row
__global__ __launch_bounds__(1024) void rate_bytes(const uint2 *__restrict__ fc, int64_t groups_per_block,
                                                   const uint4 *__restrict__ table, int *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 8192; i += 1024) reinterpret_cast<uint4 *>(smem)[i] = table[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, g4 = lane >> 4;
    const int thr = 40 + col;
    v4i spa; int spidx;
    sparse_pattern(col, spa, spidx);
    const int in_mid = (col >= 4 && col < 12) ? 1 : 0;
    const int e = ((g4 & 1) ^ in_mid) + 2 * (g4 >> 1);
    uint32_t C[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) C[t] = ((uint32_t) (t >> 2) << 16) | ((uint32_t) ((col + 4 * (t & 3) + e) & 15) << 4);
    int hits = 0;
    constexpr int NG = 4;
    const uint2 *p = fc + ((size_t) (blockIdx.x & 3) * groups_per_block + wave * NG) * 64 + lane;
    uint2 w[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) w[j] = p[j * 64];
    typedef const __attribute__((address_space(3))) v4i *lds_row;
    auto issue = [&](v4i (&rows)[8], const uint2 &c) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // result bytes: [C.b0, w.byte(t & 3), C.b2, C.b3]; v_perm_b32 selector bytes index {S0 = 4..7, S1 = 0..3}
            const uint32_t sel = 0x07060004u | ((uint32_t) (t & 3) << 8);     // byte1 <- S1 (w) byte t&3; others <- S0 (C) bytes 4,6,7
            const uint32_t ad = __builtin_amdgcn_perm(C[t], (t < 4) ? c.x : c.y, sel);
            rows[t] = *(lds_row) (uintptr_t) ad;
        }
    };
    auto reduce = [&](const v4i (&rows)[8]) {
        v4i acc = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const v8i b = {rows[2 * t][0], rows[2 * t][1], rows[2 * t][2], rows[2 * t][3],
                           rows[2 * t + 1][0], rows[2 * t + 1][1], rows[2 * t + 1][2], rows[2 * t + 1][3]};
            acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
        }
        if (acc[0] < thr || acc[1] < thr || acc[2] < thr || acc[3] < thr) ++hits;
    };
    v4i ra[8], rb[8];
    issue(ra, w[0]);
    for (int64_t g = wave * NG; g < groups_per_block; g += 16 * NG) {
        const uint2 *pn = (g + 16 * NG < groups_per_block) ? p + 16 * NG * 64 : p;
        w[0] = pn[0];
        issue(rb, w[1]); w[1] = pn[64]; reduce(ra);
        issue(ra, w[2]); w[2] = pn[128]; reduce(rb);
        issue(rb, w[3]); w[3] = pn[192]; reduce(ra);
        issue(ra, w[0]); reduce(rb);
        p = pn;
    }
    out[1 + blockIdx.x * 1024 + threadIdx.x] = hits;
}

int main()
{
    // ---- 1. layout probe ----
    std::vector<uint8_t> h(64 * 16);
    for (int l = 0; l < 64; ++l)
        for (int t = 0; t < 16; ++t) h[l * 16 + t] = (uint8_t) ((l * 7 + t * 13 + (l * t) % 5) % 64);
    uint8_t *d_data; int *d_out;
    hipMalloc(&d_data, h.size()); hipMalloc(&d_out, (1 + 256 * 1024) * sizeof(int));
    hipMemcpy(d_data, h.data(), h.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_data, d_out);
    std::vector<int> o(256);
    hipMemcpy(o.data(), d_out, 256 * sizeof(int), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n) {
            int want = 0;
            for (int g = 0; g < 4; ++g) want += h[(16 * g + i) * 16 + n];
            const int got = o[((i / 4) * 16 + n) * 4 + (i % 4)];
            if (want != got) { if (bad < 8) printf("  D[%d][%d] want %d got %d\n", i, n, want, got); ++bad; }
        }
    printf("layout probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);

    {
        std::vector<uint8_t> h2(64 * 32);
        for (int l = 0; l < 64; ++l)
            for (int t = 0; t < 32; ++t) h2[l * 32 + t] = (uint8_t) ((l * 5 + t * 11 + (l * t) % 7) % 64);
        uint8_t *d2;
        hipMalloc(&d2, h2.size());
        hipMemcpy(d2, h2.data(), h2.size(), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_sparse, dim3(1), dim3(64), 0, 0, d2, d_out);
        hipMemcpy(o.data(), d_out, 256 * sizeof(int), hipMemcpyDeviceToHost);
        int bad2 = 0;
        for (int i = 0; i < 16; ++i)
            for (int n = 0; n < 16; ++n) {
                int want = 0;
                for (int g = 0; g < 4; ++g) want += h2[(16 * g + n) * 32 + i] + h2[(16 * g + n) * 32 + 16 + i];
                const int got = o[((i / 4) * 16 + n) * 4 + (i % 4)];
                if (want != got) { if (bad2 < 8) printf("  sparse D[%d][%d] want %d got %d\n", i, n, want, got); ++bad2; }
            }
        printf("sparse layout probe: %s (%d mismatches)\n", bad2 ? "MISMATCH" : "ok", bad2);
    }

    // ---- 2. rate ----
    const int64_t gpb = 15616;                 // groups of 16 codes per block: ~250 K codes, as in the 1M x 1024 bench
    const size_t n_lane = (size_t) 4 * gpb * 64;
    std::vector<uint16_t> fc(n_lane * 8);
    auto e_of = [](int g, int n) {
        const bool inA = (n < 4 || n >= 12);
        const int e01 = (g & 1) == 0 ? (inA ? 0 : 1) : (inA ? 1 : 0);
        return e01 + 2 * (g >> 1);
    };
    uint4 *d_fc, *d_tab;
    hipMalloc(&d_fc, fc.size() * 2); hipMalloc(&d_tab, 128 * 1024);
    std::vector<uint8_t> tab(128 * 1024);
    for (auto &x : tab) x = (uint8_t) (rand() % 64);
    hipMemcpy(d_tab, tab.data(), tab.size(), hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int layout = 0; layout < 1; ++layout) {
        uint32_t x = 12345u;
        for (size_t i = 0; i < n_lane; ++i) {
            const int lane = (int) (i & 63), g = lane >> 4, n = lane & 15;
            for (int t = 0; t < 8; ++t) {
                x = x * 1664525u + 1013904223u;
                const int ks = (x >> 13) & 255, hh = t >> 2;
                int slot = (n + 4 * (t & 3) + e_of(g, n)) & 15;
                uint16_t v;
                if (layout == 0) v = (uint16_t) ((hh << 12) | (ks << 4) | slot);         // rotated: bank slot = subspace mod 16
                else v = (uint16_t) ((hh << 12) | (slot << 8) | ks);                       // plain [m][ks]: bank slot = ks mod 16
                fc[i * 8 + t] = v;
            }
        }
        hipMemcpy(d_fc, fc.data(), fc.size() * 2, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 11; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 4) hipLaunchKernelGGL(rate<4>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 5) hipLaunchKernelGGL(rate<5>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 6) hipLaunchKernelGGL(rate<6>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 7) hipLaunchKernelGGL(rate<7>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 10) hipLaunchKernelGGL(rate<10>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 9) hipLaunchKernelGGL(rate<9>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                if (mode == 8) hipLaunchKernelGGL(rate<8>, dim3(256), dim3(1024), 128 * 1024, 0, d_fc, gpb, d_tab, d_out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const double lookups = 256.0 * gpb * 64 * 8;
            printf("layout %s mode %d (%s): %.3f ms for 256 blocks x %lld codes  [%.1f TB/s of 16-byte rows; the bench's fscan_kernel: 0.49 ms]\n",
                   layout == 0 ? "rotated" : "plain  ", mode, mode == 0 ? "lds + 8 mfma" : mode == 1 ? "lds + pair adds + 4 mfma" : mode == 2 ? "8 mfma only" : mode == 3 ? "lds + quad adds + 2 mfma" : mode == 4 ? "lds only" : mode == 5 ? "lookup stream only" : mode == 6 ? "lds + 4 smfmac" : mode == 7 ? "4 smfmac + synthesized rows" : mode == 8 ? "4 smfmac, constant operands" : mode == 9 ? "8 mfma, constant operands" : "lds + 2 x 2 smfmac + adds",
                   best, (long long) gpb * 16, lookups * 16 / (best * 1e-3) / 1e12);
        }
    }
    {
        std::vector<uint8_t> fb(n_lane * 8);
        uint32_t x = 777u;
        for (auto &v : fb) { x = x * 1664525u + 1013904223u; v = (uint8_t) (x >> 13); }
        hipMemcpy(d_fc, fb.data(), fb.size(), hipMemcpyHostToDevice);
        hipFuncSetAttribute(reinterpret_cast<const void *>(rate_bytes), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(rate_bytes, dim3(256), dim3(1024), 128 * 1024, 0, reinterpret_cast<const uint2 *>(d_fc), gpb, d_tab, d_out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("byte lookups + v_perm addresses + lds + 4 smfmac: %.3f ms\n", best);
    }
    printf("hip status: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
