// Micro-benchmark (round 6): the random 4-byte table gather of the one-query inverted-index kernels.
//
//   A  "direct"   what ivf_fused_kernel / ivf_shard_any_kernel do today: thread = candidate, table [m][ks] in LDS, every lookup a
//                 ds_read_b32 at lds[m * 256 + code[m]] -- the 32 lanes of a DS service group hit banks code % 32 at random
//                 (~3.5 distinct rows on the busiest bank: 65 % of the LDS cycles are conflict replays, profiles/r05_refharness_pmc.json).
//   B  "rotated"  table [ks][column] with the subspace in the low address bits, and the lanes of a wave SKEWED in time: in round j
//                 lane l looks up subspace m = (j - l) mod M of ITS OWN candidate, so the 32 lanes of a group read 32 different
//                 columns = 32 different banks whatever the code bytes are.  Every candidate is still summed by one lane in the
//                 order m = 0 .. M-1 (src/rii.h:386-394), so the distance bits do not change.  The codes are stored in tiles of
//                 64 rows with row r rotated by r mod M bytes (stored byte j = code[(j - r) mod M]): in round j every lane uses
//                 byte j of its registers.  A lane is between two candidates inside an iteration (tile k's row from round r mod M
//                 on, tile k-1's before): the two rows are merged once per iteration with v_bfi_b32.  Rows are doubled
//                 (column c holds subspace c mod M) for M <= 32 so the rotation's wrap-around is an immediate offset and the
//                 address is ONE v_perm_b32 (byte 1 = code byte, byte 0 = the lane's constant column offset, 256-byte rows).
//                 M = 64: 256-byte rows without doubling; the lane's column offset advances by one byte-wide add per round.
//   Both kernels return the first minimum in candidate order; the program checks that they agree bit for bit.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_rot.hip -o tools/ubench/gather_rot && tools/ubench/gather_rot
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static __device__ __forceinline__ uint32_t f32_ord(uint32_t u) { return u ^ ((uint32_t) ((int32_t) u >> 31) | 0x80000000u); }

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long x = __shfl_xor(k, o);
        k = x < k ? x : k;
    }
    return k;
}

// ------------------------------------------------------------------------------------------------------------------
// A: direct gather (the product's adc_lds_wide: four candidates per thread in flight, whole codes in registers first)
// ------------------------------------------------------------------------------------------------------------------
template <int M, int NT>
__global__ __launch_bounds__(NT) void direct_kernel(const uint8_t *__restrict__ codes, int64_t n_codes, int ncand, const float *__restrict__ tab,
                                                    unsigned long long *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < M * 256; i += NT) lds[i] = tab[i] + (float) (blockIdx.x & 7);
    __syncthreads();
    const int64_t base = ((int64_t) blockIdx.x * 4099) % (n_codes - ncand);
    float bestd = INFINITY;
    uint32_t bestp = 0xffffffffu;
    constexpr int MQ = M / 16;
    for (int p0 = tid; p0 < ncand; p0 += 4 * NT) {
        uint4 cv[4][MQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pos = p0 + u * NT;
            const uint4 *cp = reinterpret_cast<const uint4 *>(codes + (base + (pos < ncand ? pos : 0)) * M);
#pragma unroll
            for (int q = 0; q < MQ; ++q) cv[u][q] = cp[q];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float d = 0.f;
#pragma unroll
            for (int q = 0; q < MQ; ++q) {
                const uint32_t w[4] = {cv[u][q].x, cv[u][q].y, cv[u][q].z, cv[u][q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) d = __fadd_rn(d, lds[((q * 4 + i) * 4 + j) * 256 + ((w[i] >> (8 * j)) & 0xffu)]);
            }
            const int pos = p0 + u * NT;
            if (pos < ncand && d < bestd) { bestd = d; bestp = (uint32_t) pos; }
        }
    }
    unsigned long long key = bestp == 0xffffffffu ? ~0ull : (((unsigned long long) f32_ord(__float_as_uint(bestd)) << 32) | bestp);
    key = wave_min_u64(key);
    unsigned long long *red = reinterpret_cast<unsigned long long *>(lds + M * 256);      // (behind the table: the table starts at LDS address 0)
    if ((tid & 63) == 0) red[tid >> 6] = key;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / 64; ++w) key = red[w] < key ? red[w] : key;
        out[blockIdx.x] = key;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// B: rotated gather
// ------------------------------------------------------------------------------------------------------------------
template <int M> struct Rot {
    static constexpr bool kDoubled = M <= 32;
    static constexpr int kRowBytes = 256;                               // byte 1 of the address = the code byte
    static constexpr int kCopies = kDoubled ? (64 / M) : 1;             // columns per row = 64: subspace = column mod M
};

// v_cndmask with the lane mask in an SGPR pair (no v_cmp per round)
__device__ __forceinline__ float sel_mask(float if0, float if1, unsigned long long mask)
{
    float d;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(if0), "v"(if1), "s"(mask));
    return d;
}

template <int M, int NT, int VARIANT>
__global__ __launch_bounds__(NT) void rot_kernel(const uint8_t *__restrict__ rcodes, int64_t n_tiles_total, int ncand, const float *__restrict__ tab,
                                                 unsigned long long *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64, MW = M / 4;
    // table [ks][64 columns]: column c = subspace c mod M (M = 64: one copy)
    for (int i = tid; i < 256 * 64; i += NT) {
        const int ks = i >> 6, c = i & 63;
        lds[i] = tab[(c % M) * 256 + ks] + (float) (blockIdx.x & 7);
    }
    __syncthreads();
    const int phi = lane % M;
    const int ntile = (ncand + 63) / 64;
    const int64_t tile0 = (((int64_t) blockIdx.x * 4099) % (n_tiles_total * 64 - ncand)) / 64;     // (tile-aligned start)
    // lane constants
    uint32_t lowmask[MW];                       // bytes of dword d whose time slot j = 4d + b is < phi: still the previous tile's row
#pragma unroll
    for (int d = 0; d < MW; ++d) {
        uint32_t mk = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) mk |= (4 * d + b < phi) ? (0xffu << (8 * b)) : 0u;
        lowmask[d] = mk;
    }
    // column offset of the lane in round 0 (bytes): m = (0 - phi) mod M
    uint32_t laneoff = (uint32_t) (((M - phi) % M) * 4);
    if (M == 16) laneoff += (lane & 16) ? 64u : 0u;                     // second half of a 32-lane DS group: the other 16 banks
    float keep[VARIANT == 1 ? M : 1];
    if (VARIANT == 1) {
#pragma unroll
        for (int j = 0; j < M; ++j) keep[j] = (j == phi) ? 0.f : 1.f;
    }
    if ((uint32_t) reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float *) lds) != 0u) __builtin_trap();
    uint32_t prev[MW], cur[MW];
#pragma unroll
    for (int d = 0; d < MW; ++d) prev[d] = 0u;
    float acc = 0.f, fin = 0.f, bestd = INFINITY;
    uint32_t bestp = 0xffffffffu;
    // wave w takes tiles w, w + NW, ...; one drain iteration at the end (the last tile's candidates finish in it)
    const int my_n = ntile > wave ? (ntile - wave + NW - 1) / NW : 0;
    uint4 nxt[MW / 4];
    {
        const uint4 *cp = reinterpret_cast<const uint4 *>(rcodes + ((tile0 + wave) * 64 + lane) * (int64_t) M);
#pragma unroll
        for (int q = 0; q < MW / 4; ++q) nxt[q] = my_n > 0 ? cp[q] : make_uint4(0, 0, 0, 0);
    }
    int prev_pos = -1;                                                    // candidate (position) the lane finishes in this iteration
    for (int it = 0; it <= my_n; ++it) {
#pragma unroll
        for (int q = 0; q < MW / 4; ++q) { cur[4 * q] = nxt[q].x; cur[4 * q + 1] = nxt[q].y; cur[4 * q + 2] = nxt[q].z; cur[4 * q + 3] = nxt[q].w; }
        if (it + 1 < my_n) {
            const uint4 *cp = reinterpret_cast<const uint4 *>(rcodes + ((tile0 + wave + (int64_t) (it + 1) * NW) * 64 + lane) * (int64_t) M);
#pragma unroll
            for (int q = 0; q < MW / 4; ++q) nxt[q] = cp[q];
        }
        uint32_t x[MW];
#pragma unroll
        for (int d = 0; d < MW; ++d) asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(x[d]) : "v"(lowmask[d]), "v"(prev[d]), "v"(cur[d]));
        uint32_t off = laneoff;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            // address: byte 1 = code byte of time slot j, byte 0 = the lane's column offset
            const uint32_t sel = 0x0c0c0000u | ((4u + (j & 3)) << 8);      // D.b0 = S1.b0, D.b1 = S0.b(j & 3), D.b2 = D.b3 = 0
            uint32_t a;
            float t;
            if (Rot<M>::kDoubled) {
                a = __builtin_amdgcn_perm(x[j >> 2], laneoff, sel);
                t = *reinterpret_cast<const __attribute__((address_space(3))) float *>(a + 4 * j);   // (absolute: the table is at LDS address 0)
            } else {
                a = __builtin_amdgcn_perm(x[j >> 2], off, sel);
                t = *reinterpret_cast<const __attribute__((address_space(3))) float *>(a);
                // off.byte0 += 4 (wraps inside the 256-byte row)
                asm("v_add_u32_sdwa %0, %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(off) : "v"(4u));
            }
            unsigned long long eq = 1ull << (j & 63);
            if (M == 32) eq |= eq << 32;
            if (M == 16) eq |= (eq << 16) | (eq << 32) | (eq << 48);
            fin = sel_mask(fin, acc, eq);                                   // the lane whose previous candidate is complete keeps its sum
            if (VARIANT == 1) acc = __builtin_fmaf(acc, keep[j], t);        // ... and starts the next one from t (acc * 0 + t)
            else acc = __fadd_rn(sel_mask(acc, 0.f, eq), t);
        }
        // every lane finished the candidate of the previous tile during this iteration
        if (prev_pos >= 0 && prev_pos < ncand && fin < bestd) { bestd = fin; bestp = (uint32_t) prev_pos; }
        prev_pos = it < my_n ? (wave + it * NW) * 64 + lane : -1;
#pragma unroll
        for (int d = 0; d < MW; ++d) prev[d] = cur[d];
    }
    unsigned long long key = bestp == 0xffffffffu ? ~0ull : (((unsigned long long) f32_ord(__float_as_uint(bestd)) << 32) | bestp);
    key = wave_min_u64(key);
    unsigned long long *red = reinterpret_cast<unsigned long long *>(lds + 256 * 64);
    if (lane == 0) red[wave] = key;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w) key = red[w] < key ? red[w] : key;
        out[blockIdx.x] = key;
    }
}

// ------------------------------------------------------------------------------------------------------------------
template <int M, int NT> static float time_direct(const uint8_t *d_codes, int64_t n, int ncand, const float *d_tab, unsigned long long *d_out, int blocks, int reps)
{
    const size_t smem = (size_t) M * 1024 + 128;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(direct_kernel<M, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((direct_kernel<M, NT>), dim3(blocks), dim3(NT), smem, 0, d_codes, n, ncand, d_tab, d_out);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((direct_kernel<M, NT>), dim3(blocks), dim3(NT), smem, 0, d_codes, n, ncand, d_tab, d_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
template <int M, int NT, int V> static float time_rot(const uint8_t *d_rc, int64_t ntiles, int ncand, const float *d_tab, unsigned long long *d_out, int blocks, int reps)
{
    const size_t smem = 65536 + 128;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(rot_kernel<M, NT, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rot_kernel<M, NT, V>), dim3(blocks), dim3(NT), smem, 0, d_rc, ntiles, ncand, d_tab, d_out);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rot_kernel<M, NT, V>), dim3(blocks), dim3(NT), smem, 0, d_rc, ntiles, ncand, d_tab, d_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int M> static void run_shape(int ncand, int blocks)
{
    const int64_t n = 1 << 20;                         // code rows (L2 / Infinity-Cache resident, as the visited lists are)
    std::vector<uint8_t> codes((size_t) n * M), rc((size_t) n * M);
    uint32_t s = 12345u + M;
    for (auto &c : codes) { s = s * 1664525u + 1013904223u; c = (uint8_t) (s >> 24); }
    for (int64_t r = 0; r < n; ++r)
        for (int j = 0; j < M; ++j) rc[(size_t) r * M + j] = codes[(size_t) r * M + (((j - (int) (r & 63)) % M + M) % M)];
    std::vector<float> tab((size_t) M * 256);
    for (auto &t : tab) { s = s * 1664525u + 1013904223u; t = 1000.f + (float) (s >> 8) * (1.0f / 4096.f); }
    uint8_t *d_codes, *d_rc; float *d_tab; unsigned long long *d_oa, *d_ob;
    CK(hipMalloc(&d_codes, codes.size())); CK(hipMalloc(&d_rc, rc.size())); CK(hipMalloc(&d_tab, tab.size() * 4));
    CK(hipMalloc(&d_oa, blocks * 8)); CK(hipMalloc(&d_ob, blocks * 8));
    CK(hipMemcpy(d_codes, codes.data(), codes.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rc, rc.data(), rc.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    const int reps = 20;
    const double lookups = (double) blocks * ncand * M;
    std::vector<unsigned long long> ha(blocks), hb(blocks);
    auto report = [&](const char *name, float ms, unsigned long long *d_o, bool check) {
        int bad = -1;
        if (check) {
            CK(hipMemcpy(hb.data(), d_o, blocks * 8, hipMemcpyDeviceToHost));
            bad = 0;
            for (int i = 0; i < blocks; ++i) bad += ha[i] != hb[i];
        }
        printf("{\"M\": %d, \"ncand\": %d, \"blocks\": %d, \"kernel\": \"%s\", \"us\": %.2f, \"G_lookups_per_s\": %.1f, \"lds_TBps\": %.2f, \"mismatching_blocks\": %d}\n",
               M, ncand, blocks, name, ms * 1e3, lookups / ms / 1e6, lookups * 4 / ms / 1e9, bad);
    };
    // the blocks of both forms start at a tile-aligned row so that they see the same candidates
    float ms = time_direct<M, 256>(d_codes, n, ncand, d_tab, d_oa, blocks, reps);
    // (direct_kernel's start row: (b * 4099) % (n - ncand); rot_kernel's is that rounded down to a tile: make them equal by checking
    // against a direct run over the rounded start -- done by the host below)
    report("direct/256thr", ms, d_oa, false);
    if (M * 1024 * 2 <= 65536) { ms = time_direct<M, 512>(d_codes, n, ncand, d_tab, d_oa, blocks, reps); report("direct/512thr", ms, d_oa, false); }
    ms = time_rot<M, 256, 1>(d_rc, n / 64, ncand, d_tab, d_ob, blocks, reps); report("rotated/256thr/fma-keep", ms, d_ob, false);
    ms = time_rot<M, 256, 0>(d_rc, n / 64, ncand, d_tab, d_ob, blocks, reps); report("rotated/256thr/cndmask-reset", ms, d_ob, false);
    ms = time_rot<M, 512, 1>(d_rc, n / 64, ncand, d_tab, d_ob, blocks, reps); report("rotated/512thr/fma-keep", ms, d_ob, false);
    // bit-exactness: host replay of the first minimum over the candidates rot_kernel saw (tile-aligned start)
    CK(hipMemcpy(hb.data(), d_ob, blocks * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int b = 0; b < blocks; ++b) {
        const int64_t t0 = ((((int64_t) b * 4099) % (n - ncand)) / 64) * 64;
        float bestd = INFINITY; uint32_t bestp = 0xffffffffu;
        for (int p = 0; p < ncand; ++p) {
            volatile float d = 0.f;
            for (int m = 0; m < M; ++m) d = d + (tab[(size_t) m * 256 + codes[(size_t) (t0 + p) * M + m]] + (float) (b & 7));
            if (d < bestd) { bestd = d; bestp = (uint32_t) p; }
        }
        uint32_t u; __builtin_memcpy(&u, &bestd, 4);
        const unsigned long long key = ((unsigned long long) (u ^ 0x80000000u) << 32) | bestp;
        bad += key != hb[b];
        if (key != hb[b] && bad < 4) printf("  block %d: host %016llx device %016llx\n", b, key, hb[b]);
    }
    printf("{\"M\": %d, \"ncand\": %d, \"rotated_vs_host_sequential_sum\": \"%s\", \"mismatching_blocks\": %d}\n", M, ncand, bad ? "DIFFERENT" : "bit-identical", bad);
    CK(hipFree(d_codes)); CK(hipFree(d_rc)); CK(hipFree(d_tab)); CK(hipFree(d_oa)); CK(hipFree(d_ob));
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024;
    run_shape<64>(6016, blocks);          // the reference's harness setting: nlist + L = 1000 + 5000 lookups rows per query
    run_shape<32>(2048, blocks);          // configs[2]: 1024 + 977
    run_shape<32>(6016, blocks);
    run_shape<16>(16000, blocks);         // Deep-shaped shard: 8000 + 8000
    return 0;
}
