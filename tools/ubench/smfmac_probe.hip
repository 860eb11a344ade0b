// Probe: which dense-K element does stored byte s with 2-bit index j of the sparse A operand of
// v_smfmac_i32_16x16x128_i8 select, in terms of (lane group, byte) of the dense B operand?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/smfmac_probe.hip -o /tmp/smfmac_probe && /tmp/smfmac_probe
// Every lane carries the same A (one stored byte = 1, index j) -> all rows equal; B is non-zero in ONE lane group, byte t
// holding t + 1 -> D = (selected byte) + 1, or 0 when the selected element lives in another lane group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

__global__ void probe(int s, int j, int gsel, int *out)
{
    const int l = threadIdx.x;
    v4i a = {0, 0, 0, 0};
    a[s >> 2] = 1 << (8 * (s & 3));
    unsigned idx = 0;
    for (int f = 0; f < 16; ++f) {
        int v = (f == s) ? j : ((f ^ 1) == s ? ((j + 2) & 3) : (f & 1 ? 1 : 0));
        idx |= (unsigned) v << (2 * f);
    }
    v8i b = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((l >> 4) == gsel)
        for (int t = 0; t < 32; ++t) b[t >> 2] |= (t + 1) << (8 * (t & 3));
    v4i c = {0, 0, 0, 0};
    v4i d = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a, b, c, (int) idx, 0, 0);
    out[l * 4 + 0] = d[0]; out[l * 4 + 1] = d[1]; out[l * 4 + 2] = d[2]; out[l * 4 + 3] = d[3];
}

int main()
{
    int *d_out;
    if (hipMalloc(&d_out, 256 * sizeof(int)) != hipSuccess) return 1;
    std::vector<int> o(256);
    printf("stored byte s, index j -> selected dense byte (+1) per lane group of B [g0 g1 g2 g3]; uniform = all 256 outputs equal\n");
    for (int s = 0; s < 16; ++s)
        for (int j = 0; j < 4; ++j) {
            printf("s=%2d j=%d:", s, j);
            for (int g = 0; g < 4; ++g) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, j, g, d_out);
                (void) hipMemcpy(o.data(), d_out, 256 * sizeof(int), hipMemcpyDeviceToHost);
                bool uni = true;
                for (int i = 1; i < 256; ++i) uni = uni && (o[i] == o[0]);
                printf(" %3d%s", o[0], uni ? "" : "*");
            }
            printf("\n");
        }
    printf("hip status: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
