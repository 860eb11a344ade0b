// sortsel.hip -- general top-k fallback: full segmented radix sort (rocPRIM) of the packed
// (orderable distance bits << 32 | index) keys emitted by scan_kernel<WRITE_KEYS>.  Used for topk > 1 on
// the linear path (incl. topk == N, rii/rii.py:280-282); the top-1 path never comes here.
#include <cstring>
#include <cstdlib>
#include "rii_internal.h"
#include <rocprim/rocprim.hpp>

namespace riiamd {

struct MulOffset {
    unsigned int len;
    __host__ __device__ unsigned int operator()(unsigned int i) const { return i * len; }
};

hipError_t segmented_sort_keys(unsigned long long *d_keys_in, unsigned long long *d_keys_out, int64_t segs,
                               int64_t len, void **d_temp, size_t *temp_bytes, hipStream_t st)
{
    if (segs == 0 || len == 0) return hipSuccess;
    if (segs * len >= (int64_t) 1 << 31) return hipErrorInvalidValue;   // caller chunks the batch
    auto beg = rocprim::make_transform_iterator(rocprim::make_counting_iterator<unsigned int>(0u),
                                                MulOffset{(unsigned int) len});
    auto end = rocprim::make_transform_iterator(rocprim::make_counting_iterator<unsigned int>(1u),
                                                MulOffset{(unsigned int) len});
    size_t need = 0;
    hipError_t e = rocprim::segmented_radix_sort_keys(nullptr, need, d_keys_in, d_keys_out,
                                                      (unsigned int) (segs * len), (unsigned int) segs, beg, end,
                                                      0, 64, st);
    if (e != hipSuccess) return e;
    if (need > *temp_bytes) {
        if (*d_temp) { e = hipFree(*d_temp); if (e != hipSuccess) return e; }
        *d_temp = nullptr; *temp_bytes = 0;
        e = hipMalloc(d_temp, need);
        if (e != hipSuccess) return e;
        *temp_bytes = need;
    }
    return rocprim::segmented_radix_sort_keys(*d_temp, need, d_keys_in, d_keys_out, (unsigned int) (segs * len),
                                              (unsigned int) segs, beg, end, 0, 64, st);
}

}  // namespace riiamd
