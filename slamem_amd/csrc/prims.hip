// prims.hip -- device-wide scans / compaction (rocPRIM, AMD's native primitive library) and the
// radix-sort entry point.  See prims.h.
#include "prims.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace slamem {

struct U4Plus {
    __host__ __device__ uint4 operator()(const uint4& a, const uint4& b) const {
        return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
};

hipError_t radix_sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, uint64_t* keys_in, uint64_t* keys_out,
                                    uint32_t* vals_in, uint32_t* vals_out, size_t n, int begin_bit, int end_bit,
                                    hipStream_t stream);  // radix_sort.hip (hand-written)

// SLAMEM_SORT=rocprim selects rocPRIM's onesweep sort (kept for A/B timing and as a cross-check of radix_sort.hip)
static bool use_rocprim_sort() {
    static const bool v = [] { const char* e = getenv("SLAMEM_SORT"); return e && strcmp(e, "rocprim") == 0; }();
    return v;
}

hipError_t sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, uint64_t* keys_in, uint64_t* keys_out,
                              uint32_t* vals_in, uint32_t* vals_out, size_t n, int begin_bit, int end_bit,
                              hipStream_t stream) {
    if (tmp == nullptr) {  // size query: enough for either implementation
        size_t a = 0, b = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, a, keys_in, keys_out, vals_in, vals_out, n,
                                                 (unsigned)begin_bit, (unsigned)end_bit, stream);
        if (e != hipSuccess) return e;
        e = radix_sort_pairs_u64_u32(nullptr, b, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
        tmp_bytes = a > b ? a : b;
        return e;
    }
    if (use_rocprim_sort())
        return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n,
                                         (unsigned)begin_bit, (unsigned)end_bit, stream);
    return radix_sort_pairs_u64_u32(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}

hipError_t scan_max_inclusive_u32(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint32_t* out, size_t n,
                                  hipStream_t stream) {
    return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::maximum<uint32_t>(), stream);
}

hipError_t scan_sum_exclusive_u32_u64(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint64_t* out, size_t n,
                                      hipStream_t stream) {
    auto in64 = rocprim::make_transform_iterator(
        in, [] __host__ __device__(uint32_t v) -> uint64_t { return (uint64_t)v; });
    // n+1 outputs: the extra element reads one past `in`; callers pad `in` with one zero element.
    return rocprim::exclusive_scan(tmp, tmp_bytes, in64, out, (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), stream);
}

hipError_t scan_sum_exclusive_uint4(void* tmp, size_t& tmp_bytes, const uint4* in, uint4* out, size_t n,
                                    hipStream_t stream) {
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, make_uint4(0, 0, 0, 0), n, U4Plus(), stream);
}

hipError_t select_flagged_u32(void* tmp, size_t& tmp_bytes, const uint32_t* in, const uint8_t* flags,
                              uint32_t* out, uint32_t* count_out_dev, size_t n, hipStream_t stream) {
    return rocprim::select(tmp, tmp_bytes, in, flags, out, count_out_dev, n, stream);
}

hipError_t select_indices_u32(void* tmp, size_t& tmp_bytes, const uint8_t* flags, uint32_t* out,
                              uint32_t* count_out_dev, size_t n, hipStream_t stream) {
    rocprim::counting_iterator<uint32_t> it(0);
    return rocprim::select(tmp, tmp_bytes, it, flags, out, count_out_dev, n, stream);
}

}  // namespace slamem
