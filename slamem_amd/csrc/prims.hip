// prims.hip -- entry point of the pair sort used by the suffix-array construction: the hand-written LSD radix sort
// of radix_sort.hip.  (Scans and compaction: scan.hip.  No library primitives are linked into libslamem_hip.so.)
#include "prims.h"

namespace slamem {

hipError_t radix_sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, uint64_t* keys_in, uint64_t* keys_out,
                                    uint32_t* vals_in, uint32_t* vals_out, size_t n, int begin_bit, int end_bit,
                                    hipStream_t stream);  // radix_sort.hip

hipError_t sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, uint64_t* keys_in, uint64_t* keys_out,
                              uint32_t* vals_in, uint32_t* vals_out, size_t n, int begin_bit, int end_bit,
                              hipStream_t stream) {
    return radix_sort_pairs_u64_u32(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}

}  // namespace slamem
