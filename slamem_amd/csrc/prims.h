// prims.h -- device-wide building blocks used by the index build and the MEM output path.
// Narrow internal interface; everything behind it is hand-written (LDS radix passes in radix_sort.hip, tile scans and
// flagged compaction in scan.hip): no library primitive is linked into libslamem_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slamem {

// All functions follow the two-call convention: tmp == nullptr -> only tmp_bytes is written.

// Stable LSD radix sort of (u64 key, u32 value) pairs on key bits [begin_bit, end_bit).
// Result is in keys_out / vals_out; the *_in buffers are clobbered.
hipError_t sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, uint64_t* keys_in, uint64_t* keys_out,
                              uint32_t* vals_in, uint32_t* vals_out, size_t n, int begin_bit, int end_bit,
                              hipStream_t stream);

// Hand-written exclusive sum scan of u32 (radix_sort.hip); in-place capable; tmp holds scan_u32_tmp_words(n) words.
uint64_t scan_u32_tmp_words(uint64_t n);
hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* tmp, hipStream_t stream);

hipError_t scan_max_inclusive_u32(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint32_t* out, size_t n,
                                  hipStream_t stream);
// out[i] = sum_{j<i} in[j]  (u32 in, u64 out); out has n+1 entries, out[n] = total.
hipError_t scan_sum_exclusive_u32_u64(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint64_t* out, size_t n,
                                      hipStream_t stream);
// The same with u64 in: out has n+1 entries, the caller keeps in[n] = 0, out[n] = total.
hipError_t scan_sum_exclusive_u64(void* tmp, size_t& tmp_bytes, const uint64_t* in, uint64_t* out, size_t n, hipStream_t stream);
// Four independent exclusive sums over uint4 lanes (FM block rank samples).
hipError_t scan_sum_exclusive_uint4(void* tmp, size_t& tmp_bytes, const uint4* in, uint4* out, size_t n,
                                    hipStream_t stream);
// out = { in[i] : flags[i] != 0 }, order preserved; *count_out_dev = number selected.
hipError_t select_flagged_u32(void* tmp, size_t& tmp_bytes, const uint32_t* in, const uint8_t* flags,
                              uint32_t* out, uint32_t* count_out_dev, size_t n, hipStream_t stream);
// out = { i : flags[i] != 0 } for i in [0, n)
hipError_t select_indices_u32(void* tmp, size_t& tmp_bytes, const uint8_t* flags, uint32_t* out,
                              uint32_t* count_out_dev, size_t n, hipStream_t stream);

}  // namespace slamem
