// scan.hip -- hand-written device-wide scans and stream compaction (gfx950), the building blocks of the index
// build and of the MEM output path besides the radix sort (radix_sort.hip):
//   inclusive max-scan of u32        group heads -> group head position            (K2, index_build.hip)
//   exclusive sum of u32 -> u64      per-item MEM counts -> output offsets         (K9, mem_search.hip)
//   exclusive sum of uint4           per-block letter counts -> rank samples       (K3, index_build.hip)
//   flagged compaction               still-ambiguous suffix groups, N rows, surviving work items
// Three phases per scan: per-tile reduction (2048 elements per workgroup), recursive scan of the tile totals,
// per-tile scan with the tile's prefix.  Everything streams with coalesced accesses; wave64 shuffles inside a wave,
// one LDS word per wave across the four waves of a workgroup.
#include "prims.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slamem {
namespace {

constexpr uint32_t kTile = 2048;  // 256 threads x 8 consecutive elements

inline unsigned tiles_of(uint64_t n) { return (unsigned)((n + kTile - 1) / kTile); }

// ---- element types and operators ------------------------------------------------------------------------
struct SumU32 { using T = uint32_t; static __device__ T id() { return 0u; } static __device__ T op(T a, T b) { return a + b; } };
struct MaxU32 { using T = uint32_t; static __device__ T id() { return 0u; } static __device__ T op(T a, T b) { return a > b ? a : b; } };
struct SumU64 { using T = uint64_t; static __device__ T id() { return 0ull; } static __device__ T op(T a, T b) { return a + b; } };
struct SumU4 {
    using T = uint4;
    static __device__ T id() { return make_uint4(0, 0, 0, 0); }
    static __device__ T op(T a, T b) { return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
};

__device__ __forceinline__ uint32_t shfl_up_t(uint32_t v, int d) { return __shfl_up(v, d); }
__device__ __forceinline__ uint64_t shfl_up_t(uint64_t v, int d) {
    uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint4 shfl_up_t(uint4 v, int d) {
    return make_uint4(__shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.z, d), __shfl_up(v.w, d));
}

// input adaptors: what element i of the scanned sequence is
struct LoadU32 { const uint32_t* p; __device__ uint32_t operator()(uint64_t i) const { return p[i]; } };
struct LoadU32As64 { const uint32_t* p; __device__ uint64_t operator()(uint64_t i) const { return (uint64_t)p[i]; } };
struct LoadU64 { const uint64_t* p; __device__ uint64_t operator()(uint64_t i) const { return p[i]; } };
struct LoadU4 { const uint4* p; __device__ uint4 operator()(uint64_t i) const { return p[i]; } };
struct LoadFlag { const uint8_t* p; __device__ uint32_t operator()(uint64_t i) const { return p[i] ? 1u : 0u; } };

// inclusive scan of one value per thread across the workgroup; *total = reduction of all 256 values
template <class M>
__device__ __forceinline__ typename M::T block_incl_scan(typename M::T v, typename M::T* total, typename M::T* sh) {
    using T = typename M::T;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = shfl_up_t(v, d);
        if ((int)lane >= d) v = M::op(o, v);
    }
    if (lane == 63u) sh[w] = v;
    __syncthreads();
    T base = M::id();
    for (uint32_t i = 0; i < w; i++) base = M::op(base, sh[i]);
    *total = M::op(M::op(sh[0], sh[1]), M::op(sh[2], sh[3]));
    __syncthreads();
    return M::op(base, v);
}

template <class M, class Load>
__global__ void __launch_bounds__(256) k_tile_reduce(Load ld, uint64_t n, typename M::T* __restrict__ sums) {
    using T = typename M::T;
    __shared__ T sh[4];
    uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * 8;
    T v = M::id();
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (base + k < n) v = M::op(v, ld(base + k));
    T total;
    (void)block_incl_scan<M>(v, &total, sh);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

template <class M, class Load, bool kInclusive>
__global__ void __launch_bounds__(256) k_tile_scan(Load ld, uint64_t n, const typename M::T* __restrict__ tile_prefix /* nullable */,
                                                   typename M::T* __restrict__ out) {
    using T = typename M::T;
    __shared__ T sh[4];
    uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * 8;
    T v[8], s = M::id();
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = base + k < n ? ld(base + k) : M::id(); s = M::op(s, v[k]); }
    T total;
    T incl = block_incl_scan<M>(s, &total, sh);
    // exclusive prefix of this thread = inclusive prefix of the previous thread
    T prev = shfl_up_t(incl, 1);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    __shared__ T wlast[4];
    if (lane == 63u) wlast[w] = incl;
    __syncthreads();
    T pre = lane ? prev : (w ? wlast[w - 1] : M::id());
    if (tile_prefix) pre = M::op(tile_prefix[blockIdx.x], pre);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        T after = M::op(pre, v[k]);
        if (base + k < n) out[base + k] = kInclusive ? after : pre;
        pre = after;
    }
}

// exclusive scan of the tile totals, in place, recursive (the totals of 2048 tiles fit one tile)
template <class M, class LoadT>
hipError_t scan_totals(typename M::T* totals, uint64_t count, typename M::T* tmp, hipStream_t stream) {
    using T = typename M::T;
    if (count == 0) return hipSuccess;
    unsigned tiles = tiles_of(count);
    if (tiles == 1) {
        hipLaunchKernelGGL((k_tile_scan<M, LoadT, false>), dim3(1), dim3(256), 0, stream, LoadT{totals}, count, (const T*)nullptr, totals);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((k_tile_reduce<M, LoadT>), dim3(tiles), dim3(256), 0, stream, LoadT{totals}, count, tmp);
    hipError_t e = scan_totals<M, LoadT>(tmp, tiles, tmp + tiles, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_tile_scan<M, LoadT, false>), dim3(tiles), dim3(256), 0, stream, LoadT{totals}, count, (const T*)tmp, totals);
    return hipGetLastError();
}

uint64_t totals_words(uint64_t n) {  // elements of T needed for all levels of tile totals
    uint64_t w = 0;
    while (n > 1) { n = (n + kTile - 1) / kTile; w += n; if (n == 1) break; }
    return w + 2;
}

template <class M, class Load, class LoadT, bool kInclusive>
hipError_t scan_impl(void* tmp, size_t& tmp_bytes, Load ld, typename M::T* out, uint64_t n, hipStream_t stream) {
    using T = typename M::T;
    const size_t need = totals_words(n) * sizeof(T) + 64;
    if (tmp == nullptr) { tmp_bytes = need; return hipSuccess; }
    if (tmp_bytes < need) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    T* totals = static_cast<T*>(tmp);
    unsigned tiles = tiles_of(n);
    if (tiles == 1) {
        hipLaunchKernelGGL((k_tile_scan<M, Load, kInclusive>), dim3(1), dim3(256), 0, stream, ld, n, (const T*)nullptr, out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((k_tile_reduce<M, Load>), dim3(tiles), dim3(256), 0, stream, ld, n, totals);
    hipError_t e = scan_totals<M, LoadT>(totals, tiles, totals + tiles, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_tile_scan<M, Load, kInclusive>), dim3(tiles), dim3(256), 0, stream, ld, n, (const T*)totals, out);
    return hipGetLastError();
}

// ---- compaction ------------------------------------------------------------------------------------------
// out[rank of i among flagged] = in ? in[i] : i, for flagged i; ranks from the tile prefix + an in-tile scan
__global__ void __launch_bounds__(256) k_compact(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ in /* nullable */,
                                                 uint64_t n, const uint32_t* __restrict__ tile_prefix /* nullable */,
                                                 uint32_t* __restrict__ out) {
    __shared__ uint32_t sh[4];
    __shared__ uint32_t wlast[4];
    uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * 8;
    uint32_t f[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { f[k] = (base + k < n && flags[base + k]) ? 1u : 0u; s += f[k]; }
    uint32_t total;
    uint32_t incl = block_incl_scan<SumU32>(s, &total, sh);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t prev = __shfl_up(incl, 1);
    if (lane == 63u) wlast[w] = incl;
    __syncthreads();
    uint32_t pos = (lane ? prev : (w ? wlast[w - 1] : 0u)) + (tile_prefix ? tile_prefix[blockIdx.x] : 0u);
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (f[k]) out[pos++] = in ? in[base + k] : (uint32_t)(base + k);
}

__global__ void k_store_count(const uint32_t* __restrict__ tile_prefix, const uint32_t* __restrict__ last_total, uint32_t tiles,
                              uint32_t* __restrict__ count_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *count_out = tile_prefix[tiles - 1] + *last_total;
}

hipError_t compact_impl(void* tmp, size_t& tmp_bytes, const uint32_t* in, const uint8_t* flags, uint32_t* out,
                        uint32_t* count_out_dev, uint64_t n, hipStream_t stream) {
    const unsigned tiles = tiles_of(n ? n : 1);
    const size_t need = ((size_t)tiles * 2 + totals_words(tiles) + 16) * sizeof(uint32_t);
    if (tmp == nullptr) { tmp_bytes = need; return hipSuccess; }
    if (tmp_bytes < need) return hipErrorInvalidValue;
    if (n == 0) return hipMemsetAsync(count_out_dev, 0, 4, stream);
    uint32_t* totals = static_cast<uint32_t*>(tmp);  // flagged elements per tile
    uint32_t* prefix = totals + tiles;                // their exclusive scan
    hipLaunchKernelGGL((k_tile_reduce<SumU32, LoadFlag>), dim3(tiles), dim3(256), 0, stream, LoadFlag{flags}, n, totals);
    hipError_t e = hipMemcpyAsync(prefix, totals, (size_t)tiles * 4, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    e = scan_totals<SumU32, LoadU32>(prefix, tiles, prefix + tiles, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_compact, dim3(tiles), dim3(256), 0, stream, flags, in, n, (const uint32_t*)prefix, out);
    hipLaunchKernelGGL(k_store_count, dim3(1), dim3(64), 0, stream, (const uint32_t*)prefix, (const uint32_t*)(totals + tiles - 1), tiles,
                       count_out_dev);
    return hipGetLastError();
}

}  // namespace

hipError_t scan_max_inclusive_u32(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t stream) {
    return scan_impl<MaxU32, LoadU32, LoadU32, true>(tmp, tmp_bytes, LoadU32{in}, out, n, stream);
}

// out has n+1 entries: the caller keeps in[n] = 0, so out[n] is the total
hipError_t scan_sum_exclusive_u32_u64(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint64_t* out, size_t n, hipStream_t stream) {
    return scan_impl<SumU64, LoadU32As64, LoadU64, false>(tmp, tmp_bytes, LoadU32As64{in}, out, n + 1, stream);
}

hipError_t scan_sum_exclusive_u64(void* tmp, size_t& tmp_bytes, const uint64_t* in, uint64_t* out, size_t n, hipStream_t stream) {
    return scan_impl<SumU64, LoadU64, LoadU64, false>(tmp, tmp_bytes, LoadU64{in}, out, n + 1, stream);
}

hipError_t scan_sum_exclusive_uint4(void* tmp, size_t& tmp_bytes, const uint4* in, uint4* out, size_t n, hipStream_t stream) {
    return scan_impl<SumU4, LoadU4, LoadU4, false>(tmp, tmp_bytes, LoadU4{in}, out, n, stream);
}

hipError_t select_flagged_u32(void* tmp, size_t& tmp_bytes, const uint32_t* in, const uint8_t* flags, uint32_t* out,
                              uint32_t* count_out_dev, size_t n, hipStream_t stream) {
    return compact_impl(tmp, tmp_bytes, in, flags, out, count_out_dev, n, stream);
}

hipError_t select_indices_u32(void* tmp, size_t& tmp_bytes, const uint8_t* flags, uint32_t* out, uint32_t* count_out_dev,
                              size_t n, hipStream_t stream) {
    return compact_impl(tmp, tmp_bytes, nullptr, flags, out, count_out_dev, n, stream);
}

}  // namespace slamem
