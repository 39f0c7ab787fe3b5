// radix_sort.hip -- hand-written LSD radix sort of (u64 key, u32 value) pairs for gfx950: the sort behind the
// suffix-array construction (K2 of index_build.hip; replaces the pointer-chasing SortLMSs / InducedSort of
// bwtindex.c:787-1310).  8-bit digits; per pass
//   k_radix_hist     one 4096-element tile per workgroup, 256-bin histogram in LDS (LDS atomics), written
//                    bin-major so that ONE exclusive scan of the table yields every (bin, tile) output offset
//   exclusive scan   of the 256 x tiles table (own three-phase scan below)
//   k_radix_scatter  re-reads the tile (coalesced), ranks every element among equal digits with wave ballots
//                    (8 __ballot per element, no atomics: stable), stages the tile in LDS sorted by digit -- the
//                    "LDS-staged radix buckets" -- and writes every bucket as one contiguous, coalesced run
// Traffic per pass: 8 B (histogram) + 12 B read + 12 B written per element; everything streams.
#include "prims.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace slamem {

namespace {

constexpr uint32_t kTile = 4096;   // elements per workgroup
constexpr uint32_t kItems = 16;    // per lane (4 waves x 64 lanes x 16)
constexpr uint32_t kScanTile = 2048;

inline unsigned grid_for(uint64_t items, unsigned block) { return (unsigned)((items + block - 1) / block); }

// ---------------------------------------------------------------------------------------------------
// exclusive scan of u32 (three-phase, recursive on the block sums)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if ((int)lane >= d) v += o;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* sh /* >= 4 words */) {
    uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63u) sh[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < w; i++) base += sh[i];
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + inc - v;
}

__global__ void __launch_bounds__(256) k_scan_sums(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t sh[4];
    uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    uint32_t v = 0;
    for (uint32_t k = 0; k < kScanTile / 256; k++) {
        uint64_t i = base + k * 256 + threadIdx.x;
        if (i < n) v += in[i];
    }
    uint32_t total;
    (void)block_excl_scan(v, &total, sh);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// out[i] = offset[block] + exclusive prefix inside the block; each thread owns 8 consecutive elements
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t* __restrict__ in, uint64_t n,
                                                    const uint32_t* __restrict__ block_off /* nullable */,
                                                    uint32_t* __restrict__ out) {
    __shared__ uint32_t sh[4];
    uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint32_t total;
    uint32_t pre = block_excl_scan(s, &total, sh) + (block_off ? block_off[blockIdx.x] : 0u);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (base + k < n) out[base + k] = pre;
        pre += v[k];
    }
}

}  // namespace

// in-place capable (in == out); tmp must hold scan_u32_tmp_words(n) words
uint64_t scan_u32_tmp_words(uint64_t n) {
    uint64_t w = 0;
    while (n > kScanTile) { n = (n + kScanTile - 1) / kScanTile; w += n; }
    return w + 1;
}

hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* tmp, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + kScanTile - 1) / kScanTile;
    if (blocks == 1) {
        hipLaunchKernelGGL(k_scan_apply, dim3(1), dim3(256), 0, stream, in, n, (const uint32_t*)nullptr, out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)blocks), dim3(256), 0, stream, in, n, tmp);
    hipError_t e = exclusive_scan_u32(tmp, tmp, blocks, tmp + blocks, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)blocks), dim3(256), 0, stream, in, n, (const uint32_t*)tmp, out);
    return hipGetLastError();
}

namespace {

// ---------------------------------------------------------------------------------------------------
// radix passes
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_radix_hist(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift,
                                                    uint32_t mask, uint32_t tiles, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * kTile;
#pragma unroll 4
    for (uint32_t k = 0; k < kItems; k++) {
        uint64_t i = base + k * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    ghist[(uint64_t)threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];  // bin-major
}

__global__ void __launch_bounds__(256) k_radix_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                       uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                       uint64_t n, uint32_t shift, uint32_t mask, uint32_t tiles,
                                                       const uint32_t* __restrict__ gofs) {
    __shared__ uint64_t skeys[kTile];
    __shared__ uint32_t svals[kTile];
    __shared__ uint32_t wcnt[4][256];   // per-wave digit counts, then per-wave digit offsets
    __shared__ uint32_t dbase[256];     // start of every digit's bucket inside the staged tile
    __shared__ uint32_t sh[4];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint64_t tile_base = (uint64_t)blockIdx.x * kTile;
    const uint64_t wave_base = tile_base + (uint64_t)w * (kTile / 4);
    const unsigned long long lt = (1ull << lane) - 1ull;

    for (uint32_t i = threadIdx.x; i < 4 * 256; i += 256) (&wcnt[0][0])[i] = 0;
    __syncthreads();

    uint64_t key[kItems];
    uint32_t val[kItems], rank[kItems];
    // the wave's 1024 elements in striped order: element (i, lane) = wave_base + i*64 + lane; processing i in
    // increasing order with lanes in order keeps the sort stable
#pragma unroll
    for (uint32_t i = 0; i < kItems; i++) {
        uint64_t idx = wave_base + i * 64u + lane;
        bool valid = idx < n;
        key[i] = valid ? keys_in[idx] : 0ull;
        val[i] = valid ? vals_in[idx] : 0u;
    }
    volatile uint32_t* mycnt = wcnt[w];
#pragma unroll
    for (uint32_t i = 0; i < kItems; i++) {
        bool valid = wave_base + i * 64u + lane < n;
        uint32_t d = (uint32_t)(key[i] >> shift) & mask;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {  // lanes holding the same digit
            bool bit = (d >> b) & 1u;
            unsigned long long bal = __ballot(bit && valid);
            m &= bit ? bal : ~bal;
        }
        uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
        uint32_t prev = 0;
        if (valid && lane == leader) {
            prev = mycnt[d];
            mycnt[d] = prev + (uint32_t)__popcll(m);
        }
        prev = __shfl(prev, valid ? (int)leader : (int)lane);
        rank[i] = prev + (uint32_t)__popcll(m & lt);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // digit t: offsets of the four waves inside the bucket, bucket starts by a block scan over digits
        uint32_t t = threadIdx.x;
        uint32_t c0 = wcnt[0][t], c1 = wcnt[1][t], c2 = wcnt[2][t], c3 = wcnt[3][t];
        uint32_t total_all;
        uint32_t start = block_excl_scan(c0 + c1 + c2 + c3, &total_all, sh);
        dbase[t] = start;
        wcnt[0][t] = start;
        wcnt[1][t] = start + c0;
        wcnt[2][t] = start + c0 + c1;
        wcnt[3][t] = start + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kItems; i++) {  // stage the tile in LDS, sorted by digit
        if (wave_base + i * 64u + lane < n) {
            uint32_t d = (uint32_t)(key[i] >> shift) & mask;
            uint32_t p = wcnt[w][d] + rank[i];
            skeys[p] = key[i];
            svals[p] = val[i];
        }
    }
    __syncthreads();
    uint32_t count = n - tile_base < kTile ? (uint32_t)(n - tile_base) : kTile;
#pragma unroll 4
    for (uint32_t k = 0; k < kItems; k++) {  // every bucket leaves as one contiguous run
        uint32_t p = k * 256 + threadIdx.x;
        if (p < count) {
            uint64_t kk = skeys[p];
            uint32_t d = (uint32_t)(kk >> shift) & mask;
            uint64_t dst = (uint64_t)gofs[(uint64_t)d * tiles + blockIdx.x] + (p - dbase[d]);
            keys_out[dst] = kk;
            vals_out[dst] = svals[p];
        }
    }
}

}  // namespace

hipError_t sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, uint64_t* keys_in, uint64_t* keys_out,
                                    uint32_t* vals_in, uint32_t* vals_out, size_t n, int begin_bit, int end_bit,
                                    hipStream_t stream) {
    const uint64_t tiles = (n + kTile - 1) / kTile;
    const uint64_t table = 256 * (tiles ? tiles : 1);
    const size_t need = (table + scan_u32_tmp_words(table) + 64) * sizeof(uint32_t);
    if (tmp == nullptr) { tmp_bytes = need; return hipSuccess; }
    if (tmp_bytes < need) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    if (n >= 0xFFFFFFFFull) return hipErrorInvalidValue;  // table offsets are 32-bit
    uint32_t* ghist = static_cast<uint32_t*>(tmp);
    uint32_t* stmp = ghist + table;
    uint64_t *ksrc = keys_in, *kdst = keys_out;
    uint32_t *vsrc = vals_in, *vdst = vals_out;
    for (int bit = begin_bit; bit < end_bit; bit += 8) {
        int width = end_bit - bit < 8 ? end_bit - bit : 8;
        uint32_t mask = (1u << width) - 1u;
        hipLaunchKernelGGL(k_radix_hist, dim3((unsigned)tiles), dim3(256), 0, stream, ksrc, (uint64_t)n, (uint32_t)bit, mask,
                           (uint32_t)tiles, ghist);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = exclusive_scan_u32(ghist, ghist, table, stmp, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_radix_scatter, dim3((unsigned)tiles), dim3(256), 0, stream, ksrc, vsrc, kdst, vdst, (uint64_t)n,
                           (uint32_t)bit, mask, (uint32_t)tiles, ghist);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        uint64_t* tk = ksrc; ksrc = kdst; kdst = tk;
        uint32_t* tv = vsrc; vsrc = vdst; vdst = tv;

    }
    if (ksrc != keys_out) {  // even number of passes (or none): the result sits in the input buffers
        hipError_t e = hipMemcpyAsync(keys_out, ksrc, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(vals_out, vsrc, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace slamem
