// synth.hip -- bench / test support (NOT part of the product ABI): the SURVEY.md Appendix C.2 generator
// on the GPU, so that bench.py's inputs are resident in HBM when the timed region starts.
// Counter-based splitmix64: draw k of a stream = mix(seed + (k+1)*GAMMA); same values as slamem_amd/synth.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t draw(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void k_ref(uint8_t* out, uint64_t n, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = "ACGT"[draw(seed, i) & 3];
}

// one thread per read base; reads stored back to back, `length` bytes each
__global__ void k_reads(const uint8_t* ref, uint64_t n, uint8_t* out, uint64_t first, uint64_t count, uint32_t length,
                        uint32_t sub_thr, uint64_t seed, uint32_t rc_percent, uint64_t avoid_at, uint64_t avoid_len) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * length) return;
    uint64_t r = t / length;
    uint32_t i = (uint32_t)(t - r * length);
    uint64_t base = n + (first + r) * (uint64_t)(length + 2);
    uint64_t p = draw(seed, base) % (n - length + 1);
    if (avoid_len && p + length > avoid_at && p < avoid_at + avoid_len) p += avoid_len + length;  // not out of a block of N
    bool flip = (draw(seed, base + 1 + length) % 100) < rc_percent;
    uint8_t c = ref[p + i];
    uint64_t x = draw(seed, base + 1 + i);
    if ((uint32_t)x < sub_thr) {
        uint32_t code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
        c = "CGTAGTACTACG"[code * 3 + (uint32_t)((x >> 32) % 3)];
    }
    if (flip) {
        c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
        out[r * length + (length - 1 - i)] = c;
    } else {
        out[r * length + i] = c;
    }
}

extern "C" int slamem_synth_reference(void* out_dev, uint64_t n, uint64_t seed, void* stream) {
    hipLaunchKernelGGL(k_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint8_t*)out_dev, n, seed);
    return (int)hipGetLastError();
}

extern "C" int slamem_synth_reads_avoid(const void* ref_dev, uint64_t n, void* out_dev, uint64_t first, uint64_t count,
                                        uint32_t length, double sub, uint64_t seed, uint32_t rc_percent, uint64_t avoid_at,
                                        uint64_t avoid_len, void* stream) {
    uint32_t thr = (uint32_t)(uint64_t)(sub * 4294967296.0);
    uint64_t total = count * length;
    hipLaunchKernelGGL(k_reads, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)ref_dev, n, (uint8_t*)out_dev, first, count, length, thr, seed, rc_percent, avoid_at, avoid_len);
    return (int)hipGetLastError();
}

extern "C" int slamem_synth_reads(const void* ref_dev, uint64_t n, void* out_dev, uint64_t first, uint64_t count,
                                  uint32_t length, double sub, uint64_t seed, uint32_t rc_percent, void* stream) {
    return slamem_synth_reads_avoid(ref_dev, n, out_dev, first, count, length, sub, seed, rc_percent, 0, 0, stream);
}

// ---------------------------------------------------------------------------------------------------
// Repeat model of SURVEY.md 8(d) for the chromosome-sized configurations (BASELINE.json configs[3] and [4]):
// 0.5 % of the text is overwritten by copies of 1-10 kbp segments with 1 % substitutions.  Fully determined by
// (n, seed) -- same values as slamem_amd/synth.py::plant_repeats:
//   stream P = seed + kRepeatSalt, stream M = seed + kRepeatSalt + 1
//   segment k: len = 1000 + P[3k] % 9001, src = P[3k+1] % (n-len+1), dst = P[3k+2] % (n-len+1); segments are planted
//   until their lengths add up to >= n/200; a later segment overwrites an earlier one where destinations overlap
//   letter i of segment k = base(src+i) -- the UNPLANTED text "ACGT"[draw(seed, src+i) & 3], so segments do not depend
//   on each other -- substituted when (u32)M[k*16384+i] < 0.01 * 2^32 by alt(c)[(M[..] >> 32) % 3]
// ---------------------------------------------------------------------------------------------------
static const uint64_t kRepeatSalt = 0x7265706561747321ull;

static inline uint64_t draw_host(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void k_plant_segment(uint8_t* text, uint64_t seed, uint64_t k, uint64_t src, uint64_t dst, uint32_t len,
                                uint32_t sub_thr) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    uint32_t code = (uint32_t)(draw(seed, src + i) & 3u);
    uint8_t c = "ACGT"[code];
    uint64_t x = draw(seed + kRepeatSalt + 1, k * 16384ull + i);
    if ((uint32_t)x < sub_thr) c = "CGTAGTACTACG"[code * 3 + (uint32_t)((x >> 32) % 3)];
    text[dst + i] = c;
}

// returns the number of planted letters through *planted_out (may be null)
extern "C" int slamem_synth_plant_repeats(void* text_dev, uint64_t n, uint64_t seed, void* stream, uint64_t* planted_out) {
    if (n < 20000) return -1;
    const uint32_t thr = (uint32_t)(uint64_t)(0.01 * 4294967296.0);
    uint64_t planted = 0;
    for (uint64_t k = 0; planted < n / 200; k++) {
        uint32_t len = 1000u + (uint32_t)(draw_host(seed + kRepeatSalt, 3 * k) % 9001u);
        uint64_t src = draw_host(seed + kRepeatSalt, 3 * k + 1) % (n - len + 1);
        uint64_t dst = draw_host(seed + kRepeatSalt, 3 * k + 2) % (n - len + 1);
        hipLaunchKernelGGL(k_plant_segment, dim3((len + 255) / 256), dim3(256), 0, (hipStream_t)stream, (uint8_t*)text_dev, seed,
                           k, src, dst, len, thr);
        planted += len;
    }
    if (planted_out) *planted_out = planted;
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// "Genome-like" repeat load (round 3; an extra model beside SURVEY.md 8(d)'s; same values as slamem_amd/synth.py::
// plant_genome_like): one interspersed family of n / 2480 copies of a 300 bp consensus, each 5-15 % diverged from it and
// confined to its own stretch of the text (no overlaps: the result does not depend on the order of writes); a satellite
// array of 10^4 units of 171 bp, 2 % diverged each; a block of N of min(n / 8, 30 M) letters in the middle.
// ---------------------------------------------------------------------------------------------------
static const uint64_t kGenomeSalt = 0x67656E6F6D652121ull;
constexpr uint32_t kFamilyLen = 300, kSatUnit = 171, kSatCopies = 10000;

__global__ void k_plant_family(uint8_t* text, uint64_t sf, uint64_t copies, uint64_t stride) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= copies * kFamilyLen) return;
    uint64_t k = t / kFamilyLen;
    uint32_t i = (uint32_t)(t - k * kFamilyLen);
    uint64_t dst = k * stride + draw(sf + 1, k) % (stride - kFamilyLen);
    uint64_t thr = (500ull + draw(sf + 2, k) % 1001ull) * 429497ull;
    uint32_t code = (uint32_t)(draw(sf, i) & 3u);
    uint64_t x = draw(sf + 3, k * 512ull + i);
    uint8_t c = "ACGT"[code];
    if ((x & 0xFFFFFFFFull) < thr) c = "CGTAGTACTACG"[code * 3 + (uint32_t)((x >> 32) % 3)];
    text[dst + i] = c;
}

__global__ void k_plant_satellite(uint8_t* text, uint64_t sf, uint64_t at, uint64_t total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint32_t code = (uint32_t)(draw(sf + 4, i % kSatUnit) & 3u);
    uint64_t x = draw(sf + 5, i);
    uint8_t c = "ACGT"[code];
    if ((uint32_t)x < (uint32_t)(uint64_t)(0.02 * 4294967296.0)) c = "CGTAGTACTACG"[code * 3 + (uint32_t)((x >> 32) % 3)];
    text[at + i] = c;
}

__global__ void k_plant_n(uint8_t* text, uint64_t at, uint64_t len) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) text[at + i] = 'N';
}

extern "C" int slamem_synth_plant_genome_like(void* text_dev, uint64_t n, uint64_t seed, int with_n_block, void* stream) {
    if (n < 10000000ull) return -1;
    const uint64_t copies = n / 2480, stride = n / copies, sat_at = n / 3, n_at = n / 2;
    const uint64_t n_len = n / 8 < 30000000ull ? n / 8 : 30000000ull;
    const uint64_t sf = seed + kGenomeSalt, sat = (uint64_t)kSatUnit * kSatCopies;
    hipStream_t st = (hipStream_t)stream;
    uint8_t* t = (uint8_t*)text_dev;
    hipLaunchKernelGGL(k_plant_family, dim3((unsigned)((copies * kFamilyLen + 255) / 256)), dim3(256), 0, st, t, sf, copies, stride);
    hipLaunchKernelGGL(k_plant_satellite, dim3((unsigned)((sat + 255) / 256)), dim3(256), 0, st, t, sf, sat_at, sat);
    if (with_n_block) hipLaunchKernelGGL(k_plant_n, dim3((unsigned)((n_len + 255) / 256)), dim3(256), 0, st, t, n_at, n_len);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Random 64-byte-line gather ceiling (SURVEY.md 7.2 "Roofline honesty"): every lane walks ILP independent
// chains of dependent random reads of whole 64-B blocks (4 x 16-B loads, the access shape of an FM-block
// rank query in k_find_mems).  Also the calibration workload for FETCH_SIZE on this access pattern:
// exactly lanes * iters * ILP block reads of 64 B each.
// ---------------------------------------------------------------------------------------------------
template <int ILP>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, uint64_t nblk, uint32_t iters,
                                                uint64_t* __restrict__ sink) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[ILP];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s[k] = draw(0x1234u + k, g);
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            uint64_t b = s[k] % nblk;
            const uint4* p = table + b * 4;
            uint4 a = p[0], c = p[1], d = p[2], e = p[3];
            uint64_t v = a.x ^ c.y ^ d.z ^ e.w;
            acc += v;
            s[k] = draw(s[k] + v, it);  // next index depends on the loaded data
        }
    }
    if (acc == 0x9999999999999999ull) sink[0] = acc;  // keep the loads alive
}

extern "C" int slamem_gather_bench(const void* table_dev, uint64_t nblk, uint64_t lanes, uint32_t iters, int ilp,
                                   void* sink_dev, void* stream) {
    dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (ilp == 1) hipLaunchKernelGGL(k_gather<1>, grid, block, 0, st, (const uint4*)table_dev, nblk, iters, (uint64_t*)sink_dev);
    else if (ilp == 2) hipLaunchKernelGGL(k_gather<2>, grid, block, 0, st, (const uint4*)table_dev, nblk, iters, (uint64_t*)sink_dev);
    else if (ilp == 4) hipLaunchKernelGGL(k_gather<4>, grid, block, 0, st, (const uint4*)table_dev, nblk, iters, (uint64_t*)sink_dev);
    else return -1;
    return (int)hipGetLastError();
}

// Variants that separate the address-path cost (TA: per lane access? per line?) from the fabric cost (per line):
//   mode 0: every lane reads ONE 16-B piece of a random line                (1 lane access / line)
//   mode 1: the 4 lanes of a quad read the 4 pieces of the SAME random line  (4 lane accesses / line, 1 instruction)
//   mode 2: the 16 lanes of a row read 4 consecutive lines (256 B)           (rows of 256 B)
//   mode 3: 8 lanes read 2 consecutive lines (an aligned 128-B pair)          (what a 128-byte index block would cost)
__global__ void __launch_bounds__(256) k_gather_modes(const uint4* __restrict__ table, uint64_t nblk, uint32_t iters,
                                                      int mode, uint64_t* __restrict__ sink) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lane = threadIdx.x & 63u;
    uint64_t chain = mode == 0 ? g : mode == 1 ? (g >> 2) : mode == 2 ? (g >> 4) : (g >> 3);
    uint32_t piece = mode == 0 ? (uint32_t)(g & 3u) : mode == 1 ? (lane & 3u) : mode == 2 ? (lane & 15u) : (lane & 7u);
    uint64_t s = draw(0x777u, chain);
    uint64_t acc = 0;
    uint64_t span = mode == 2 ? 4 : mode == 3 ? 2 : 1;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t b = (s % (nblk / span)) * span;
        uint4 a = table[b * 4 + piece];
        uint32_t v = a.x ^ a.y;
        // every lane of the group must follow the same chain: take the group leader's value
        uint32_t lead = mode == 0 ? v : mode == 1 ? __shfl(v, lane & ~3u) : mode == 2 ? __shfl(v, lane & ~15u) : __shfl(v, lane & ~7u);
        acc += v;
        s = draw(s + lead, it);
    }
    if (acc == 0x9999999999999999ull) sink[0] = acc;
}

// What several small reads of ONE random line cost (an in-line cascade of filter tests): every lane walks a chain of
// random lines of kLineBytes and reads kLoads dwords at hashed offsets inside each, all issued before the first is used.
template <int kLoads, int kLineBytes>
__global__ void __launch_bounds__(256) k_gather_inline(const uint32_t* __restrict__ table, uint64_t nlines, uint32_t iters,
                                                       uint64_t* __restrict__ sink) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = draw(0x999u, g);
    uint64_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t* p = table + (s % nlines) * (kLineBytes / 4);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < kLoads; k++) v ^= p[(uint32_t)(s >> (8 + 5 * k)) & (kLineBytes / 4 - 1)];
        acc += v;
        s = draw(s + v, it);
    }
    if (acc == 0x9999999999999999ull) sink[0] = acc;
}

extern "C" int slamem_gather_inline(const void* table_dev, uint64_t table_bytes, uint64_t lanes, uint32_t iters, int loads,
                                    int line_bytes, void* sink_dev, void* stream) {
    dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t* t = (const uint32_t*)table_dev;
    uint64_t* sk = (uint64_t*)sink_dev;
#define GI(L, B) hipLaunchKernelGGL((k_gather_inline<L, B>), grid, block, 0, st, t, table_bytes / B, iters, sk)
    if (loads == 1 && line_bytes == 64) GI(1, 64);
    else if (loads == 3 && line_bytes == 64) GI(3, 64);
    else if (loads == 9 && line_bytes == 64) GI(9, 64);
    else if (loads == 1 && line_bytes == 128) GI(1, 128);
    else if (loads == 3 && line_bytes == 128) GI(3, 128);
    else if (loads == 9 && line_bytes == 128) GI(9, 128);
    else return -1;
#undef GI
    return (int)hipGetLastError();
}

extern "C" int slamem_gather_modes(const void* table_dev, uint64_t nblk, uint64_t lanes, uint32_t iters, int mode,
                                   void* sink_dev, void* stream) {
    dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_gather_modes, grid, block, 0, (hipStream_t)stream, (const uint4*)table_dev, nblk, iters, mode,
                       (uint64_t*)sink_dev);
    return (int)hipGetLastError();
}
