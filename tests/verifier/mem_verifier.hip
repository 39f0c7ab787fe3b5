// mem_verifier.hip -- TEST INFRASTRUCTURE ONLY: never linked into libslamem_hip.so, never used by the product path.
//
// An index-independent check of COMPLETENESS of a MEM set at sizes no CPU oracle fits (BASELINE configs[3], [4]):
// the definition of SURVEY.md A.5 -- every (r, q, len >= l) with T[r..r+len) == Q[q..q+len), maximal on both sides --
// evaluated without any suffix array, BWT or LCP structure.  For a SAMPLE of strands, the first k = min(l, 21) letters of
// every l-letter window are entered into a hash set as exact 3-bit-per-letter keys (host side: tests/mem_verifier.py);
// the kernel below streams the WHOLE text once and reports every text position whose k-mer is in the set.  The host then
// joins the hits with the windows, keeps the left-maximal seeds and extends them to the right against the text.
// The semantics this pins are the reference's: slamem.c:139-193 prints every row of the interval and of every ancestor
// >= l deep, which is exactly the set of maximal matches (validated against brute force in C.4).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// letter code as the index orders them: A,C,G,T = 0..3, everything else ('N' after normalisation) = 4
__device__ __forceinline__ uint32_t code_of(uint8_t c) {
    c &= 0xDF;
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

__global__ void k_insert(const uint64_t *keys, uint64_t count, uint64_t *table, uint64_t mask) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t key = keys[i];
    uint64_t s = mix64(key) & mask;
    for (uint64_t step = 0; step <= mask; ++step) {  // the host sizes the table >= 8x the keys: always terminates early
        unsigned long long old = atomicCAS((unsigned long long *)&table[s], (unsigned long long)EMPTY, (unsigned long long)key);
        if (old == EMPTY || old == key) return;
        s = (s + 1) & mask;
    }
}

// One thread per SPAN consecutive text positions.  Position r is reported when r + k <= n and the key of T[r..r+k) is in
// the table.  hits[] receives at most cap positions; *nhits counts all of them (the host checks nhits <= cap).
constexpr int SPAN = 16;
__global__ void k_scan_text(const uint8_t *text, uint64_t n, int k, const uint64_t *table, uint64_t mask,
                            uint64_t *hits, uint64_t cap, unsigned long long *nhits) {
    uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * SPAN;
    if (base + k > n) return;
    const uint64_t keymask = (k * 3 >= 64) ? ~0ull : ((1ull << (3 * k)) - 1);
    uint64_t key = 0;
    for (int i = 0; i < k - 1; ++i) key = (key << 3) | code_of(text[base + i]);
    for (int j = 0; j < SPAN; ++j) {
        uint64_t r = base + j;
        if (r + k > n) break;
        key = ((key << 3) | code_of(text[r + k - 1])) & keymask;
        uint64_t s = mix64(key) & mask;
        for (uint64_t step = 0; step <= mask; ++step) {
            uint64_t v = table[s];
            if (v == EMPTY) break;
            if (v == key) {
                unsigned long long at = atomicAdd(nhits, 1ull);
                if (at < cap) hits[at] = r;
                break;
            }
            s = (s + 1) & mask;
        }
    }
}

extern "C" int memv_table_insert(const uint64_t *keys_dev, uint64_t count, uint64_t *table_dev, uint64_t slots,
                                 void *stream) {
    if (!count) return 0;
    if (slots & (slots - 1)) return -1;
    hipLaunchKernelGGL(k_insert, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys_dev, count,
                       table_dev, slots - 1);
    return (int)hipGetLastError();
}

extern "C" int memv_scan_text(const uint8_t *text_dev, uint64_t n, int k, const uint64_t *table_dev, uint64_t slots,
                              uint64_t *hits_dev, uint64_t cap, uint64_t *nhits_dev, void *stream) {
    if (k < 1 || k > 21 || (slots & (slots - 1))) return -1;
    if (n < (uint64_t)k) return 0;
    uint64_t threads = (n + SPAN - 1) / SPAN;
    uint64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return -2;
    hipLaunchKernelGGL(k_scan_text, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, text_dev, n, k, table_dev,
                       slots - 1, hits_dev, cap, (unsigned long long *)nhits_dev);
    return (int)hipGetLastError();
}
