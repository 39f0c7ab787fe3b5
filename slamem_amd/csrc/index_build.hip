// index_build.hip -- north_star (a): build the FM-index and the parent-interval structure on the GPU.
//
// Replaces, as one device pipeline (no host fallback):
//   FMI_BuildIndex          bwtindex.c:1318-1696  (GetLMSs 706, SortLMSs 787, InducedSort 1024, LF/SA samples 1451-1522)
//   BuildSampledLCPArray    lcparray.c:545-1106   (LCP samples 627-706, PSV/NSV links 782-989)
//   PackedNumberArray       packednumbers.c:18-71 (temporary BWT; not needed: the BWT goes straight into bit-planes)
//
// The reference is a sequential induced sort (pointer chasing); the arrays it produces are uniquely
// defined by the text, so this build uses what maps to the hardware instead:
//   K1  pack text to 4-bit codes + letter histogram            streaming, coalesced 16-B loads
//   K2  suffix sort by prefix doubling: LSD radix sort of 48-bit 16-mer keys, then rounds that only
//       re-sort the still-ambiguous groups by (group, rank[i+h])
//   K3  BWT -> 128-row FM blocks (ballot -> bit-planes, rank samples by block scan)
//   K4  SA kept in full (288 GB HBM: locate is one read instead of a ~31-step LF walk, bwtindex.c:402-420)
//   K5  exact LCP, Kasai order over text chunks, 16 characters per 64-bit compare
//   K7  PSV/NSV for every row through a 32-ary min hierarchy (replaces the sampled links of lcparray.c:782-989)
#include "common.h"
#include "prims.h"

#include <stdlib.h>
#include <string.h>
#include <vector>
#include <chrono>
#include <stdio.h>

namespace slamem {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ascii_code(uint32_t ch) {
    uint32_t x = ch & 0xDFu;  // upper case
    return x == 'A' ? 2u : x == 'C' ? 3u : x == 'G' ? 4u : x == 'T' ? 5u : 1u;  // everything else is N (bwtindex.c:184)
}

// 16 consecutive 4-bit codes starting at text position p (big-endian nibbles: first character on top).
__device__ __forceinline__ uint64_t window16(const uint64_t* __restrict__ pk, uint64_t p) {
    uint64_t w0 = pk[p >> 4];
    uint32_t sh = (uint32_t)(p & 15u) * 4u;
    if (sh == 0) return w0;
    uint64_t w1 = pk[(p >> 4) + 1];
    return (w0 << sh) | (w1 >> (64u - sh));
}

__device__ __forceinline__ uint32_t nibble_at(const uint64_t* __restrict__ pk, uint64_t p) {
    return (uint32_t)(pk[p >> 4] >> (60u - 4u * (uint32_t)(p & 15u))) & 15u;
}

// ------------------------------------------------------------------------------------------
// K1: pack + histogram.  One thread per 16 characters (one 16-byte load, one 8-byte store).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_text(const uint8_t* __restrict__ text, uint32_t n,
                                                   uint64_t* __restrict__ pk, uint64_t nwords,
                                                   uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[6];
    if (threadIdx.x < 6) sh[threadIdx.x] = 0;
    __syncthreads();
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    if (w < nwords) {
        uint64_t base = w * 16;
        uint64_t word = 0;
        if (base + 16 <= n && ((uintptr_t)(text + base) & 15u) == 0) {
            uint4 v = *reinterpret_cast<const uint4*>(text + base);
            uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                uint32_t code = ascii_code((q[k >> 2] >> ((k & 3) * 8)) & 0xFFu);
                word |= (uint64_t)code << (60 - 4 * k);
                c1 += code == 1; c2 += code == 2; c3 += code == 3; c4 += code == 4; c5 += code == 5;
            }
        } else {
            for (int k = 0; k < 16; k++) {
                uint64_t p = base + k;
                if (p < n) {
                    uint32_t code = ascii_code(text[p]);
                    word |= (uint64_t)code << (60 - 4 * k);
                    c1 += code == 1; c2 += code == 2; c3 += code == 3; c4 += code == 4; c5 += code == 5;
                }
            }
        }
        pk[w] = word;
    }
    if (c1) atomicAdd(&sh[1], c1);
    if (c2) atomicAdd(&sh[2], c2);
    if (c3) atomicAdd(&sh[3], c3);
    if (c4) atomicAdd(&sh[4], c4);
    if (c5) atomicAdd(&sh[5], c5);
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x < 6 && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// K2: suffix sort
// ------------------------------------------------------------------------------------------
// 48-bit key of the first 16 characters of suffix i (3 bits per character, '$'/past-the-end = 0).
__global__ void __launch_bounds__(256) k_make_keys(const uint64_t* __restrict__ pk, uint32_t rows,
                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    uint64_t x = window16(pk, i) & 0x7777777777777777ull;
    x = (x & 0x0F0F0F0F0F0F0F0Full) | ((x & 0xF0F0F0F0F0F0F0F0ull) >> 1);  // 2 chars -> 6 bits per byte
    x = (x & 0x00FF00FF00FF00FFull) | ((x & 0xFF00FF00FF00FF00ull) >> 2);  // 4 chars -> 12 bits per 16
    x = (x & 0x0000FFFF0000FFFFull) | ((x & 0xFFFF0000FFFF0000ull) >> 4);  // 8 chars -> 24 bits per 32
    x = (x & 0x00000000FFFFFFFFull) | ((x >> 32) << 24);                    // 16 chars -> 48 bits
    keys[i] = x;
    vals[i] = (uint32_t)i;
}

// group heads after a sort: head = key differs from the previous one; tmp = head ? position : 0
__global__ void __launch_bounds__(256) k_heads(const uint64_t* __restrict__ keys, uint64_t m,
                                               const uint32_t* __restrict__ pos /* nullptr: identity */,
                                               uint8_t* __restrict__ head, uint32_t* __restrict__ tmp) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    uint8_t h = (k == 0) || (keys[k] != keys[k - 1]);
    head[k] = h;
    uint32_t p = pos ? pos[k] : (uint32_t)k;
    tmp[k] = h ? p : 0u;
}

// rank[suffix] = position of its group head;  active = group has more than one member;
// optionally write the suffixes back to their SA positions.
__global__ void __launch_bounds__(256) k_assign_ranks(const uint32_t* __restrict__ sfx, const uint32_t* __restrict__ gh,
                                                      const uint8_t* __restrict__ head, uint64_t m,
                                                      const uint32_t* __restrict__ pos /* nullptr: identity */,
                                                      uint32_t* __restrict__ sa_writeback /* nullable */,
                                                      uint32_t* __restrict__ rank, uint8_t* __restrict__ active) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    uint32_t s = sfx[k];
    rank[s] = gh[k];
    if (sa_writeback) sa_writeback[pos ? pos[k] : (uint32_t)k] = s;
    active[k] = !(head[k] && (k + 1 == m || head[k + 1]));
}

// keys for one doubling round: (group head position, rank of the suffix h characters further on)
__global__ void __launch_bounds__(256) k_round_keys(const uint32_t* __restrict__ pos, uint64_t m,
                                                    const uint32_t* __restrict__ sa, const uint32_t* __restrict__ rank,
                                                    uint32_t n, uint64_t h, uint64_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    uint32_t s = sa[pos[k]];
    uint64_t t = (uint64_t)s + h;
    if (t > n) t = n;  // cannot happen for a suffix that still shares its first h characters; keeps reads in bounds
    keys[k] = ((uint64_t)rank[s] << 32) | (uint64_t)rank[t];
    vals[k] = s;
}

// ------------------------------------------------------------------------------------------
// K3: BWT -> FM blocks.  One lane per BWT row, one wave per 64-row half block.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bwt_planes(const uint32_t* __restrict__ sa, const uint64_t* __restrict__ pk,
                                                    uint32_t rows, uint32_t nblocks, FMBlock* __restrict__ fm,
                                                    uint4* __restrict__ halfcnt, uint8_t* __restrict__ is_n,
                                                    uint32_t* __restrict__ dollar_row) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t code = 0;
    bool valid = r < rows;
    if (valid) {
        uint32_t s = sa[r];
        if (s == 0) { *dollar_row = (uint32_t)r; code = 0; }
        else code = nibble_at(pk, (uint64_t)s - 1);  // BWT[r] = T[SA[r]-1]   (bwtindex.c:1092,1204)
        is_n[r] = (code == 1);
    }
    bool ex = !valid || code < 2;
    uint32_t c2 = code - 2u;
    unsigned long long b0 = __ballot(!ex && (c2 & 1u));
    unsigned long long b1 = __ballot(!ex && (c2 & 2u));
    unsigned long long be = __ballot(ex);
    uint64_t half = r >> 6;  // 64-row half block index
    if ((threadIdx.x & 63u) == 0 && (half >> 1) < nblocks) {
        FMBlock* b = &fm[half >> 1];
        uint32_t hsel = (uint32_t)(half & 1u);
        b->p0[hsel] = b0;
        b->p1[hsel] = b1;
        b->ex[hsel] = be;
        unsigned long long ne = ~be;
        uint4 c;
        c.x = __popcll(~b0 & ~b1 & ne);  // A
        c.y = __popcll(b0 & ~b1 & ne);   // C
        c.z = __popcll(~b0 & b1 & ne);   // G
        c.w = __popcll(b0 & b1 & ne);    // T
        halfcnt[half] = c;
    }
}

__global__ void __launch_bounds__(256) k_block_counts(const uint4* __restrict__ halfcnt, uint32_t nblocks,
                                                      uint4* __restrict__ blkcnt) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    uint4 a = halfcnt[2 * (uint64_t)b], c = halfcnt[2 * (uint64_t)b + 1];
    blkcnt[b] = make_uint4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
}

// cnt[c-2] = C[c] + occ(c, rows before the block)      (reference: letterJumpsSample, bwtindex.c:1454,1481)
__global__ void __launch_bounds__(256) k_rank_samples(const uint4* __restrict__ pre, uint32_t nblocks, uint4 C,
                                                      FMBlock* __restrict__ fm) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    uint4 p = pre[b];
    fm[b].cnt[0] = C.x + p.x;
    fm[b].cnt[1] = C.y + p.y;
    fm[b].cnt[2] = C.z + p.z;
    fm[b].cnt[3] = C.w + p.w;
}

// ------------------------------------------------------------------------------------------
// K1b: k-mer presence filter (not in the reference: lets the search skip strands that share no k-mer with the
// text, e.g. the wrong strand of every read).  One lane per text position p: the (k-2)-mer that starts there picks
// the line, and the lane enters it, the k-mers that start at p-2..p and the (k+2)-mers that start at p-4..p -- all the
// k- and (k+2)-mers of the text that contain it.  Windows containing N are not entered (a query window with N is
// treated as present, so N == N matches are never filtered out).
// ------------------------------------------------------------------------------------------
// 16 packed letters (4-bit ids, first on top) -> their 2-bit values (id - 2; first letter in the highest bits of the 32) and a
// bit per letter that is one of A,C,G,T (first letter = bit 15)
__device__ __forceinline__ void letters16(uint64_t x, uint32_t& v2, uint32_t& ok16) {
    const uint64_t k1 = 0x1111111111111111ull;
    uint64_t okn = ((x >> 1) | (x >> 2) | (x >> 3)) & k1;            // nibble >= 2
    uint64_t c = (x - 2ull * okn) & (3ull * okn);                     // id - 2 where ok, 0 elsewhere (no borrow: ok nibbles are >= 2)
    // gather the low 2 bits of every nibble: 64 -> 32 bits
    c = (c & 0x0303030303030303ull) | ((c & 0x3030303030303030ull) >> 2);
    c = (c & 0x000F000F000F000Full) | ((c & 0x0F000F000F000F00ull) >> 4);
    c = (c & 0x000000FF000000FFull) | ((c & 0x00FF000000FF0000ull) >> 8);
    c = (c & 0xFFFFull) | ((c >> 16) & 0xFFFF0000ull);
    v2 = (uint32_t)c;
    // gather bit 0 of every nibble: 64 -> 16 bits
    uint64_t o = okn;
    o = (o & 0x0101010101010101ull) | ((o & 0x1010101010101010ull) >> 3);
    o = (o & 0x0003000300030003ull) | ((o & 0x0300030003000300ull) >> 6);
    o = (o & 0x0000000F0000000Full) | ((o & 0x000F0000000F0000ull) >> 12);
    o = (o & 0xFFull) | ((o >> 24) & 0xFF00ull);
    ok16 = (uint32_t)o;
}

// The entries of text position p: hash of the (k-2)-mer that starts there (-> the line) and the bits of every entry in the
// line's eight words.  false: the (k-2)-mer does not lie in the text or holds an N (nothing is entered).
__device__ __forceinline__ bool kfilter_entries(const uint64_t* __restrict__ pk, uint32_t n, uint32_t k, uint32_t levels,
                                                uint64_t p, uint64_t& h0, unsigned long long (&words)[8]) {
    const uint32_t k1 = k - 2u;
    if (p + k1 > n) return false;
    // the letters [p-4, p+k1+4) as 2-bit values, first letter in the highest bits; ok: bit per letter, set when the
    // letter exists and is not N
    const uint32_t wn = k1 + 8u;  // <= 32 (k <= 26)
    uint64_t w = 0, ok = 0;
    if (p >= 4u) {
        // 32 letters from p-4 on in two packed words (behind the text the packed copy holds zeros: '$' and padding, not ok)
        uint32_t va, oa, vb, ob;
        letters16(window16(pk, p - 4u), va, oa);
        letters16(window16(pk, p + 12u), vb, ob);
        const uint64_t v64 = ((uint64_t)va << 32) | vb, o32 = ((uint64_t)oa << 16) | ob;
        w = v64 >> (2u * (32u - wn));
        ok = o32 >> (32u - wn);
    } else {
        for (uint32_t d = 0; d < wn; d++) {
            const int64_t t = (int64_t)p - 4 + (int64_t)d;
            uint32_t c = (t >= 0 && t < (int64_t)n) ? nibble_at(pk, (uint64_t)t) : 0u;
            w = (w << 2) | (uint64_t)(c >= 2u ? c - 2u : 0u);
            ok = (ok << 1) | (uint64_t)(c >= 2u);
        }
    }
    auto piece = [&](uint32_t first, uint32_t len, uint64_t& v) {  // letters [first, first+len) of the window
        const uint32_t sh = wn - first - len;
        const uint64_t all = (1ull << len) - 1ull;
        v = (w >> (2u * sh)) & (len >= 32u ? ~0ull : (1ull << (2u * len)) - 1ull);
        return ((ok >> sh) & all) == all;
    };
    // the nine entries: hash and whether the window lies in the text and holds no N
    uint64_t v, hs[9];
    bool in[9];
    if (!piece(4u, k1, v)) return false;
    hs[0] = kfilter_hash(v ^ kFilterShortSalt);
    in[0] = true;
    h0 = hs[0];
#pragma unroll
    for (uint32_t o = 2u; o <= 4u; o++) {
        in[o - 1u] = piece(o, k, v);
        hs[o - 1u] = kfilter_hash(v);
    }
    const bool third = levels >= 3u && k + 2u <= 32u;  // third level (it fits the 64-bit rolling value of the search)
#pragma unroll
    for (uint32_t o = 0u; o <= 4u; o++) {
        in[4u + o] = third && piece(o, k + 2u, v);
        hs[4u + o] = kfilter_hash(v ^ kFilterLongSalt);
    }
#pragma unroll
    for (uint32_t wd = 0; wd < 8u; wd++) {
        unsigned long long m = 0ull;
#pragma unroll
        for (uint32_t e = 0; e < 9u; e++)
            if (in[e] && kfilter_word(hs[e]) == wd) m |= (unsigned long long)kfilter_bits(hs[e]);
        words[wd] = m;
    }
    return true;
}

// The direct form: one atomic per word of the line that gets bits (5.5 on average per position, 550 M random atomics at
// 100 Mbp: 21.8 ms, 40 % of round 2's build).  Kept as the checker of the sorted form below (SLAMEM_KFILTER_ATOMIC=1; the
// filter's content does not depend on the order of the entries) and for texts too short to be worth a sort.
__global__ void __launch_bounds__(256) k_kfilter_build(const uint64_t* __restrict__ pk, uint32_t n, uint32_t k,
                                                       uint32_t log2_words, uint32_t levels,
                                                       unsigned long long* __restrict__ filter) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t h0;
    unsigned long long words[8];
    if (!kfilter_entries(pk, n, k, levels, p, h0, words)) return;
    unsigned long long* line = filter + kfilter_line(h0, log2_words);
#pragma unroll
    for (uint32_t wd = 0; wd < 8u; wd++)
        if (words[wd]) atomicOr(&line[wd], words[wd]);
}

// The sorted form.  A line of the filter is written ONCE, whole, by the lane that owns it: (1) every text position gets the
// key "line of its (k-2)-mer" (positions without one sort behind all lines), (2) the positions are sorted by that key on the
// hand-written radix sort -- sequential passes over 12 bytes per position instead of 5.5 random read-modify-writes --,
// (3) the first position of every run of equal keys gathers the entries of its whole run (three positions on average) in
// registers and stores the line's 64 bytes.  The region is zeroed beforehand for the lines that get nothing.
__global__ void __launch_bounds__(256) k_kfilter_keys(const uint64_t* __restrict__ pk, uint32_t n, uint32_t k,
                                                      uint32_t log2_words, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (uint64_t)n) return;
    const uint32_t k1 = k - 2u;
    uint64_t key = 1ull << (log2_words - 3u);  // behind every line
    if (p + k1 <= n) {  // (k1 <= 24: two packed words hold the letters)
        uint32_t va, oa, vb, ob;
        letters16(window16(pk, p), va, oa);
        letters16(window16(pk, p + 16u), vb, ob);
        const uint64_t v64 = ((uint64_t)va << 32) | vb, o32 = ((uint64_t)oa << 16) | ob;
        const uint64_t v = v64 >> (2u * (32u - k1));
        const uint64_t all = (1ull << k1) - 1ull;
        if (((o32 >> (32u - k1)) & all) == all) key = kfilter_line(kfilter_hash(v ^ kFilterShortSalt), log2_words) >> 3;
    }
    keys[p] = key;
    vals[p] = (uint32_t)p;
}

// Every lane computes the entries of ITS position (all lanes busy: the hashes are most of the work), the block's lanes put
// them into LDS, and the first lane of every run of equal keys inside the block ORs its run together and writes the line:
// with one 64-byte store when the whole run lies inside the block (nearly all: three positions per line on average), with
// atomics when the run crosses a block boundary (the other part writes the same line).  A satellite array or a homopolymer --
// one (k-2)-mer thousands of times -- is many block-sized pieces of one run, all but the entries of 256 positions combined
// before they reach memory.
__global__ void __launch_bounds__(256) k_kfilter_fill(const uint64_t* __restrict__ pk, uint32_t n, uint32_t k,
                                                      uint32_t log2_words, uint32_t levels,
                                                      const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                      unsigned long long* __restrict__ filter) {
    __shared__ unsigned long long sh_w[256][9];  // (9: the rows start in different banks)
    __shared__ uint64_t sh_key[256 + 2];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = threadIdx.x;
    const uint64_t none = 1ull << (log2_words - 3u);
    const uint64_t key = i < (uint64_t)n ? keys[i] : none;
    unsigned long long words[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    if (key != none) {
        uint64_t h0;
        if (!kfilter_entries(pk, n, k, levels, (uint64_t)vals[i], h0, words)) {
#pragma unroll
            for (uint32_t wd = 0; wd < 8u; wd++) words[wd] = 0ull;
        }
    }
#pragma unroll
    for (uint32_t wd = 0; wd < 8u; wd++) sh_w[t][wd] = words[wd];
    sh_key[t + 1] = key;
    if (t == 0) {  // the keys on both sides of the block: does a run go on there?
        const uint64_t b0 = (uint64_t)blockIdx.x * blockDim.x;
        sh_key[0] = b0 > 0 ? keys[b0 - 1] : ~0ull;
        sh_key[257] = b0 + 256 < (uint64_t)n ? keys[b0 + 256] : ~0ull;
    }
    __syncthreads();
    if (key == none || (t != 0 && sh_key[t] == key)) return;  // nothing here / not the first of its run inside the block
    bool whole = sh_key[t] != key;  // (t == 0: the run may have started in the block before)
    unsigned long long acc[8] = {words[0], words[1], words[2], words[3], words[4], words[5], words[6], words[7]};
    uint32_t j = t + 1;
    for (; j < 256u && sh_key[j + 1] == key; j++) {
#pragma unroll
        for (uint32_t wd = 0; wd < 8u; wd++) acc[wd] |= sh_w[j][wd];
    }
    if (j == 256u && sh_key[257] == key) whole = false;  // it goes on in the next block
    unsigned long long* line = filter + (key << 3);
    if (!whole) {
#pragma unroll
        for (uint32_t wd = 0; wd < 8u; wd++)
            if (acc[wd]) atomicOr(&line[wd], acc[wd]);
        return;
    }
    uint4* l4 = reinterpret_cast<uint4*>(line);
    l4[0] = make_uint4((uint32_t)acc[0], (uint32_t)(acc[0] >> 32), (uint32_t)acc[1], (uint32_t)(acc[1] >> 32));
    l4[1] = make_uint4((uint32_t)acc[2], (uint32_t)(acc[2] >> 32), (uint32_t)acc[3], (uint32_t)(acc[3] >> 32));
    l4[2] = make_uint4((uint32_t)acc[4], (uint32_t)(acc[4] >> 32), (uint32_t)acc[5], (uint32_t)(acc[5] >> 32));
    l4[3] = make_uint4((uint32_t)acc[6], (uint32_t)(acc[6] >> 32), (uint32_t)acc[7], (uint32_t)(acc[7] >> 32));
}

// ------------------------------------------------------------------------------------------
// K1d: text bit-planes and the seed table (the seed-and-compare path of the search, mem_search.hip k_seed_mems)
// ------------------------------------------------------------------------------------------
// even bits of x (bit 2j -> bit j)
__device__ __forceinline__ uint32_t even_bits16(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
// one lane per unit of 64 letters (the occurs-once word is filled in by the seed build)
__global__ void __launch_bounds__(256) k_text_planes(const uint64_t* __restrict__ pk, uint32_t n, uint64_t units,
                                                     TextPlanes* __restrict__ tpl) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t p0 = 0, p1 = 0, nm = 0;
    if (u < units && u * 64 < (uint64_t)n) {
#pragma unroll
        for (uint32_t w = 0; w < 4u; w++) {
            uint32_t v2, ok16;
            letters16(pk[u * 4 + w], v2, ok16);  // first letter in the highest bits / bit 15
            const uint32_t r = __brev(v2);       // letter j at bits 2j (its high bit) and 2j+1 (its low bit)
            p1 |= (uint64_t)even_bits16(r) << (16u * w);
            p0 |= (uint64_t)even_bits16(r >> 1) << (16u * w);
            nm |= (uint64_t)((__brev(ok16) >> 16) ^ 0xFFFFu) << (16u * w);
        }
        const uint64_t left = (uint64_t)n - u * 64;  // letters of the text in this unit
        if (left < 64) nm &= (1ull << left) - 1ull;  // positions behind the text are not marked (the compare bounds them)
    }
    if (u < units) tpl[u] = TextPlanes{p0, p1, nm, 0ull};
}

// k letters of the text from position t as plane fields; false: a letter is not A,C,G,T
__device__ __forceinline__ bool text_field(const TextPlanes* __restrict__ tpl, uint64_t t,
                                           uint32_t k, uint32_t& f0, uint32_t& f1) {
    const uint64_t u = t >> 6;
    const uint32_t sh = (uint32_t)t & 63u;
    const TextPlanes a = tpl[u], b = tpl[u + 1];
    const uint64_t na = a.nm, nb = b.nm;
    const uint32_t m = (k >= 32u) ? 0xFFFFFFFFu : (1u << k) - 1u;
    f0 = (uint32_t)((a.p0 >> sh) | ((b.p0 << 1) << (63u - sh))) & m;
    f1 = (uint32_t)((a.p1 >> sh) | ((b.p1 << 1) << (63u - sh))) & m;
    return ((uint32_t)((na >> sh) | ((nb << 1) << (63u - sh))) & m) == 0u;
}
// sort key of text position t: its bucket above a byte (the tag's hash bits above the orientation bit, so that the two
// orientations of a k-mer sort next to each other); positions without a k-mer over A,C,G,T sort behind every bucket.
// Also the first form of the "occurs once" plane: a bit for every position that has a k-mer which is not its own reverse
// complement (k_seed_fill clears those that occur more than once); a wave covers one 64-letter unit
__global__ void __launch_bounds__(256) k_seed_keys(TextPlanes* __restrict__ tpl,
                                                   uint32_t n, uint32_t k, uint32_t log2b, uint64_t* __restrict__ keys,
                                                   uint32_t* __restrict__ vals, uint64_t units) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t key = 1ull << (log2b + 8u);
    uint32_t f0, f1;
    bool once = false;
    if (t < (uint64_t)n && t + k <= (uint64_t)n && text_field(tpl, t, k, f0, f1)) {
        uint32_t bucket, tagbits, orient, pal;
        seed_place(f0, f1, k, log2b, bucket, tagbits, orient, pal);
        key = ((uint64_t)bucket << 8) | (uint64_t)((tagbits << 1) | orient);
        once = pal == 0u;
    }
    const unsigned long long m = __ballot(once);
    if ((threadIdx.x & 63u) == 0u && (t >> 6) < units) tpl[t >> 6].uq = m;  // (every lane of the grid has read its planes: a lane reads units t >> 6 and the next, the word written is neither's p0 / p1 / nm)
    if (t >= (uint64_t)n) return;
    keys[t] = key;
    vals[t] = (uint32_t)t;
}
// tag byte of a bucket entry (hash bits, bit 7: orientation) from the low byte of its sort key (hash bits above the orientation bit)
__device__ __forceinline__ uint8_t seed_tag_of_sorted(uint32_t key) { return (uint8_t)(((key >> 1) & 0x7Fu) | ((key & 1u) << 7)); }
// the sorted positions into their buckets: a lane finds its place in its bucket's run by looking back, the first lane of a
// run also counts it (13 = more than fit) and says how many entries of the spill list the run asks for (13 to 28 k-mers:
// those beyond the twelfth, rounded up to four)
__global__ void __launch_bounds__(256) k_seed_fill(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                   uint64_t n, uint32_t log2b, SeedBucket* __restrict__ table,
                                                   uint32_t* __restrict__ want, TextPlanes* __restrict__ tpl) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    uint32_t asks = 0;
    if ((key >> (log2b + 8u)) == 0ull) {
        const uint64_t b = key >> 8;
        // the same k-mer (in either orientation) next to this one in the sorted order: it occurs more than once
        if ((i > 0 && (keys[i - 1u] >> 1) == (key >> 1)) || (i + 1u < n && (keys[i + 1u] >> 1) == (key >> 1)))
            atomicAnd(reinterpret_cast<unsigned long long*>(&tpl[vals[i] >> 6].uq), ~(1ull << (vals[i] & 63u)));
        uint32_t r = 0;
        while (r < kSeedSlots && r < i && (keys[i - r - 1u] >> 8) == b) r++;
        if (r < kSeedSlots) {
            table[b].pos[r] = vals[i];
            table[b].tag[r] = seed_tag_of_sorted((uint32_t)key);
        }
        if (r == 0u) {
            uint32_t c = 1;
            while (c <= kSeedSlots + kSeedSpillMax && i + c < n && (keys[i + c] >> 8) == b) c++;
            table[b].count = c > kSeedSlots ? kSeedSlots + 1u : c;
            if (c > kSeedSlots && c <= kSeedSlots + kSeedSpillMax) asks = (c - kSeedSlots + 3u) & ~3u;
        }
    }
    want[i] = asks;
}
// the runs that asked, at their places of the spill list (the exclusive sums of what was asked for, so the list does not
// depend on the order the lanes run in); a run that does not fit the list any more stays "more than fit"
__global__ void __launch_bounds__(256) k_seed_spill(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                    const uint32_t* __restrict__ want, const uint32_t* __restrict__ place,
                                                    uint64_t n, SeedBucket* __restrict__ table, uint64_t* __restrict__ spill,
                                                    uint32_t cap, uint32_t* __restrict__ used) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t asks = want[i];
    if (asks == 0u) return;
    const uint32_t at = place[i];
    if ((uint64_t)at + asks > (uint64_t)cap) return;
    const uint64_t b = keys[i] >> 8;
    uint32_t c = kSeedSlots;
    while (c < kSeedSlots + kSeedSpillMax && i + c < n && (keys[i + c] >> 8) == b) c++;
    for (uint32_t e = 0; e < asks; e++) {
        const uint64_t j = i + kSeedSlots + e;
        spill[at + e] = kSeedSlots + e < c ? ((uint64_t)vals[j] | ((uint64_t)seed_tag_of_sorted((uint32_t)keys[j]) << 32)) : ~0ull;
    }
    table[b].count = kSeedSpilled | ((c - kSeedSlots) << 24) | (at >> 2);
    atomicMax(used, at + asks);
}

// ------------------------------------------------------------------------------------------
// K5: exact LCP in text order (Kasai): one thread per chunk of text positions, the match length
// carried from position i to i+1 never drops by more than one.  Stores LCP+1 (0 = "-1" sentinel).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kLcpChunk = 32;

// A lane of k_lcp_kasai starts its 32 positions without the carry of the position before them, so a repeat of
// length R (a centromere's run of N, a tandem array, a long duplication) would cost every lane inside it up to R/16
// steps: R^2/1024 in total.  Two sampled passes bound that: text positions that are multiples of kLcpCoarse2 are
// compared from scratch (one wave each, 4096 letters per step), multiples of kLcpCoarse1 start from the sample
// before them, and the lanes of k_lcp_kasai start from those -- all through  LCP(i+d) >= LCP(i) - d  (the Kasai
// invariant).  And when the predecessor in suffix order moves along with the position (pred(i+d) = pred(i)+d, which is
// what happens inside every long repeat) the value is EXACT without reading the text: LCP(i+d) = LCP(i) - d.  That
// matters beyond the saved work: inside a repeat every position's comparison ends at the same text position, and
// millions of loads of one cache line crawl (measured 23 ns each: 190 ms for an 8 Mbp run of N).
constexpr uint32_t kLcpCoarse1 = 1024, kLcpCoarse2 = 32768;
// The whole wave extends one comparison: suffixes at ii and jj agree on hh letters; returns their exact LCP.
// 4096 letters per step (every lane 64 of them); reads up to 4096 letters behind the mismatch (kPackSlack).
__device__ __forceinline__ uint32_t wave_extend_lcp(const uint64_t* __restrict__ pk, uint64_t ii, uint64_t jj, uint32_t hh,
                                                    uint32_t lane) {
    for (;;) {
        const uint64_t a = ii + hh + lane * 64u, b = jj + hh + lane * 64u;
        uint64_t x0 = window16(pk, a) ^ window16(pk, b);
        uint64_t x1 = window16(pk, a + 16) ^ window16(pk, b + 16);
        uint64_t x2 = window16(pk, a + 32) ^ window16(pk, b + 32);
        uint64_t x3 = window16(pk, a + 48) ^ window16(pk, b + 48);
        uint32_t mine = x0 ? (uint32_t)__clzll((long long)x0) >> 2
                      : x1 ? 16u + ((uint32_t)__clzll((long long)x1) >> 2)
                      : x2 ? 32u + ((uint32_t)__clzll((long long)x2) >> 2)
                      : x3 ? 48u + ((uint32_t)__clzll((long long)x3) >> 2) : 64u;
        uint64_t m = __ballot(mine != 64u);
        if (m) {
            const int first = __ffsll((unsigned long long)m) - 1;
            return hh + (uint32_t)first * 64u + __shfl(mine, first);
        }
        hh += 4096u;
    }
}

// one WAVE per sampled position; out_h[k] = LCP of the suffix at k*stride with its predecessor, out_j[k] = where that
// predecessor starts
__global__ void __launch_bounds__(256) k_lcp_sampled(const uint64_t* __restrict__ pk, const uint32_t* __restrict__ sa,
                                                     const uint32_t* __restrict__ rank, uint32_t rows, uint32_t stride,
                                                     const uint32_t* __restrict__ prev_h, const uint32_t* __restrict__ prev_j,
                                                     uint32_t prev_stride, uint32_t* __restrict__ out_h,
                                                     uint32_t* __restrict__ out_j) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t i = k * stride;
    if (i >= rows) return;  // whole wave
    const uint32_t r = rank[i];
    if (r == 0u) { if (lane == 0u) { out_h[k] = 0u; out_j[k] = 0xFFFFFFFFu; } return; }
    const uint64_t j = sa[r - 1];
    uint32_t h = 0;
    bool exact = false;
    if (prev_h) {
        uint32_t p = prev_h[i / prev_stride], d = (uint32_t)(i % prev_stride);
        if (p > d) { h = p - d; exact = (uint64_t)prev_j[i / prev_stride] + d == j; }
    }
    if (!exact) h = wave_extend_lcp(pk, i, j, h, lane);
    if (lane == 0u) { out_h[k] = h; out_j[k] = (uint32_t)j; }
}

__global__ void __launch_bounds__(256) k_lcp_kasai(const uint64_t* __restrict__ pk, const uint32_t* __restrict__ sa,
                                                   const uint32_t* __restrict__ rank, uint32_t rows,
                                                   const uint32_t* __restrict__ coarse_h, const uint32_t* __restrict__ coarse_j,
                                                   uint32_t* __restrict__ l32, uint32_t* __restrict__ max_lcp) {
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i = t * kLcpChunk;
    uint64_t i1 = i + kLcpChunk;
    if (i1 > rows) i1 = rows;
    if (i > rows) i = i1 = rows;  // lanes behind the text idle, but stay for the wave-wide steps
    uint32_t h = 0, mx = 0, r = 0;
    // (h, pj): a lower bound of the next position's value, and -- if pj is not ~0 -- the predecessor start for which the
    // bound is the exact value
    uint64_t pj = ~0ull, j = 0;
    if (i < i1) {
        uint32_t p = coarse_h[i / kLcpCoarse1], d = (uint32_t)(i % kLcpCoarse1);
        if (p > d) { h = p - d; pj = (uint64_t)coarse_j[i / kLcpCoarse1] + d; }
    }
    for (;;) {
        // ---- every lane on its own: positions whose comparison ends within 256 letters ---------------------------
        bool stuck = false;
        while (i < i1) {
            r = rank[i];
            if (r == 0) { h = 0; pj = ~0ull; i++; continue; }  // the '$' suffix: row 0 keeps the sentinel
            j = sa[r - 1];
            if (j != pj) {
                // extend: compare 16 characters per step; the unique '$' guarantees a mismatch before either suffix ends
                for (uint32_t it = 0;; it++) {
                    uint64_t x = window16(pk, i + h) ^ window16(pk, j + h);
                    if (x) { h += (uint32_t)__clzll((long long)x) >> 2; break; }
                    h += 16;
                    if (it == 15u) { stuck = true; break; }
                }
                if (stuck) break;
            }
            l32[r] = h + 1;
            mx = h > mx ? h : mx;
            if (h) { h--; pj = j + 1; } else pj = ~0ull;
            i++;
        }
        // ---- long comparisons (the first positions of a long repeat that no sample covers): the wave together,
        //      4096 letters per step; a later one that runs parallel to an earlier one follows from it
        uint64_t todo = __ballot(stuck);
        if (!todo) break;
        uint64_t last_i = 0, last_j = 0;
        uint32_t last_h = 0;
        bool have_last = false;
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint64_t ii = __shfl((unsigned long long)i, src), jj = __shfl((unsigned long long)j, src);
            uint32_t hh = __shfl(h, src);
            if (have_last && ii > last_i && ii - last_i < last_h && jj == last_j + (ii - last_i)) {
                hh = last_h - (uint32_t)(ii - last_i);
            } else {
                hh = wave_extend_lcp(pk, ii, jj, hh, lane);
            }
            last_i = ii; last_j = jj; last_h = hh; have_last = true;
            if ((int)lane == src) h = hh;
        }
        if (stuck) {  // finish the position that was stuck, then go on
            l32[r] = h + 1;
            mx = h > mx ? h : mx;
            if (h) { h--; pj = j + 1; } else pj = ~0ull;
            i++;
        }
    }
    if (mx) atomicMax(max_lcp, mx);
}

__global__ void k_lcp_sentinels(uint32_t rows, uint32_t* l32, uint32_t* psv, uint32_t* nsv) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        l32[0] = 0; l32[rows] = 0;  // LCP[0] = LCP[n+1] = -1   (lcparray.c:624,667)
        psv[0] = 0; nsv[0] = rows; psv[rows] = 0; nsv[rows] = rows;
    }
}

// K-mer jump table (no reference counterpart): the BWT interval of every K-mer over A,C,G,T, so that a scan takes its
// first K backward steps -- FMI_FollowLetter from the root, slamem.c:110-121, the steps with the widest intervals: two
// FM lines each -- in ONE read.  Suffixes that share their first K letters are contiguous in suffix order.
// pass 1: key of every row (2 bits per letter, first letter on top; ~0 = shorter than K or holds N / '$')
__global__ void __launch_bounds__(256) k_kjump_keys(const uint32_t* __restrict__ sa, const uint64_t* __restrict__ pk,
                                                    uint32_t n, uint32_t K, uint32_t* __restrict__ keys) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    uint32_t s = sa[r];
    uint32_t key = 0xFFFFFFFFu;
    if ((uint64_t)s + K <= n) {
        uint64_t x = window16(pk, s);
        uint64_t v = ((x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ull;  // nibble >= 2: one of A,C,G,T
        const uint64_t topk = ~0ull << (4u * (16u - K));
        if ((v & topk) == (0x1111111111111111ull & topk)) {
            x = (x & topk) | (0x2222222222222222ull & ~topk);  // (no borrow may run into the K letters)
            x = (x - 0x2222222222222222ull) & topk;
            x = (x & 0x0303030303030303ull) | ((x & 0x3030303030303030ull) >> 2);
            x = (x & 0x000F000F000F000Full) | ((x & 0x0F000F000F000F00ull) >> 4);
            x = (x & 0x000000FF000000FFull) | ((x & 0x00FF000000FF0000ull) >> 8);
            x = (x & 0xFFFFull) | ((x >> 16) & 0xFFFF0000ull);
            key = (uint32_t)x >> (2u * (16u - K));
        }
    }
    keys[r] = key;
}
// pass 2: the first and the last row of every key
__global__ void __launch_bounds__(256) k_kjump_bounds(const uint32_t* __restrict__ keys, uint32_t n, uint2* __restrict__ table) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    uint32_t k = keys[r];
    if (k == 0xFFFFFFFFu) return;
    if (r == 0 || keys[r - 1] != k) table[k].x = (uint32_t)r;
    if (r == n || keys[r + 1] != k) table[k].y = (uint32_t)r;
}
__global__ void __launch_bounds__(256) k_kjump_init(uint2* __restrict__ table, uint64_t entries) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < entries) table[i] = make_uint2(1u, 0u);  // top > bottom: the K-mer does not occur
}

// k-mer occurrence bitmap (no reference counterpart): bit x of the bitmap = the k-mer with value x (2 bits per letter,
// first letter on top) occurs in the text.  Directly addressed -- no hash -- so that K8 can test a handful of windows
// around a substituted letter in one trip (mem_search.hip, states SKV / SKQ / SKP).
__global__ void __launch_bounds__(256) k_kbits_build(const uint64_t* __restrict__ pk, uint32_t n, uint32_t k,
                                                     unsigned long long* __restrict__ bits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + k > n) return;
    uint64_t x = window16(pk, i);
    const uint64_t v = ((x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ull;  // nibble >= 2: one of A,C,G,T
    const uint64_t topk = ~0ull << (4u * (16u - k));
    if ((v & topk) != (0x1111111111111111ull & topk)) return;
    x = (x & topk) | (0x2222222222222222ull & ~topk);
    x = (x - 0x2222222222222222ull) & topk;
    x = (x & 0x0303030303030303ull) | ((x & 0x3030303030303030ull) >> 2);
    x = (x & 0x000F000F000F000Full) | ((x & 0x0F000F000F000F00ull) >> 4);
    x = (x & 0x000000FF000000FFull) | ((x & 0x00FF000000FF0000ull) >> 8);
    x = (x & 0xFFFFull) | ((x >> 16) & 0xFFFF0000ull);
    const uint32_t key = (uint32_t)x >> (2u * (16u - k));
    atomicOr(&bits[key >> 6], 1ull << (key & 63u));
}

// text-ordered records (TextRec) and the parent-depth class of every text position (one byte each, packed into the
// groups by k_text_groups): one random 16-byte read of the row's record per position
__global__ void __launch_bounds__(256) k_text_records(const uint32_t* __restrict__ isa, const RowRec* __restrict__ rec,
                                                      uint32_t rows, TextRec* __restrict__ out, uint8_t* __restrict__ cls) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= rows) return;
    uint32_t row = isa[s];
    RowRec r = rec[row];
    uint32_t d1 = r.lcp1 > r.lcp1n ? r.lcp1 : r.lcp1n;  // depth + 1 of the parent of [row,row]  (lcparray.c:514-518)
    TextRec t;
    t.row = row;
    t.ptop = r.lcp1 == d1 ? r.psv : row;
    t.pbot = r.lcp1n == d1 ? r.nsvn - 1u : row;
    t.pdepth1 = d1;
    out[s] = t;
    cls[s] = (uint8_t)depth_class((int)d1 - 1);
}

// TextGroup g = the packed letters of positions 16g..16g+15 (K1's word) + the classes of positions 16g+1..16g+16
__global__ void __launch_bounds__(256) k_text_groups(const uint64_t* __restrict__ pk, const uint8_t* __restrict__ cls,
                                                     uint32_t rows, uint64_t ngroups, TextGroup* __restrict__ out) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    uint64_t c = 0;
    for (uint32_t i = 0; i < 16; i++) {
        uint64_t x = g * 16 + i + 1;  // the position to the right of letter 16g+i
        uint64_t v = x < rows ? (uint64_t)cls[x] : 15ull;  // past the text: "stop"
        c |= v << (60u - 4u * i);
    }
    TextGroup t;
    t.letters = pk[g];
    t.classes = c;
    out[g] = t;
}

// the 16-byte row records: record i describes the boundaries of row i: {LCP[i]+1, PSV[i], LCP[i+1]+1, NSV[i+1]}
// (and, beside the records: how many rows have an LCP of at least kLcpGe[i] -- what tells a launch of the search how repeat-rich
//  the text is at its minimum length, IndexView::lcp_ge / ArenaHeader::lcp_ge)
__global__ void __launch_bounds__(256) k_pack_records(const uint32_t* __restrict__ l32, const uint32_t* __restrict__ psv,
                                                      const uint32_t* __restrict__ nsv, uint32_t rows,
                                                      RowRec* __restrict__ rec, uint32_t* __restrict__ lcp_ge) {
    __shared__ uint32_t sh[10];
    if (threadIdx.x < 10) sh[threadIdx.x] = 0;
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= rows) {
        RowRec r;
        r.lcp1 = l32[i];
        r.psv = psv[i];
        r.lcp1n = i < rows ? l32[i + 1] : 0u;
        r.nsvn = i < rows ? nsv[i + 1] : rows;
        rec[i] = r;
        if (r.lcp1 > kLcpGe[0]) {  // LCP >= 18: rare on a text without repeats (chance matches end near log4 n)
#pragma unroll
            for (int b = 0; b < 10; b++)
                if (r.lcp1 > kLcpGe[b]) atomicAdd(&sh[b], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 10 && sh[threadIdx.x]) atomicAdd(&lcp_ge[threadIdx.x], sh[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// K7: PSV / NSV of every row through a 32-ary hierarchy of minima.
// ------------------------------------------------------------------------------------------
constexpr int kMaxLevels = 8;
struct MinLevels {
    const uint32_t* lv[kMaxLevels];  // lv[0] = the values themselves
    uint64_t size[kMaxLevels];
    int count;
};

__global__ void __launch_bounds__(256) k_min_level(const uint32_t* __restrict__ in, uint64_t in_size,
                                                   uint32_t* __restrict__ out, uint64_t out_size) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= out_size) return;
    uint64_t lo = b * 32, hi = lo + 32;
    if (hi > in_size) hi = in_size;
    uint32_t m = 0xFFFFFFFFu;
    for (uint64_t j = lo; j < hi; j++) { uint32_t v = in[j]; m = v < m ? v : m; }
    out[b] = m;
}

__device__ __forceinline__ uint32_t psv_search(const MinLevels& L, uint64_t i, uint32_t v) {
    const uint32_t* a = L.lv[0];
    uint64_t lo = i & ~31ull;
    for (uint64_t j = i; j-- > lo;)
        if (a[j] < v) return (uint32_t)j;
    uint64_t b = i >> 5;
    int k = 1;
    uint64_t c = 0;
    for (;;) {  // climb: blocks to the left of b inside b's group of 32
        const uint32_t* m = L.lv[k];
        uint64_t glo = b & ~31ull;
        bool found = false;
        for (uint64_t x = b; x-- > glo;)
            if (m[x] < v) { c = x; found = true; break; }
        if (found) break;
        b >>= 5;
        k++;
        if (k >= L.count) return 0u;  // unreachable (a[0] = 0 is smaller than every real value); bounds the loop
    }
    while (k > 1) {  // descend to the right-most child whose minimum is smaller
        const uint32_t* m = L.lv[k - 1];
        uint64_t base = c * 32, x = base + 32;
        if (x > L.size[k - 1]) x = L.size[k - 1];
        while (x-- > base)
            if (m[x] < v) break;
        c = x;
        k--;
    }
    uint64_t base = c * 32, j = base + 32;
    if (j > L.size[0]) j = L.size[0];
    while (j-- > base)
        if (a[j] < v) break;
    return (uint32_t)j;
}

__device__ __forceinline__ uint32_t nsv_search(const MinLevels& L, uint64_t i, uint32_t v) {
    const uint32_t* a = L.lv[0];
    uint64_t hi = (i | 31ull) + 1;
    if (hi > L.size[0]) hi = L.size[0];
    for (uint64_t j = i + 1; j < hi; j++)
        if (a[j] < v) return (uint32_t)j;
    uint64_t b = i >> 5;
    int k = 1;
    uint64_t c = 0;
    for (;;) {
        const uint32_t* m = L.lv[k];
        uint64_t ghi = (b | 31ull) + 1;
        if (ghi > L.size[k]) ghi = L.size[k];
        bool found = false;
        for (uint64_t x = b + 1; x < ghi; x++)
            if (m[x] < v) { c = x; found = true; break; }
        if (found) break;
        b >>= 5;
        k++;
        if (k >= L.count) return (uint32_t)(L.size[0] - 1);  // unreachable (the last value is 0); bounds the loop
    }
    while (k > 1) {
        const uint32_t* m = L.lv[k - 1];
        uint64_t x = c * 32, end = x + 32;
        if (end > L.size[k - 1]) end = L.size[k - 1];
        for (; x < end; x++)
            if (m[x] < v) break;
        c = x;
        k--;
    }
    uint64_t j = c * 32, end = j + 32;
    if (end > L.size[0]) end = L.size[0];
    for (; j < end; j++)
        if (a[j] < v) break;
    return (uint32_t)j;
}

__global__ void __launch_bounds__(256) k_links(MinLevels L, uint32_t rows, uint32_t* __restrict__ psv,
                                               uint32_t* __restrict__ nsv) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;  // rows 1..n
    if (i >= rows) return;
    uint32_t v = L.lv[0][i];  // >= 1 for every real row
    psv[i] = psv_search(L, i, v);
    nsv[i] = nsv_search(L, i, v);
}

// ------------------------------------------------------------------------------------------
// download helpers (structure-level parity)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bwt_codes(IndexView ix, uint8_t* __restrict__ out) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > ix.n) return;
    const FMBlock* b = &ix.fm[r >> kFmRowsLog2];
    uint32_t o = (uint32_t)r & (kFmRows - 1), hsel = o >> 6, bit = o & 63u;
    uint32_t code;
    if ((b->ex[hsel] >> bit) & 1ull) code = (r == ix.dollar_row) ? 0u : 1u;
    else code = 2u + (uint32_t)((b->p0[hsel] >> bit) & 1ull) + 2u * (uint32_t)((b->p1[hsel] >> bit) & 1ull);
    out[r] = (uint8_t)code;
}

// field 0: LCP (as int32, -1 sentinels), 1: PSV, 2: NSV (entries 0 .. n+1), 3: SA (entries 0 .. n)
__global__ void __launch_bounds__(256) k_rec_field(const RowRec* __restrict__ rec, const uint32_t* __restrict__ sa,
                                                   uint32_t rows, uint64_t count, int field, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t v;
    if (field == 3) v = sa[i];
    else if (field == 0) v = (i < rows ? rec[i].lcp1 : rec[rows - 1u].lcp1n) - 1u;
    else if (field == 1) v = i < rows ? rec[i].psv : 0u;
    else v = i >= 1 ? rec[i - 1].nsvn : rows;
    out[i] = v;
}

// ------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------
namespace {

struct DevBuf {
    void* p = nullptr;
    bool owned = true;
    ~DevBuf() { release(); }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; }
    hipError_t alloc(size_t bytes) { owned = true; return hipMalloc(&p, bytes ? bytes : 16); }
    void view(void* q) { release(); p = q; owned = false; }  // memory that belongs to somebody else (the arena)
    template <class T> T* as() { return static_cast<T*>(p); }
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    hipError_t init() { hipError_t e = hipEventCreate(&a); return e != hipSuccess ? e : hipEventCreate(&b); }
};

inline unsigned grid_for(uint64_t items, unsigned block = 256) { return (unsigned)((items + block - 1) / block); }
inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
inline int bits_for(uint64_t v) { int b = 0; while ((1ull << b) <= v && b < 63) b++; return b ? b : 1; }

}  // namespace

void make_view(slamem_index* idx) {
    char* base = static_cast<char*>(idx->arena);
    const ArenaHeader& h = idx->hdr;
    idx->view.fm = reinterpret_cast<const FMBlock*>(base + h.off_fm);
    idx->view.rec = reinterpret_cast<const RowRec*>(base + h.off_rec);
    idx->view.sa = reinterpret_cast<const uint32_t*>(base + h.off_sa);
    idx->view.nrows = reinterpret_cast<const uint32_t*>(base + h.off_nrows);
    idx->view.kfilter = h.off_kfilter ? reinterpret_cast<const uint64_t*>(base + h.off_kfilter) : nullptr;
    idx->view.tgrp = h.off_tgrp ? reinterpret_cast<const TextGroup*>(base + h.off_tgrp) : nullptr;
    idx->view.prec = h.off_tgrp ? reinterpret_cast<const TextRec*>(base + h.off_prec) : nullptr;
    idx->view.kjump = h.off_kjump ? reinterpret_cast<const uint2*>(base + h.off_kjump) : nullptr;
    idx->view.kjump_k = h.off_kjump ? h.kjump_k : 0u;
    idx->view.kbits = h.off_kbits ? reinterpret_cast<const uint64_t*>(base + h.off_kbits) : nullptr;
    idx->view.kbits_k = h.off_kbits ? h.kbits_k : 0u;
    idx->view.kfilter_log2 = h.kfilter_log2;
    idx->view.kfilter_k = h.kfilter_k;
    idx->view.kfilter_levels = h.kfilter_levels == 2u ? 2u : 3u;
    idx->view.seed = h.off_seed ? reinterpret_cast<const SeedBucket*>(base + h.off_seed) : nullptr;
    idx->view.tpl = h.off_seed ? reinterpret_cast<const TextPlanes*>(base + h.off_tpl) : nullptr;
    idx->view.spill = h.off_seed ? reinterpret_cast<const uint64_t*>(base + h.off_spill) : nullptr;
    idx->view.seed_k = h.off_seed ? h.seed_k : 0u;
    idx->view.seed_log2 = h.off_seed ? h.seed_log2 : 0u;
    idx->view.n = h.n;
    idx->view.nblocks = h.nblocks;
    idx->view.dollar_row = h.dollar_row;
    idx->view.num_n = h.num_n;
}

// Arena layout for a text of n letters (num_n of them N) -- the header's section offsets and sizes.
//   layout 1 (full):    every section; ~37.5 B per letter + 16-32 B of presence filter
//   layout 2 (compact): no text-ordered sections (K8 walks the index where it would have compared with the text) and a
//                       presence filter of half the size (no (k+2)-mers): ~21.5 B per letter + 8-16 B of filter
// Environment switches (experiments): SLAMEM_KFILTER=0, SLAMEM_TEXT_SECTIONS=0, SLAMEM_KJUMP=<K>, SLAMEM_KBITS / SLAMEM_SKIP.
// (build_index_device: 0 when the seed table of a text of 2^28 letters or more does not fit the free HBM beside the build)
static thread_local int tl_big_seed = 1;
static void plan_arena(uint32_t n, uint32_t num_n, int layout, ArenaHeader& hdr) {
    const uint64_t R = (uint64_t)n + 1;
    const uint32_t nblocks = (uint32_t)((R + 1 + kFmRows - 1) >> kFmRowsLog2);  // occ(c, <= n) reads offset n+1
    const bool compact = layout == 2;
    hdr.magic_lo = kArenaMagicLo;
    hdr.magic_hi = kArenaMagicHi;
    hdr.version = kArenaVersion;
    hdr.n = n;
    hdr.nblocks = nblocks;
    hdr.num_n = num_n;
    hdr.layout = compact ? 2u : 1u;
    uint64_t off = kHeaderBytes;
    hdr.off_fm = off;    off = align_up(off + (uint64_t)nblocks * sizeof(FMBlock), 256);
    hdr.off_rec = off;   off = align_up(off + (R + 1) * sizeof(RowRec), 256);
    hdr.off_sa = off;    off = align_up(off + R * 4, 256);
    hdr.off_nrows = off; off = align_up(off + (uint64_t)(num_n ? num_n : 1) * 4, 256);
    {   // k-mer presence filter: 128 bits per text character (rounded up to a power of two of words), n >= k only
        const char* kf = getenv("SLAMEM_KFILTER");
        // k grows with the text: a random k-mer occurs with probability ~ n / 4^k, which must stay well below
        // 1 / (probes per strand) for the filter to discriminate; k = ceil(log4 n) + 4 keeps it near 0.2 %
        uint32_t kf_k = 4;
        for (uint64_t v = 1; v < (uint64_t)n; v <<= 2) kf_k++;
        if (kf_k < 12) kf_k = 12;
        if (kf_k > 26) kf_k = 26;  // (never reached: n < 2^32 gives k <= 20; the build packs k+6 letters into 64 bits)
        bool want = !(kf && atoi(kf) == 0) && n >= kf_k;
        if (want) {
            // two 64-bit words per text position (nine entries of three bits each per position: a sixth of the bits set)
            uint32_t lg = 10;
            while ((1ull << lg) < 2ull * (uint64_t)n && lg < 32) lg++;
            if (compact && lg > 10) lg--;  // half the words: one per position (or fewer), (k-2)- and k-mers only
            // texts above 2^31 letters get fewer than two words per position (2^32 words is the most the line index of
            // a hash allows here without a 64 GiB section): their filter holds no (k+2)-mers -- four entries per position
            // instead of nine, so that a test still lets only ~0.2 % through (with all nine at 1.4 words per position it
            // was 1.8 %: at 3.1 Gbp and -l 20, 7.3 M of 10 M strands survived instead of 6.3 M); the search then stops
            // its cascade at the k-mers
            hdr.kfilter_levels = (1ull << lg) < 2ull * (uint64_t)n ? 2u : 3u;
            hdr.off_kfilter = off;
            hdr.kfilter_log2 = lg;
            hdr.kfilter_k = kf_k;
            off = align_up(off + (8ull << lg), 256);
        }
    }
    const uint64_t ngroups = (R >> 4) + 2;
    {   // text-ordered sections for the direct extension of unique matches (K8); SLAMEM_TEXT_SECTIONS=0 builds without
        const char* de = getenv("SLAMEM_TEXT_SECTIONS");
        if (!compact && !(de && atoi(de) == 0)) {
            hdr.off_tgrp = off; off = align_up(off + ngroups * sizeof(TextGroup), 256);
            hdr.off_prec = off; off = align_up(off + R * sizeof(TextRec), 256);
        }
    }
    {   // K-mer jump table: K = floor(log4 n) - 1, at most 12 (134 MB); SLAMEM_KJUMP=0 builds without
        const char* kj = getenv("SLAMEM_KJUMP");
        uint32_t K = 0;
        for (uint64_t v = n; v >= 4; v >>= 2) K++;
        K = K > 1 ? K - 1 : 0;
        if (K > 12) K = 12;
        if (kj && atoi(kj) >= 0 && (uint32_t)atoi(kj) < K) K = (uint32_t)atoi(kj);
        if (K > 0) {
            hdr.kjump_k = K;
            hdr.off_kjump = off; off = align_up(off + (8ull << (2u * K)), 256);
        }
    }
    {   // k-mer occurrence bitmap: k = ceil(log4 n) + 2 (at most 16: the value fits 32 bits, the bitmap 512 MB), only while
        // fewer than a tenth of all k-mers occur -- a denser bitmap proves nothing absent.  Only the skipping states of K8
        // read it, and they are not the default: built when SLAMEM_SKIP=1 (or SLAMEM_KBITS=1) is set
        const char* kb = getenv("SLAMEM_KBITS");
        const char* sk = getenv("SLAMEM_SKIP");
        const bool want_kbits = !compact && ((kb && atoi(kb) != 0) || (!kb && sk && atoi(sk) != 0));
        uint32_t k = 2;
        for (uint64_t v = 1; v < (uint64_t)n; v <<= 2) k++;
        if (k < 8) k = 8;
        if (k > 16) k = 16;
        if (want_kbits && n >= k && (uint64_t)n * 10ull <= (1ull << (2u * k))) {
            hdr.kbits_k = k;
            hdr.off_kbits = off; off = align_up(off + ((1ull << (2u * k)) >> 3), 256);
        }
    }
    {   // seed table + text units (the seed-and-compare path for reads): full layout; SLAMEM_SEED=0 builds without.  Two to four
        // k-mers per 12-slot bucket for texts below 2^28 letters (more than 12 in one bucket: 3e-4 of the buckets at four), k so
        // that the tag fits 7 bits: k = 10 .. 16.  Texts of 2^28 letters and more (round 4: configs[4]'s 3.1 Gbp): four to eight
        // per bucket (34 GB of table at 3.1 Gbp instead of 69; 0.6 % of the buckets use the spill list) and seeds of 16 to 18
        // letters -- keys of up to 36 bits -- unless the caller found that the table does not fit beside the build (g_big_seed)
        const char* se = getenv("SLAMEM_SEED");
        const char* sk = getenv("SLAMEM_SEED_K");  // (tests: seeds of 17 / 18 letters on a small text: as many buckets as the tag needs)
        const bool big = n >= (1u << 28);
        if (!compact && !(se && atoi(se) == 0) && n >= 64u && (!big || tl_big_seed)) {
            uint32_t lg = 10;
            while (((big ? 8ull : 4ull) << lg) < (uint64_t)n) lg++;
            uint32_t k = (lg + 7u) / 2u;
            if (k > kSeedMaxK) k = kSeedMaxK;
            if (!big && k > 16u) k = 16u;
            // (experiments: SLAMEM_SEED=<k> asks for shorter seeds -- fewer windows per read, more chance occurrences)
            if (se && atoi(se) >= 8 && (uint32_t)atoi(se) < k && 2u * (uint32_t)atoi(se) >= lg) k = (uint32_t)atoi(se);
            if (sk && (atoi(sk) == 17 || atoi(sk) == 18)) { k = (uint32_t)atoi(sk); if (lg < 2u * k - 7u) lg = 2u * k - 7u; }
            hdr.seed_k = k;
            hdr.seed_log2 = lg;
            const uint64_t units = text_units(n);
            hdr.off_seed = off; off = align_up(off + (sizeof(SeedBucket) << lg), 256);
            hdr.off_tpl = off;  off = align_up(off + units * sizeof(TextPlanes), 256);
            hdr.spill_cap = (uint32_t)seed_spill_entries(n);
            hdr.off_spill = off; off = align_up(off + (uint64_t)hdr.spill_cap * 8, 256);
        }
    }
    hdr.total_bytes = off;
}

// What the suffix sort borrows from the arena (regions written only after it) and what it must allocate beside it.
// Shared by the build (which walks the same list) and by the estimate below.
static const uint64_t kBorrowBytesPerRow[6] = {8, 4, 4, 4, 4, 4};  // keysB, valsA, valsB, tmp32, gh, posB
static uint64_t own_scratch_bytes(const ArenaHeader& hdr) {
    const uint64_t R = (uint64_t)hdr.n + 1;
    uint64_t left[3] = {(R + 1) * sizeof(RowRec), hdr.off_kfilter ? (8ull << hdr.kfilter_log2) : 0ull,
                        hdr.off_prec ? R * sizeof(TextRec) : 0ull};
    // never borrowed: packed text, keysA (later LCP+1 and PSV), rank, two flag arrays, posA (later NSV), radix-sort scratch
    uint64_t own = ((R + 15) / 16 + 264) * 8 + ((R + 1) * 8 + 64) + R * 4 + 2 * R + (R + 1) * 4 + R * 4 + (64ull << 20);
    for (uint64_t per : kBorrowBytesPerRow) {
        uint64_t bytes = align_up(R * per, 16);
        bool fit = false;
        for (uint64_t& l : left)
            if (!fit && l >= bytes) { l -= bytes; fit = true; }
        if (!fit) own += bytes;
    }
    return own;
}

int estimate_build_bytes(uint32_t n, int layout, uint64_t* arena_bytes, uint64_t* peak_bytes) {
    if (n == 0 || n > 0xFFFFFFF0u || (layout != 1 && layout != 2)) {
        set_error("slamem_index_build_bytes: n must be 1 .. 2^32-17 and layout 1 (full) or 2 (compact)");
        return SLAMEM_ERR_ARG;
    }
    ArenaHeader hdr;
    memset(&hdr, 0, sizeof(hdr));
    plan_arena(n, 0, layout, hdr);
    if (arena_bytes) *arena_bytes = hdr.total_bytes;
    if (peak_bytes) *peak_bytes = hdr.total_bytes + own_scratch_bytes(hdr);
    return SLAMEM_OK;
}

// layout 0: full when its build peak fits the free HBM of the device (SLAMEM_HBM_BUDGET_GB caps what counts as free),
// compact when only that fits; SLAMEM_INDEX_LAYOUT=full|compact decides for callers that pass 0.
static int choose_layout(uint32_t n, int layout, int* chosen) {
    if (layout == 1 || layout == 2) { *chosen = layout; return SLAMEM_OK; }
    if (layout != 0) { set_error("slamem_index_build: layout must be 0 (auto), 1 (full) or 2 (compact)"); return SLAMEM_ERR_ARG; }
    const char* env = getenv("SLAMEM_INDEX_LAYOUT");
    if (env && (!strcmp(env, "full") || !strcmp(env, "1"))) { *chosen = 1; return SLAMEM_OK; }
    if (env && (!strcmp(env, "compact") || !strcmp(env, "2"))) { *chosen = 2; return SLAMEM_OK; }
    size_t free_b = 0, total_b = 0;
    SLAMEM_HIP(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = free_b;
    if ((env = getenv("SLAMEM_HBM_BUDGET_GB")) != nullptr && atof(env) > 0) {
        uint64_t cap = (uint64_t)(atof(env) * 1073741824.0);
        if (cap < budget) budget = cap;
    }
    uint64_t arena = 0, peak_full = 0, peak_compact = 0;
    estimate_build_bytes(n, 1, &arena, &peak_full);
    estimate_build_bytes(n, 2, &arena, &peak_compact);
    const uint64_t margin = 256ull << 20;
    if (peak_full + margin <= budget) { *chosen = 1; return SLAMEM_OK; }
    if (peak_compact + margin <= budget) { *chosen = 2; return SLAMEM_OK; }
    set_error("slamem_index_build: a text of %u letters needs %.1f GB of HBM while it is built (%.1f GB in the compact layout); "
              "%.1f GB are free", n, (double)peak_full / 1e9, (double)peak_compact / 1e9, (double)budget / 1e9);
    return SLAMEM_ERR_NOMEM;
}

int build_index_device(const void* text_dev, uint32_t n, int device, hipStream_t stream, int layout_arg, slamem_index** out) {
    if (!text_dev || !out || n == 0 || n > 0xFFFFFFF0u) {
        set_error("slamem_index_build: text must hold 1 .. 2^32-17 characters");
        return SLAMEM_ERR_ARG;
    }
    SLAMEM_HIP(hipSetDevice(device));
    int layout = 0;
    struct BigSeedReset { ~BigSeedReset() { tl_big_seed = 1; } } big_seed_reset;  // (the estimate calls of other threads' callers see the default)
    tl_big_seed = 1;
    if (n >= (1u << 28)) {
        // the seed table of a text this long (34 GB at 3.1 Gbp) is built only when the full layout's build peak still fits the
        // free HBM with it (SLAMEM_HBM_BUDGET_GB caps what counts as free; SLAMEM_SEED_BIG=0: never)
        size_t free_b = 0, total_b = 0;
        SLAMEM_HIP(hipMemGetInfo(&free_b, &total_b));
        uint64_t budget = free_b, arena = 0, peak = 0;
        const char* env = getenv("SLAMEM_HBM_BUDGET_GB");
        if (env && atof(env) > 0 && (uint64_t)(atof(env) * 1073741824.0) < budget) budget = (uint64_t)(atof(env) * 1073741824.0);
        estimate_build_bytes(n, 1, &arena, &peak);
        const char* sb = getenv("SLAMEM_SEED_BIG");
        if ((sb && atoi(sb) == 0) || peak + (2048ull << 20) > budget) tl_big_seed = 0;
    }
    {
        int rc = choose_layout(n, layout_arg, &layout);
        if (rc != SLAMEM_OK) return rc;
    }
    const uint32_t rows = n + 1;
    const uint64_t R = rows;
    constexpr uint64_t kPackSlack = 264;  // zero words behind the text: window16() reads one word ahead, wave_extend_lcp up to 4096 letters more
    const uint64_t nwords = (R + 15) / 16 + kPackSlack;
    const uint32_t nblocks = (uint32_t)((R + 1 + kFmRows - 1) >> kFmRowsLog2);  // occ(c, <= n) reads offset n+1

    Timings& tm = thread_timings();
    // SLAMEM_BUILD_TRACE=1: host wall clock of every phase (allocations included) on stderr
    const bool trace = getenv("SLAMEM_BUILD_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_last = t_begin;
    auto mark = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[build] %-28s %8.1f ms  (at %.1f ms)\n", what,
                std::chrono::duration<double, std::milli>(now - t_last).count(),
                std::chrono::duration<double, std::milli>(now - t_begin).count());
        t_last = now;
    };
    EventPair ev_all, ev;
    SLAMEM_HIP(ev_all.init());
    SLAMEM_HIP(ev.init());
    SLAMEM_HIP(hipEventRecord(ev_all.a, stream));

    // ---- K1 ---------------------------------------------------------------------------------
    DevBuf pk, scal;
    SLAMEM_HIP(pk.alloc(nwords * 8));
    SLAMEM_HIP(scal.alloc(64 * 4));  // [0..5] histogram, [8] dollar_row, [9] max_lcp, [10] select count
    uint32_t* d_scal = scal.as<uint32_t>();
    SLAMEM_HIP(hipMemsetAsync(pk.p, 0, nwords * 8, stream));
    SLAMEM_HIP(hipMemsetAsync(d_scal, 0, 64 * 4, stream));
    SLAMEM_HIP(hipEventRecord(ev.a, stream));
    hipLaunchKernelGGL(k_pack_text, dim3(grid_for(nwords - kPackSlack)), dim3(256), 0, stream,
                       static_cast<const uint8_t*>(text_dev), n, pk.as<uint64_t>(), nwords - kPackSlack, d_scal);
    SLAMEM_HIP(hipGetLastError());
    SLAMEM_HIP(hipEventRecord(ev.b, stream));
    uint32_t h_scal[64];
    SLAMEM_HIP(hipMemcpyAsync(h_scal, d_scal, sizeof(h_scal), hipMemcpyDeviceToHost, stream));
    SLAMEM_HIP(hipStreamSynchronize(stream));
    SLAMEM_HIP(hipEventElapsedTime(&tm.t.build_pack_ms, ev.a, ev.b));
    const uint32_t num_n = h_scal[1];
    mark("K1 pack (+alloc)");

    // ---- arena ------------------------------------------------------------------------------
    ArenaHeader hdr;
    memset(&hdr, 0, sizeof(hdr));
    plan_arena(n, num_n, layout, hdr);
    hdr.C[0] = 0;
    hdr.C[1] = 1;
    for (int c = 2; c < 6; c++) hdr.C[c] = hdr.C[c - 1] + h_scal[c - 1];
    const uint64_t ngroups = (R >> 4) + 2;

    // ---- memory plan -------------------------------------------------------------------------------------------
    // The arena is allocated FIRST and lends its row-record and filter regions to the suffix sort (they are written
    // only after it), the LCP / PSV / NSV arrays reuse sort buffers, and nothing is freed before the end: the driver
    // wipes freed VRAM in the background at ~33 GiB/s and an allocation that needs memory still waiting for its wipe
    // blocks (measured at 3.1 Gbp: 9.3 of 10.7 s of the build were two such hipMalloc calls; tools/malloc_probe.hip).
    // Peak = arena + 22.5 B per row (the suffix array is built in place in the arena).
    DevBuf arena;
    if (arena.alloc(hdr.total_bytes) != hipSuccess) {
        set_error("slamem_index_build: cannot allocate %llu bytes of HBM for the index", (unsigned long long)hdr.total_bytes);
        return SLAMEM_ERR_NOMEM;
    }
    char* base = arena.as<char>();
    struct Region { char* p; uint64_t left; };
    Region lend[3] = {{base + hdr.off_rec, (R + 1) * sizeof(RowRec)},
                      {hdr.off_kfilter ? base + hdr.off_kfilter : nullptr, hdr.off_kfilter ? (8ull << hdr.kfilter_log2) : 0ull},
                      {hdr.off_prec ? base + hdr.off_prec : nullptr, hdr.off_prec ? R * sizeof(TextRec) : 0ull}};
    auto borrow = [&](DevBuf& b, uint64_t bytes) -> hipError_t {  // from the arena if it fits, else an own allocation
        bytes = align_up(bytes, 16);
        for (Region& r : lend)
            if (r.p && r.left >= bytes) { b.view(r.p); r.p += bytes; r.left -= bytes; return hipSuccess; }
        return b.alloc(bytes);
    };

    // ---- K2: suffix sort ----------------------------------------------------------------------
    DevBuf keysA, keysB, valsA, valsB, rank, flagA, flagB, tmp32, gh, posA, posB, sorttmp, sabuf;
    // keysA later holds LCP+1 and PSV, posA holds NSV (R+1 words each): sized for that, never borrowed
    sabuf.view(base + hdr.off_sa);  // the suffix array is built in its place in the arena
    if (keysA.alloc((R + 1) * 8 + 64) != hipSuccess || rank.alloc(R * 4) != hipSuccess ||
        flagA.alloc(R) != hipSuccess || flagB.alloc(R) != hipSuccess || posA.alloc((R + 1) * 4) != hipSuccess ||
        borrow(keysB, R * 8) != hipSuccess || borrow(valsA, R * 4) != hipSuccess || borrow(valsB, R * 4) != hipSuccess ||
        borrow(tmp32, R * 4) != hipSuccess || borrow(gh, R * 4) != hipSuccess || borrow(posB, R * 4) != hipSuccess) {
        set_error("slamem_index_build: cannot allocate suffix-sort scratch (%llu rows)", (unsigned long long)R);
        return SLAMEM_ERR_NOMEM;
    }
    mark("arena + scratch hipMalloc");
    uint32_t* d_sa = sabuf.as<uint32_t>();
    size_t tmp_bytes = 0, need = 0;
    SLAMEM_HIP(sort_pairs_u64_u32(nullptr, need, keysA.as<uint64_t>(), keysB.as<uint64_t>(), valsA.as<uint32_t>(),
                                  valsB.as<uint32_t>(), R, 0, 64, stream));
    tmp_bytes = need;
    SLAMEM_HIP(scan_max_inclusive_u32(nullptr, need, tmp32.as<uint32_t>(), gh.as<uint32_t>(), R, stream));
    tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
    SLAMEM_HIP(select_indices_u32(nullptr, need, flagA.as<uint8_t>(), posA.as<uint32_t>(), d_scal + 10, R, stream));
    tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
    SLAMEM_HIP(select_flagged_u32(nullptr, need, posA.as<uint32_t>(), flagA.as<uint8_t>(), posB.as<uint32_t>(), d_scal + 10, R, stream));
    tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
    SLAMEM_HIP(scan_sum_exclusive_uint4(nullptr, need, (const uint4*)nullptr, (uint4*)nullptr, nblocks, stream));
    tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
    if (tmp_bytes < scan_u32_tmp_words(R) * 4) tmp_bytes = scan_u32_tmp_words(R) * 4;
    SLAMEM_HIP(sorttmp.alloc(tmp_bytes));

    SLAMEM_HIP(hipEventRecord(ev.a, stream));
    hipLaunchKernelGGL(k_make_keys, dim3(grid_for(R)), dim3(256), 0, stream, pk.as<uint64_t>(), rows,
                       keysA.as<uint64_t>(), valsA.as<uint32_t>());
    SLAMEM_HIP(hipGetLastError());
    need = tmp_bytes;
    SLAMEM_HIP(sort_pairs_u64_u32(sorttmp.p, need, keysA.as<uint64_t>(), keysB.as<uint64_t>(), valsA.as<uint32_t>(),
                                  d_sa, R, 0, 48, stream));
    hipLaunchKernelGGL(k_heads, dim3(grid_for(R)), dim3(256), 0, stream, keysB.as<uint64_t>(), R,
                       (const uint32_t*)nullptr, flagA.as<uint8_t>(), tmp32.as<uint32_t>());
    need = tmp_bytes;
    SLAMEM_HIP(scan_max_inclusive_u32(sorttmp.p, need, tmp32.as<uint32_t>(), gh.as<uint32_t>(), R, stream));
    hipLaunchKernelGGL(k_assign_ranks, dim3(grid_for(R)), dim3(256), 0, stream, d_sa, gh.as<uint32_t>(),
                       flagA.as<uint8_t>(), R, (const uint32_t*)nullptr, (uint32_t*)nullptr, rank.as<uint32_t>(),
                       flagB.as<uint8_t>());
    SLAMEM_HIP(hipGetLastError());
    need = tmp_bytes;
    SLAMEM_HIP(select_indices_u32(sorttmp.p, need, flagB.as<uint8_t>(), posA.as<uint32_t>(), d_scal + 10, R, stream));
    uint32_t m = 0;
    SLAMEM_HIP(hipMemcpyAsync(&m, d_scal + 10, 4, hipMemcpyDeviceToHost, stream));
    SLAMEM_HIP(hipStreamSynchronize(stream));

    uint32_t rounds = 0;
    uint64_t h = 16;
    uint32_t* pos_cur = posA.as<uint32_t>();
    uint32_t* pos_nxt = posB.as<uint32_t>();
    const int key_bits = 32 + bits_for(R);
    while (m > 0) {
        rounds++;
        if (rounds > 40) { set_error("slamem_index_build: suffix sort did not converge"); return SLAMEM_ERR_HIP; }
        hipLaunchKernelGGL(k_round_keys, dim3(grid_for(m)), dim3(256), 0, stream, pos_cur, (uint64_t)m, d_sa,
                           rank.as<uint32_t>(), n, h, keysA.as<uint64_t>(), valsA.as<uint32_t>());
        SLAMEM_HIP(hipGetLastError());
        need = tmp_bytes;
        SLAMEM_HIP(sort_pairs_u64_u32(sorttmp.p, need, keysA.as<uint64_t>(), keysB.as<uint64_t>(), valsA.as<uint32_t>(),
                                      valsB.as<uint32_t>(), m, 0, key_bits > 64 ? 64 : key_bits, stream));
        hipLaunchKernelGGL(k_heads, dim3(grid_for(m)), dim3(256), 0, stream, keysB.as<uint64_t>(), (uint64_t)m,
                           (const uint32_t*)pos_cur, flagA.as<uint8_t>(), tmp32.as<uint32_t>());
        need = tmp_bytes;
        SLAMEM_HIP(scan_max_inclusive_u32(sorttmp.p, need, tmp32.as<uint32_t>(), gh.as<uint32_t>(), m, stream));
        hipLaunchKernelGGL(k_assign_ranks, dim3(grid_for(m)), dim3(256), 0, stream, valsB.as<uint32_t>(),
                           gh.as<uint32_t>(), flagA.as<uint8_t>(), (uint64_t)m, (const uint32_t*)pos_cur, d_sa,
                           rank.as<uint32_t>(), flagB.as<uint8_t>());
        SLAMEM_HIP(hipGetLastError());
        need = tmp_bytes;
        SLAMEM_HIP(select_flagged_u32(sorttmp.p, need, pos_cur, flagB.as<uint8_t>(), pos_nxt, d_scal + 10, m, stream));
        SLAMEM_HIP(hipMemcpyAsync(&m, d_scal + 10, 4, hipMemcpyDeviceToHost, stream));
        SLAMEM_HIP(hipStreamSynchronize(stream));
        uint32_t* t = pos_cur; pos_cur = pos_nxt; pos_nxt = t;
        h *= 2;
    }
    SLAMEM_HIP(hipEventRecord(ev.b, stream));
    SLAMEM_HIP(hipEventSynchronize(ev.b));
    SLAMEM_HIP(hipEventElapsedTime(&tm.t.build_sort_ms, ev.a, ev.b));
    hdr.sort_rounds = rounds;
    // rank[] is now the inverse suffix array; the borrowed regions of the arena are free again.
    mark("K2 suffix sort");
    FMBlock* d_fm = reinterpret_cast<FMBlock*>(base + hdr.off_fm);
    RowRec* d_rec = reinterpret_cast<RowRec*>(base + hdr.off_rec);
    uint32_t* d_nrows = reinterpret_cast<uint32_t*>(base + hdr.off_nrows);
    SLAMEM_HIP(hipMemsetAsync(base, 0, kHeaderBytes, stream));
    if (hdr.off_kjump) {  // before anything else reuses the sort's scratch: the row keys live in tmp32
        uint2* d_kj = reinterpret_cast<uint2*>(base + hdr.off_kjump);
        hipLaunchKernelGGL(k_kjump_init, dim3(grid_for(1ull << (2u * hdr.kjump_k))), dim3(256), 0, stream, d_kj, 1ull << (2u * hdr.kjump_k));
        hipLaunchKernelGGL(k_kjump_keys, dim3(grid_for(R)), dim3(256), 0, stream, d_sa, pk.as<uint64_t>(), n, hdr.kjump_k, tmp32.as<uint32_t>());
        hipLaunchKernelGGL(k_kjump_bounds, dim3(grid_for(R)), dim3(256), 0, stream, tmp32.as<uint32_t>(), n, d_kj);
        SLAMEM_HIP(hipGetLastError());
    }
    mark("K1c k-mer jump table");
    if (hdr.off_kbits) {
        unsigned long long* d_bits = reinterpret_cast<unsigned long long*>(base + hdr.off_kbits);
        SLAMEM_HIP(hipMemsetAsync(d_bits, 0, (1ull << (2u * hdr.kbits_k)) >> 3, stream));
        hipLaunchKernelGGL(k_kbits_build, dim3(grid_for((uint64_t)n - hdr.kbits_k + 1)), dim3(256), 0, stream, pk.as<uint64_t>(), n,
                           hdr.kbits_k, d_bits);
        SLAMEM_HIP(hipGetLastError());
    }
    if (hdr.off_seed) {  // text bit-planes, then the seed table by sorting the positions by bucket (the sort's buffers are free).
                         // BEFORE the presence filter is built: the B buffers may lie in the filter's region
        const uint64_t units = text_units(n);
        TextPlanes* d_tpl = reinterpret_cast<TextPlanes*>(base + hdr.off_tpl);
        SeedBucket* d_seed = reinterpret_cast<SeedBucket*>(base + hdr.off_seed);
        hipLaunchKernelGGL(k_text_planes, dim3(grid_for(units)), dim3(256), 0, stream, pk.as<uint64_t>(), n, units, d_tpl);
        hipLaunchKernelGGL(k_seed_keys, dim3(grid_for(units * 64)), dim3(256), 0, stream, d_tpl, n,
                           hdr.seed_k, hdr.seed_log2, keysB.as<uint64_t>(), valsB.as<uint32_t>(), units);
        SLAMEM_HIP(hipGetLastError());
        need = tmp_bytes;
        SLAMEM_HIP(sort_pairs_u64_u32(sorttmp.p, need, keysB.as<uint64_t>(), keysA.as<uint64_t>(), valsB.as<uint32_t>(),
                                      valsA.as<uint32_t>(), n, 0, (int)hdr.seed_log2 + 8 + 1, stream));
        SLAMEM_HIP(hipMemsetAsync(d_seed, 0, sizeof(SeedBucket) << hdr.seed_log2, stream));
        // (the sort's input buffers are free: what the runs ask of the spill list, and the places they get)
        uint32_t* d_want = valsB.as<uint32_t>();
        uint32_t* d_place = keysB.as<uint32_t>();
        hipLaunchKernelGGL(k_seed_fill, dim3(grid_for(n)), dim3(256), 0, stream, (const uint64_t*)keysA.as<uint64_t>(),
                           (const uint32_t*)valsA.as<uint32_t>(), (uint64_t)n, hdr.seed_log2, d_seed, d_want,
                           d_tpl);
        SLAMEM_HIP(hipGetLastError());
        SLAMEM_HIP(exclusive_scan_u32(d_want, d_place, n, static_cast<uint32_t*>(sorttmp.p), stream));
        SLAMEM_HIP(hipMemsetAsync(d_scal + 11, 0, 4, stream));
        hipLaunchKernelGGL(k_seed_spill, dim3(grid_for(n)), dim3(256), 0, stream, (const uint64_t*)keysA.as<uint64_t>(),
                           (const uint32_t*)valsA.as<uint32_t>(), (const uint32_t*)d_want, (const uint32_t*)d_place, (uint64_t)n,
                           d_seed, reinterpret_cast<uint64_t*>(base + hdr.off_spill), hdr.spill_cap, d_scal + 11);
        SLAMEM_HIP(hipGetLastError());
        SLAMEM_HIP(hipMemcpyAsync(&hdr.spill_used, d_scal + 11, 4, hipMemcpyDeviceToHost, stream));
        SLAMEM_HIP(hipStreamSynchronize(stream));
        if (hdr.spill_used < hdr.spill_cap)  // (a saved index is its bytes: no stale memory in it)
            SLAMEM_HIP(hipMemsetAsync(base + hdr.off_spill + (uint64_t)hdr.spill_used * 8, 0xFF, (uint64_t)(hdr.spill_cap - hdr.spill_used) * 8, stream));
        mark("K1d seed table");
    }
    if (hdr.off_kfilter) {
        unsigned long long* d_filter = reinterpret_cast<unsigned long long*>(base + hdr.off_kfilter);
        static const bool env_atomic = [] { const char* v = getenv("SLAMEM_KFILTER_ATOMIC"); return v && atoi(v) != 0; }();
        if (env_atomic || n < (1u << 16)) {
            SLAMEM_HIP(hipMemsetAsync(d_filter, 0, 8ull << hdr.kfilter_log2, stream));
            hipLaunchKernelGGL(k_kfilter_build, dim3(grid_for((uint64_t)n - hdr.kfilter_k + 3)), dim3(256), 0, stream,
                               pk.as<uint64_t>(), n, hdr.kfilter_k, hdr.kfilter_log2, hdr.kfilter_levels, d_filter);
        } else {
            // the suffix sort's buffers are free: keys in keysA / keysB, positions in valsA / valsB.  valsB (and nothing else
            // of these) may have been borrowed from the filter's own region: the result is asked for in the A buffers, and
            // the region is zeroed only after the sort is through with the B buffers (same stream: in order)
            hipLaunchKernelGGL(k_kfilter_keys, dim3(grid_for(n)), dim3(256), 0, stream, pk.as<uint64_t>(), n, hdr.kfilter_k,
                               hdr.kfilter_log2, keysB.as<uint64_t>(), valsB.as<uint32_t>());
            SLAMEM_HIP(hipGetLastError());
            mark("K1b filter keys");
            need = tmp_bytes;
            SLAMEM_HIP(sort_pairs_u64_u32(sorttmp.p, need, keysB.as<uint64_t>(), keysA.as<uint64_t>(), valsB.as<uint32_t>(),
                                          valsA.as<uint32_t>(), n, 0, (int)hdr.kfilter_log2 - 3 + 1, stream));
            mark("K1b filter sort");
            SLAMEM_HIP(hipMemsetAsync(d_filter, 0, 8ull << hdr.kfilter_log2, stream));
            hipLaunchKernelGGL(k_kfilter_fill, dim3(grid_for(n)), dim3(256), 0, stream, pk.as<uint64_t>(), n, hdr.kfilter_k,
                               hdr.kfilter_log2, hdr.kfilter_levels, (const uint64_t*)keysA.as<uint64_t>(),
                               (const uint32_t*)valsA.as<uint32_t>(), d_filter);
        }
        SLAMEM_HIP(hipGetLastError());
    }


    mark("K1b filter fill");
    // ---- K3: BWT planes + rank samples ------------------------------------------------------------
    SLAMEM_HIP(hipEventRecord(ev.a, stream));
    SLAMEM_HIP(hipMemsetAsync(d_fm, 0, (uint64_t)nblocks * sizeof(FMBlock), stream));
    uint4* d_half = keysA.as<uint4>();            // 2*nblocks uint4  (scratch reuse: 4*nblocks*16 <= R*8 for R >= 8*nblocks)
    uint4* d_blk = d_half + 2 * (uint64_t)nblocks;  // nblocks uint4
    uint4* d_pre = d_blk + nblocks;               // nblocks uint4
    DevBuf small;                                 // tiny texts: the aliasing bound above fails, use a private buffer
    if ((uint64_t)nblocks * 64 > R * 8) {
        SLAMEM_HIP(small.alloc((uint64_t)nblocks * 64));
        d_half = small.as<uint4>();
        d_blk = d_half + 2 * (uint64_t)nblocks;
        d_pre = d_blk + nblocks;
    }
    SLAMEM_HIP(hipMemsetAsync(d_half, 0, (uint64_t)nblocks * 32, stream));
    hipLaunchKernelGGL(k_bwt_planes, dim3(grid_for((uint64_t)nblocks * kFmRows)), dim3(256), 0, stream, d_sa,
                       pk.as<uint64_t>(), rows, nblocks, d_fm, d_half, flagA.as<uint8_t>(), d_scal + 8);
    SLAMEM_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_block_counts, dim3(grid_for(nblocks)), dim3(256), 0, stream, d_half, nblocks, d_blk);
    need = tmp_bytes;
    SLAMEM_HIP(scan_sum_exclusive_uint4(sorttmp.p, need, d_blk, d_pre, nblocks, stream));
    hipLaunchKernelGGL(k_rank_samples, dim3(grid_for(nblocks)), dim3(256), 0, stream, d_pre, nblocks,
                       make_uint4(hdr.C[2], hdr.C[3], hdr.C[4], hdr.C[5]), d_fm);
    SLAMEM_HIP(hipGetLastError());
    if (num_n) {
        need = tmp_bytes;
        SLAMEM_HIP(select_indices_u32(sorttmp.p, need, flagA.as<uint8_t>(), d_nrows, d_scal + 10, R, stream));
    }
    SLAMEM_HIP(hipEventRecord(ev.b, stream));
    SLAMEM_HIP(hipEventSynchronize(ev.b));
    SLAMEM_HIP(hipEventElapsedTime(&tm.t.build_bwt_ms, ev.a, ev.b));

    // ---- K5: LCP --------------------------------------------------------------------------------------
    SLAMEM_HIP(hipEventRecord(ev.a, stream));
    // scratch reuse: LCP+1 and PSV in the first sort key buffer (its use as K3 scratch is over), NSV in posA
    mark("K3 BWT");
    uint32_t* d_l32 = keysA.as<uint32_t>();
    uint32_t* d_psv = d_l32 + align_up(R + 1, 4);
    uint32_t* d_nsv = posA.as<uint32_t>();
    {   // sampled passes (scratch: the PSV / NSV buffers, free until K7)
        uint64_t n2 = (R + kLcpCoarse2 - 1) / kLcpCoarse2, n1 = (R + kLcpCoarse1 - 1) / kLcpCoarse1;
        hipLaunchKernelGGL(k_lcp_sampled, dim3(grid_for(n2 * 64)), dim3(256), 0, stream, pk.as<uint64_t>(), d_sa,
                           rank.as<uint32_t>(), rows, kLcpCoarse2, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 1u,
                           d_psv, d_psv + n2);
        hipLaunchKernelGGL(k_lcp_sampled, dim3(grid_for(n1 * 64)), dim3(256), 0, stream, pk.as<uint64_t>(), d_sa,
                           rank.as<uint32_t>(), rows, kLcpCoarse1, (const uint32_t*)d_psv, (const uint32_t*)(d_psv + n2),
                           kLcpCoarse2, d_nsv, d_nsv + n1);
        hipLaunchKernelGGL(k_lcp_kasai, dim3(grid_for((R + kLcpChunk - 1) / kLcpChunk)), dim3(256), 0, stream,
                           pk.as<uint64_t>(), d_sa, rank.as<uint32_t>(), rows, (const uint32_t*)d_nsv,
                           (const uint32_t*)(d_nsv + n1), d_l32, d_scal + 9);
    }
    // after the sampled passes: they use the PSV / NSV buffers as scratch
    hipLaunchKernelGGL(k_lcp_sentinels, dim3(1), dim3(64), 0, stream, rows, d_l32, d_psv, d_nsv);
    SLAMEM_HIP(hipGetLastError());
    SLAMEM_HIP(hipEventRecord(ev.b, stream));
    SLAMEM_HIP(hipEventSynchronize(ev.b));
    SLAMEM_HIP(hipEventElapsedTime(&tm.t.build_lcp_ms, ev.a, ev.b));

    // ---- K7: PSV / NSV ------------------------------------------------------------------------------------
    SLAMEM_HIP(hipEventRecord(ev.a, stream));
    MinLevels L;
    memset(&L, 0, sizeof(L));
    L.lv[0] = d_l32;
    L.size[0] = R + 1;
    L.count = 1;
    {
        uint32_t* lvbuf = valsA.as<uint32_t>();  // about (R+1)/31 words in total; valsA holds R words and is dead (if it
                                                 // was borrowed from the record region: that is written last, below)
        uint64_t used = 0, cap = R - 1;           // in uint32 units (one word kept for the guard level)
        uint64_t sz = R + 1;
        std::vector<uint64_t> sizes;
        uint64_t total = 0;
        for (uint64_t s = sz; s > 1;) { s = (s + 31) / 32; sizes.push_back(s); total += s; }
        if (total > cap) { set_error("slamem_index_build: internal: min-hierarchy scratch"); return SLAMEM_ERR_HIP; }
        for (size_t k = 0; k < sizes.size(); k++) {
            if (L.count >= kMaxLevels) { set_error("slamem_index_build: internal: too many min levels"); return SLAMEM_ERR_HIP; }
            uint32_t* outp = lvbuf + used;
            hipLaunchKernelGGL(k_min_level, dim3(grid_for(sizes[k])), dim3(256), 0, stream, L.lv[L.count - 1],
                               L.size[L.count - 1], outp, sizes[k]);
            L.lv[L.count] = outp;
            L.size[L.count] = sizes[k];
            L.count++;
            used += sizes[k];
        }
        // one extra all-zero top level so that a climb never indexes past the last real level
        if (L.count < kMaxLevels) {
            uint32_t* outp = lvbuf + used;
            SLAMEM_HIP(hipMemsetAsync(outp, 0, 4, stream));
            L.lv[L.count] = outp;
            L.size[L.count] = 1;
            L.count++;
        }
    }
    SLAMEM_HIP(hipGetLastError());
    if (n >= 1) {
        hipLaunchKernelGGL(k_links, dim3(grid_for(n)), dim3(256), 0, stream, L, rows, d_psv, d_nsv);
        SLAMEM_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_pack_records, dim3(grid_for(R + 1)), dim3(256), 0, stream, d_l32, d_psv, d_nsv, rows, d_rec, d_scal + 16);
    SLAMEM_HIP(hipGetLastError());
    if (hdr.off_tgrp) {  // text-ordered sections, last: their regions were lent to the sort, the records are complete now
        uint8_t* d_cls = flagB.as<uint8_t>();  // one byte per text position (scratch of the sort rounds, dead since K2)
        hipLaunchKernelGGL(k_text_records, dim3(grid_for(R)), dim3(256), 0, stream, rank.as<uint32_t>(), d_rec, rows,
                           reinterpret_cast<TextRec*>(base + hdr.off_prec), d_cls);
        hipLaunchKernelGGL(k_text_groups, dim3(grid_for(ngroups)), dim3(256), 0, stream, pk.as<uint64_t>(), d_cls, rows, ngroups,
                           reinterpret_cast<TextGroup*>(base + hdr.off_tgrp));
        SLAMEM_HIP(hipGetLastError());
    }
    SLAMEM_HIP(hipEventRecord(ev.b, stream));
    SLAMEM_HIP(hipMemcpyAsync(h_scal, d_scal, sizeof(h_scal), hipMemcpyDeviceToHost, stream));
    SLAMEM_HIP(hipStreamSynchronize(stream));
    SLAMEM_HIP(hipEventElapsedTime(&tm.t.build_links_ms, ev.a, ev.b));
    hdr.dollar_row = h_scal[8];
    hdr.max_lcp = h_scal[9];
    for (int b = 0; b < 10; b++) hdr.lcp_ge[b] = h_scal[16 + b];
    mark("K5 LCP + K7 links + records");

    SLAMEM_HIP(hipMemcpyAsync(base, &hdr, sizeof(hdr), hipMemcpyHostToDevice, stream));
    SLAMEM_HIP(hipEventRecord(ev_all.b, stream));
    SLAMEM_HIP(hipStreamSynchronize(stream));
    SLAMEM_HIP(hipEventElapsedTime(&tm.t.build_total_ms, ev_all.a, ev_all.b));

    slamem_index* idx = static_cast<slamem_index*>(calloc(1, sizeof(slamem_index)));
    if (!idx) { set_error("out of host memory"); return SLAMEM_ERR_NOMEM; }
    idx->hdr = hdr;
    idx->arena = arena.p;
    arena.p = nullptr;  // ownership moves to the handle
    idx->arena_bytes = hdr.total_bytes;
    idx->device = device;
    idx->owns_arena = 1;
    make_view(idx);
    *out = idx;
    return SLAMEM_OK;
}

// K6: the reference's run-sampling of the LCP array (lcparray.c:627-706) and its corner links (lcparray.c:782-956),
// as statistics over the per-row records.  acc: [0] samples [1] oversized lcp [2] sum lcp (as u64 two's complement)
// [3] max lcp [4] oversized links [5] sum |distance| [6] max |distance|
__global__ void __launch_bounds__(256) k_sampled_stats(const RowRec* __restrict__ rec, uint32_t n,
                                                       unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long sh[7];
    if (threadIdx.x < 7) sh[threadIdx.x] = 0;
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // row 0..n
    if (i <= n) {
        RowRec a = rec[i];
        long long lcp = (long long)a.lcp1 - 1, nxt = (long long)a.lcp1n - 1;
        // sum / max run over rows 1..n+1 (value of the NEXT row), lcparray.c:668-669
        atomicAdd(&sh[2], (unsigned long long)nxt);
        if (nxt > 0) atomicMax(&sh[3], (unsigned long long)nxt);
        if (lcp != nxt) {
            atomicAdd(&sh[0], 1ull);
            if (lcp == -1 || lcp >= 255) atomicAdd(&sh[1], 1ull);
            long long dist = -1;
            if (nxt > lcp) { if (i != 0) dist = (long long)i - (long long)a.psv; }          // top corner -> PSV
            else { if (i != n) dist = (long long)a.nsvn - 1 - (long long)i; }               // bottom corner -> NSV-1
            if (dist >= 0) {
                atomicAdd(&sh[5], (unsigned long long)dist);
                atomicMax(&sh[6], (unsigned long long)dist);
                if (dist >= 128) atomicAdd(&sh[4], 1ull);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 7 && sh[threadIdx.x]) {
        if (threadIdx.x == 3 || threadIdx.x == 6) atomicMax(&acc[threadIdx.x], sh[threadIdx.x]);
        else atomicAdd(&acc[threadIdx.x], sh[threadIdx.x]);
    }
}

int sampled_lcp_stats(const slamem_index* idx, slamem_sslcp_stats* out) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    DevBuf d;
    SLAMEM_HIP(d.alloc(8 * 8));
    SLAMEM_HIP(hipMemset(d.p, 0, 64));
    const uint64_t R = (uint64_t)idx->hdr.n + 1;
    hipLaunchKernelGGL(k_sampled_stats, dim3(grid_for(R)), dim3(256), 0, 0, idx->view.rec, idx->hdr.n,
                       d.as<unsigned long long>());
    SLAMEM_HIP(hipGetLastError());
    unsigned long long h[8];
    SLAMEM_HIP(hipMemcpy(h, d.p, 64, hipMemcpyDeviceToHost));
    out->num_samples = h[0];
    out->num_oversized_lcp = h[1];
    out->sum_lcp = (int64_t)h[2];
    out->max_lcp = (uint32_t)h[3];
    out->pad = 0;
    out->num_oversized_links = h[4] + 2;  // the first and the last sample are stored as oversized (lcparray.c:755,966-970)
    out->sum_link_distance = h[5];
    out->max_link_distance = h[6];
    return SLAMEM_OK;
}

int download_array(const slamem_index* idx, int which, void* host_dst, uint64_t count) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    const uint64_t R = (uint64_t)idx->hdr.n + 1;
    const IndexView& v = idx->view;
    int field = -1;
    uint64_t want = 0;
    switch (which) {
    case SLAMEM_ARRAY_SA: field = 3; want = R; break;
    case SLAMEM_ARRAY_LCP: field = 0; want = R + 1; break;
    case SLAMEM_ARRAY_PSV: field = 1; want = R + 1; break;
    case SLAMEM_ARRAY_NSV: field = 2; want = R + 1; break;
    case SLAMEM_ARRAY_BWT: want = R; break;
    default:
        set_error("slamem_index_download: unknown array id %d", which);
        return SLAMEM_ERR_ARG;
    }
    if (count != want) {
        set_error("slamem_index_download: wrong element count for array %d", which);
        return SLAMEM_ERR_ARG;
    }
    DevBuf d;
    if (which == SLAMEM_ARRAY_BWT) {
        SLAMEM_HIP(d.alloc(R));
        hipLaunchKernelGGL(k_bwt_codes, dim3(grid_for(R)), dim3(256), 0, 0, v, d.as<uint8_t>());
        SLAMEM_HIP(hipGetLastError());
        SLAMEM_HIP(hipMemcpy(host_dst, d.p, R, hipMemcpyDeviceToHost));
        return SLAMEM_OK;
    }
    SLAMEM_HIP(d.alloc(want * 4));
    hipLaunchKernelGGL(k_rec_field, dim3(grid_for(want)), dim3(256), 0, 0, v.rec, v.sa, (uint32_t)R, want, field, d.as<uint32_t>());
    SLAMEM_HIP(hipGetLastError());
    SLAMEM_HIP(hipMemcpy(host_dst, d.p, want * 4, hipMemcpyDeviceToHost));
    return SLAMEM_OK;
}

}  // namespace slamem
