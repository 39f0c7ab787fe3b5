// common.h -- internal definitions shared by the HIP translation units of libslamem_hip.so.
// gfx950 (MI355X) only.  Nothing here is part of the public ABI (include/slamem_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/slamem_hip.h"

namespace slamem {

// ---------------------------------------------------------------------------------
// Alphabet (reference: bwtindex.c:39-41,183-196):  $=0 N=1 A=2 C=3 G=4 T=5
// ---------------------------------------------------------------------------------
constexpr uint32_t kArenaMagicLo = 0x4D414C53u;  // "SLAM"
constexpr uint32_t kArenaMagicHi = 0x58494845u;  // "EHIX"
constexpr uint32_t kArenaVersion = 13;  // 13: spill list of the seed table; 12: seed table + text bit-planes (the seed-and-compare path of the search); 11: presence filter in lines keyed by the (k-2)-mer; 10: k-mer occurrence bitmap; 9: K-mer jump table; 8: text-ordered groups + parent records
constexpr uint64_t kHeaderBytes = 4096;
constexpr uint32_t kFmRowsLog2 = 7;  // 128 BWT rows per FM block
constexpr uint32_t kFmRows = 1u << kFmRowsLog2;

// FM block: 128 BWT rows in ONE 64-byte line (one HBM/L2 sector pair per rank query).
//   cnt[c-2] = C[c] + occ(c, rows < 128k) for c in A,C,G,T          (reference a1: letterJumpsSample, bwtindex.c:1454,1481)
//   p0/p1    = two bit-planes of (letter id - 2) for A,C,G,T rows    (reference a1: bwtBits[3], bwtindex.c:33-37)
//   ex       = rows whose letter is N or '$' (their plane bits are 0)
// N is searched through the sorted list of N rows (rare letter), '$' is one known row.
struct __attribute__((aligned(64))) FMBlock {
    uint32_t cnt[4];
    uint64_t p0[2];
    uint64_t p1[2];
    uint64_t ex[2];
};
static_assert(sizeof(FMBlock) == 64, "FM block must be one 64-byte line");

// Per-row record: what the parent-interval operation needs about the interval boundary BETWEEN this row and the next,
// in ONE 16-byte load.  The parent of [t,b] is decided by LCP[t] and LCP[b+1] and reaches from PSV[t] to NSV[b+1]-1:
// record t gives the first pair, record b the second -- for a single row (t = b) that is one record, one line.
//   lcp1  = LCP[row] + 1 (0 = the -1 sentinel of rows 0 and n+1, lcparray.c:624,667),  psv  = PSV[row]
//   lcp1n = LCP[row+1] + 1,                                                             nsvn = NSV[row+1]
//   PSV / NSV = nearest row above / below with a smaller LCP (what the reference's sampled prefix links encode,
//   lcparray.c:782-989).  SA[row] (the reference samples every 32nd row and LF-walks, bwtindex.c:402-420) lives in its
//   own array: only an emitted MEM needs it.
struct __attribute__((aligned(16))) RowRec {
    uint32_t lcp1, psv, lcp1n, nsvn;
};
static_assert(sizeof(RowRec) == 16, "row record must be 16 bytes");

// Text-ordered sections (no reference counterpart): what lets K8 extend a match that has become ONE row -- i.e. one text
// position r = SA[row] -- by comparing the query with the text itself, 16 letters per trip from sequential memory,
// instead of one random FM-block line per letter (the reference walks these letters one FMI_FollowLetter at a time,
// slamem.c:121).
//   TextGroup g: letters of text positions 16g .. 16g+15 as 4-bit ids (first letter in the top nibble, as K1 packs
//                them), and for each of them the CLASS of the parent depth of the position to its right:
//                cls nibble of position s = depth_class(pd[s+1]), pd[x] = max(LCP[ISA[x]], LCP[ISA[x]+1]) = depth of the
//                parent of the single-row interval of the suffix starting at x (lcparray.c:514-518).  While the letters
//                agree nothing can be emitted unless that parent is >= min_len deep, which the class tells (conservatively)
//                without touching the row records.
//   TextRec s:   row = ISA[s] and the parent interval of [row,row] with its depth + 1 -- where a direct run ends, ONE
//                16-byte read gives back the row and (when it ended on a disagreeing letter) the widened interval to retry
//                the letter on, instead of ISA, FM block and row record one after the other.
struct __attribute__((aligned(16))) TextGroup { uint64_t letters, classes; };
struct __attribute__((aligned(16))) TextRec { uint32_t row, ptop, pbot, pdepth1; };
static_assert(sizeof(TextGroup) == 16 && sizeof(TextRec) == 16, "text-ordered records are 16 bytes");
// parent-depth classes: class c = parent depth in [kDepthClass[c], kDepthClass[c+1]);  -1 (root) is class 0
__host__ __device__ inline uint32_t depth_class(int d) {
    const int t[16] = {0, 8, 10, 12, 14, 16, 18, 20, 25, 30, 40, 50, 75, 100, 150, 255};
    uint32_t c = 0;
    for (int k = 1; k < 16; k++) c += d >= t[k];
    return c;
}

// Arena header (first 4 KiB of the index arena; also the on-disk header).
struct ArenaHeader {
    uint32_t magic_lo, magic_hi;
    uint32_t version;
    uint32_t n;           // text length; rows = n + 1
    uint64_t total_bytes;
    uint64_t off_fm;      // FMBlock[nblocks]
    uint64_t off_rec;     // RowRec[n+2]    per-row {LCP+1, PSV, LCP(next)+1, NSV(next)}
    uint64_t off_sa;      // uint32[n+1]    suffix array
    uint64_t off_nrows;   // uint32[num_n]  sorted BWT rows holding N
    uint64_t off_kfilter; // uint64[1 << kfilter_log2]  k-mer presence filter (0 = absent)
    uint64_t off_tgrp;    // TextGroup[(n >> 4) + 2]  text-ordered: 16 letters + 16 parent-depth classes per 16 bytes
    uint64_t off_prec;    // TextRec[n+1]             text-ordered: row and parent interval of the suffix at each position
    uint64_t off_kjump;   // uint2[4^kjump_k]  K-mer jump table: BWT interval of every K-mer (absent: top > bottom)
    uint32_t kfilter_log2;
    uint32_t kfilter_k;
    uint32_t nblocks;
    uint32_t dollar_row;
    uint32_t num_n;
    uint32_t max_lcp;
    uint32_t sort_rounds;
    uint32_t C[6];        // C[c] = number of characters of text+'$' smaller than c
    uint32_t kjump_k;     // K of the jump table (0 = none)
    uint32_t kbits_k;     // k of the occurrence bitmap (0 = none)
    uint32_t kfilter_levels;  // 3 (or 0): (k-2)-, k- and (k+2)-mers; 2: no (k+2)-mers (texts above 2^31 letters)
    uint64_t off_kbits;   // uint64[4^kbits_k / 64]  one bit per k-mer over A,C,G,T: does it occur in the text?
    uint32_t layout;      // 1 = full, 2 = compact (no text-ordered sections, half-size presence filter); 0 in arenas of older builds = full
    uint32_t lcp_ge[10];  // rows whose LCP is >= kLcpGe[i]: how repeat-rich the text is at a given minimum length (all 0 in arenas of older builds: unknown)
    // seed-and-compare sections (version 12; all 0: the index has none)
    uint32_t seed_k;      // letters of a seed (<= 18; 17 and 18: texts of 2^29 letters and more)
    uint32_t seed_log2;   // log2 of the number of buckets of the seed table
    uint64_t off_seed;    // SeedBucket[1 << seed_log2]   every k-mer of the text over A,C,G,T by its canonical form
    uint64_t off_tpl;     // TextPlanes[text_units(n)]    the text in units of 64 letters: two bit-planes, the letter mask, the occurs-once plane
    uint64_t off_tnm;     // 0 (version 12 kept the letter mask here; it lies in the units now)
    uint64_t off_tnb;     // 0 (version 12: a bit per unit "holds a letter that is not A,C,G,T")
    // (version 13)
    uint64_t off_spill;   // uint64[spill_cap]            k-mers 13..28 of the buckets that hold that many (SeedBucket::count)
    uint32_t spill_cap;   // entries the section holds
    uint32_t spill_used;  // entries in use
    uint64_t off_tuq;     // 0 (the occurs-once plane lies in the units)
};
// thresholds of ArenaHeader::lcp_ge
constexpr uint32_t kLcpGe[10] = {18, 20, 25, 30, 40, 50, 75, 100, 150, 255};
static_assert(sizeof(ArenaHeader) <= kHeaderBytes, "header too large");

// ---- seed-and-compare sections (no reference counterpart) ---------------------------------------------------------
// What they replace for reads: the reference finds a MEM by walking the query letter by letter through the index
// (slamem.c:114-199).  A MEM of at least L letters contains a k-letter window that starts at a multiple of s = L - k + 1 of
// the strand; the seed table gives every text position of that window, and the MEM is then the run of agreeing letters
// around it on that diagonal -- found by comparing the strand with the text itself, 64 letters per XOR.
//   TextPlanes u : letters 64u .. 64u+63 of the text: two bit-planes (A,C,G,T = 0..3; bit i of p0 / p1 = low / high bit of
//                  letter 64u+i; letters that are not A,C,G,T and positions behind the text are 0), nm: bit per letter "not one
//                  of A,C,G,T", uq: bit per letter "the k-mer that starts here occurs once in the text" (either orientation; not
//                  its own reverse complement).  32 bytes: what a compare needs of the text under a strand is 128 contiguous bytes
//   SeedBucket   : one 64-byte line: up to 12 text positions whose k-mer hashes here (by k-mer, then ascending), a tag byte each (the low
//                  bits of the hash -- the hash is a bijection of the canonical k-mer, so bucket + tag identify it exactly --
//                  and bit 7: the text holds the reverse complement of the canonical form), and the number of k-mers that
//                  hash here.  More than 12: count = 13 (a strand that meets the bucket is left to the index walk), or, for
//                  up to 28, bit 31 + the number beyond 12 in bits 24-28 + their place in the spill list (in units of four
//                  entries, bits 0-23): entry = position | tag << 32, runs padded to four entries with ~0
struct __attribute__((aligned(32))) TextPlanes { uint64_t p0, p1, nm, uq; };
constexpr uint32_t kSeedSlots = 12;
constexpr uint32_t kSeedSpillMax = 16;            // k-mers of a bucket beyond its slots that the spill list takes
constexpr uint32_t kSeedSpilled = 0x80000000u;    // SeedBucket::count of such a bucket
__host__ __device__ inline uint64_t seed_spill_entries(uint64_t n) { return ((n / 16 + 3) & ~3ull) + 64; }
struct __attribute__((aligned(64))) SeedBucket {
    uint32_t pos[kSeedSlots];
    uint8_t tag[kSeedSlots];
    uint32_t count;
};
static_assert(sizeof(SeedBucket) == 64 && sizeof(TextPlanes) == 32, "seed bucket = one line, text unit = 32 bytes");
__host__ __device__ inline uint64_t text_units(uint64_t n) { return (n + 63) / 64 + 4; }  // (a compare reads four units from the one its diagonal starts in)
// k letters as two k-bit plane fields (bit i = letter i) -> the 2k-bit key
__host__ __device__ inline uint32_t seed_key(uint32_t p0, uint32_t p1, uint32_t k) { return p0 | (p1 << k); }
// the key of the reverse complement: letter i of it is the complement (3 - code: both bits flipped) of letter k-1-i
__host__ __device__ inline uint32_t seed_rev_field(uint32_t f, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (__brev(f) >> (32u - k)) ^ ((1u << k) - 1u);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < k; i++) r |= ((f >> i) & 1u) << (k - 1u - i);
    return r ^ ((1u << k) - 1u);
#endif
}
// a bijection of the 2k-bit keys (odd multipliers and right shifts): the high bits choose the bucket, the low ones are the tag
__host__ __device__ inline uint32_t seed_mix(uint32_t key, uint32_t bits) {
    const uint32_t mask = bits >= 32u ? 0xFFFFFFFFu : (1u << bits) - 1u;
    uint32_t x = key;
    x = (x * 0x9E3779B1u) & mask;
    x ^= x >> (bits >> 1);
    x = (x * 0x85EBCA6Bu) & mask;
    x ^= x >> ((bits >> 1) + 1u);
    x = (x * 0xC2B2AE35u) & mask;
    return x;
}
// the same for keys of 33 to 36 bits (seeds of 17 and 18 letters: texts of 2^29 letters and more)
__host__ __device__ inline uint64_t seed_mix64(uint64_t key, uint32_t bits) {
    const uint64_t mask = (1ull << bits) - 1ull;
    uint64_t x = key;
    x = (x * 0x9E3779B97F4A7C15ull) & mask;
    x ^= x >> (bits >> 1);
    x = (x * 0xC2B2AE3D27D4EB4Full) & mask;
    x ^= x >> ((bits >> 1) + 1u);
    x = (x * 0x165667B19E3779F9ull) & mask;
    return x;
}
constexpr uint32_t kSeedMaxK = 18;
// A window's k letters (two plane fields) -> its bucket, the tag's hash bits, which of the two forms it is (1: the reverse
// complement is the canonical one) and whether it is its own reverse complement.  log2b = log2 of the number of buckets.
__host__ __device__ inline void seed_place(uint32_t f0, uint32_t f1, uint32_t k, uint32_t log2b, uint32_t& bucket, uint32_t& tagbits,
                                           uint32_t& orient, uint32_t& pal) {
    const uint32_t r0 = seed_rev_field(f0, k), r1 = seed_rev_field(f1, k), tb = 2u * k - log2b;
    if (k <= 16u) {
        const uint32_t x = seed_key(f0, f1, k), y = seed_key(r0, r1, k);
        const uint32_t h = seed_mix(x < y ? x : y, 2u * k);
        bucket = h >> tb; tagbits = h & ((1u << tb) - 1u); orient = x > y ? 1u : 0u; pal = x == y ? 1u : 0u;
    } else {
        const uint64_t x = (uint64_t)f0 | ((uint64_t)f1 << k), y = (uint64_t)r0 | ((uint64_t)r1 << k);
        const uint64_t h = seed_mix64(x < y ? x : y, 2u * k);
        bucket = (uint32_t)(h >> tb); tagbits = (uint32_t)h & ((1u << tb) - 1u); orient = x > y ? 1u : 0u; pal = x == y ? 1u : 0u;
    }
}

// What the kernels see (passed by value).
struct IndexView {
    const FMBlock* fm;
    const RowRec* rec;
    const uint32_t* sa;
    const uint32_t* nrows;
    const uint64_t* kfilter;  // nullptr when the index has no presence filter
    const TextGroup* tgrp;    // nullptr when the index has no text-ordered sections
    const TextRec* prec;
    const uint2* kjump;       // nullptr when the index has no K-mer jump table
    uint32_t kjump_k;
    uint32_t kbits_k;
    const uint64_t* kbits;    // nullptr when the index has no k-mer occurrence bitmap
    uint32_t n;
    uint32_t nblocks;
    uint32_t dollar_row;
    uint32_t num_n;
    uint32_t kfilter_log2;
    uint32_t kfilter_k;
    uint32_t kfilter_levels;  // 2 or 3
    const SeedBucket* seed;   // nullptr when the index has no seed sections
    const TextPlanes* tpl;
    const uint64_t* spill;
    uint32_t seed_k;
    uint32_t seed_log2;
};

// Presence filter: a blocked Bloom filter in 64-byte lines of eight words.  The line is chosen by a (k-2)-mer of the
// text; it holds that (k-2)-mer and every k-mer and (k+2)-mer of the text that contains it (three and five of them),
// each as three bits of one word.  A cascade of tests (k-2 letters, then the k-mers around them, then the (k+2)-mers
// around those) therefore reads ONE line from HBM: the confirmations hit the cache.
__host__ __device__ inline uint64_t kfilter_hash(uint64_t kmer) {
    kmer ^= kmer >> 33;
    kmer *= 0xff51afd7ed558ccdull;
    kmer ^= kmer >> 33;
    kmer *= 0xc4ceb9fe1a85ec53ull;
    kmer ^= kmer >> 33;
    return kmer;
}
// the filter holds three k-mer lengths, k-2, k and k+2 (the last only while k+2 <= 32); the shorter and the longer
// ones are hashed with these salts
constexpr uint64_t kFilterShortSalt = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kFilterLongSalt = 0xD6E8FEB86659FD93ull;
__host__ __device__ inline uint64_t kfilter_bits(uint64_t h) {
    return (1ull << ((h >> 46) & 63u)) | (1ull << ((h >> 52) & 63u)) | (1ull << ((h >> 58) & 63u));
}
__host__ __device__ inline uint64_t kfilter_word(uint64_t h) { return (h >> 43) & 7ull; }  // word of the line
// first word of the line of a (k-2)-mer whose salted hash is h (log2_words: 64-bit words of the whole filter)
__host__ __device__ inline uint64_t kfilter_line(uint64_t h, uint32_t log2_words) {
    return (h & ((1ull << (log2_words - 3u)) - 1ull)) << 3;
}

// Raw record written by the search kernel before the per-block compaction (K9).
struct RawKey { uint32_t block; uint32_t k; };

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SLAMEM_HIP(call)                                                         \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) return ::slamem::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

struct Timings {
    slamem_timings t;
};
Timings& thread_timings();
bool search_stats_wanted();
slamem_search_stats& last_search_stats();
double* last_search_clock();  // diagnostics: {us until the work list was empty, us of tail after that, sum of wave run times}

}  // namespace slamem

// The opaque handle of the C ABI.
namespace slamem {
// The note a batch leaves in the index handle for the next one (slamem_index::seed_words_hint; mem_search.hip, collect()):
// `hint` the note as it was (0, 4 or 6 plane words), `words` the form the batch ran, `words_avg` the form its average read
// length alone asks for; over the sampled reads: c4 / c6 reads left to the index walk only for their length that four / six
// words would hold, `needed` some read was longer than the next narrower form holds.  More than an eighth of the sample left
// for their length: the form that holds them; a form wider than the average asks for, taken from the note, that no read
// needed: one step back.
inline uint32_t seed_words_next(uint32_t hint, uint32_t words, uint32_t words_avg, uint64_t c4, uint64_t c6, uint64_t sampled,
                                bool needed) {
    if (c6 * 8u > sampled) return 6u;
    if ((c4 + c6) * 8u > sampled) return hint > 4u ? hint : 4u;
    if (words > words_avg && words == hint && !needed) return hint == 6u ? 4u : 0u;
    return hint;
}
}  // namespace slamem

struct slamem_index {
    slamem::ArenaHeader hdr;   // host copy
    void* arena;               // device
    uint64_t arena_bytes;
    int device;
    int owns_arena;
    slamem::IndexView view;
    // what the last batches searched against this index said about their reads (mem_search.hip: the seed kernel's form is chosen
    // by a batch's average read length; a batch with many reads longer than that form holds raises this, so that the next
    // batch takes the wider form): 0, 4 or 6 plane words.  Relaxed atomic access; decides speed only, never an answer.
    uint32_t seed_words_hint;
};

namespace slamem {
void make_view(slamem_index* idx);
int build_index_device(const void* text_dev, uint32_t n, int device, hipStream_t stream, int layout, slamem_index** out);
int estimate_build_bytes(uint32_t n, int layout, uint64_t* arena_bytes, uint64_t* peak_bytes);
int find_mems_device(const slamem_index* idx, const void* queries_dev, const uint64_t* offsets_dev,
                     uint32_t num_queries, uint64_t query_bytes, uint32_t min_len, int both_strands, int match_type,
                     slamem_mem* mems_dev, uint64_t mems_capacity, uint64_t* block_offsets_dev, void* workspace_dev,
                     uint64_t workspace_bytes, hipStream_t stream, uint64_t* total_out);
uint64_t find_mems_workspace_bytes(uint64_t num_queries, int both_strands, uint64_t query_bytes, uint64_t mems_capacity);
// One batch through the search in steps that may be issued apart and on different streams (mem_search.hip; used by stream.hip):
// tables (one small sync) -> prep (K8a, work list, K7q; asynchronous) -> search (K8, K9, scalars to host_scalars; asynchronous)
// -> collect (after the search stream has finished the batch: totals, capacity check, timings of the calling thread).
// host_scalars: nine 64-bit words, pinned if possible (nullptr: the job's own).
struct SearchJob;
SearchJob* search_job_new();
void search_job_delete(SearchJob* j);
int search_job_init(SearchJob* j, const slamem_index* idx, const void* queries_dev, const uint64_t* offsets_dev, uint32_t num_queries,
                    uint64_t query_bytes, uint32_t min_len, int both_strands, int match_type, slamem_mem* mems_dev,
                    uint64_t mems_capacity, uint64_t* block_offsets_dev, void* workspace_dev, uint64_t workspace_bytes,
                    unsigned long long* host_scalars);
// (between init and tables) the number of slices of the batch when the caller knows it -- no record longer than a slice: one
// per record -- which saves tables() its host round trip
void search_job_slices_hint(SearchJob* j, uint32_t slices);
// (between init and the search) at most this many waves for the batch's K8 (0: the whole chip)
void search_job_k8_wave_cap(SearchJob* j, uint32_t waves);
constexpr uint32_t kSearchSliceLen = 4096;
int search_job_tables(SearchJob* j, hipStream_t stream);
int search_job_prep(SearchJob* j, hipStream_t stream);
int search_job_search(SearchJob* j, hipStream_t stream);   // = search_job_k8(j, stream, nullptr, 0) + search_job_place
// K8 without its tail: search_job_k8(..., carry_out = 1) ends when the batch's work list is empty and passes
// its unfinished lanes on; the next batch's search_job_k8(next, stream, j, ...) -- or search_job_flush(j) when none follows --
// finishes them; search_job_place (K9 + scalars) of a batch comes after that.
int search_job_can_carry_into(const SearchJob* j, const SearchJob* next);
int search_job_k8(SearchJob* j, hipStream_t stream, SearchJob* carry_from, int carry_out);
int search_job_carried_out(const SearchJob* j);
int search_job_flush(SearchJob* j, hipStream_t stream);
int search_job_place(SearchJob* j, hipStream_t stream);
int search_job_collect(SearchJob* j, uint64_t* total_out);
}  // namespace slamem
