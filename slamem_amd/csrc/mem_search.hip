// mem_search.hip -- north_star (b): per-query backward search + parent-interval MEM extension.
//
// Replaces the hot body of GetMatches (slamem.c:105-199) and every index query it calls:
//   FMI_FollowLetter          bwtindex.c:359-400   -> follow()
//   FMI_LetterJump/BitsSetCount bwtindex.c:315-356 -> occ_lt()  (hardware popcount on a 128-row block)
//   FMI_GetCharAtBWTPos       bwtindex.c:304-313   -> bwt_code()
//   FMI_PositionInText        bwtindex.c:402-420   -> one read of the full suffix array
//   GetEnclosingLCPInterval   lcparray.c:330-423   -> parent()  (semantics of lcparray.c:514-523)
//   GetLcpPosFromBwtPos / GetLcpValueFromLcpPos / GetBwtPosFromLcpPos / GetPrefixLinkFromLcpPos / IsTopCorner
//                             lcparray.c:119-328   -> subsumed: LCP, both links and SA are stored for EVERY row in one
//                                                     16-byte record (one round trip per parent call instead of
//                                                     the 4-6 dependent probes of the sampled structure)
//   ReverseComplementSequence sequence.c:413-430   -> folded into the query fetch of the reverse-strand lane
//
// Work mapping: one lane per (query record, strand) -- the scan of one strand is a chain of ~2 dependent
// random reads per base, so throughput comes from occupancy (thousands of independent chains per CU), not
// from splitting a chain.  MEMs are appended to a raw list through one wave-aggregated atomic per emitting
// wave-instruction and put into (block, emission order) by K9.
#include "common.h"
#include "prims.h"

#include <atomic>
#include <type_traits>
#include <new>
#include <stdlib.h>
#include <string.h>

namespace slamem {

// ------------------------------------------------------------------------------------------
// device-side index queries
// ------------------------------------------------------------------------------------------
struct Blk {
    uint4 a;  // cnt[0..3]
    uint4 b;  // p0[0], p0[1]
    uint4 c;  // p1[0], p1[1]
    uint4 d;  // ex[0], ex[1]
};

__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

__device__ __forceinline__ Blk load_blk(const FMBlock* __restrict__ fm, uint32_t bi) {
    const uint4* p = reinterpret_cast<const uint4*>(fm + bi);
    Blk k;
    k.a = p[0]; k.b = p[1]; k.c = p[2]; k.d = p[3];
    return k;
}

// C[c] + occ(c, rows < off of this block), c2 = letter id - 2 in 0..3, off in 0..127
__device__ __forceinline__ uint32_t occ_lt(const Blk& k, uint32_t c2, uint32_t off) {
    uint64_t p00 = u64_of(k.b.x, k.b.y), p01 = u64_of(k.b.z, k.b.w);
    uint64_t p10 = u64_of(k.c.x, k.c.y), p11 = u64_of(k.c.z, k.c.w);
    uint64_t e0 = u64_of(k.d.x, k.d.y), e1 = u64_of(k.d.z, k.d.w);
    uint64_t f0 = (c2 & 1u) ? ~0ull : 0ull, f1 = (c2 & 2u) ? ~0ull : 0ull;
    uint64_t m0 = ~(p00 ^ f0) & ~(p10 ^ f1) & ~e0;
    uint64_t m1 = ~(p01 ^ f0) & ~(p11 ^ f1) & ~e1;
    uint32_t cnt = c2 == 0 ? k.a.x : c2 == 1 ? k.a.y : c2 == 2 ? k.a.z : k.a.w;
    uint32_t lo = off < 64u ? off : 64u, hi = off < 64u ? 0u : off - 64u;
    uint64_t mlo = lo == 64u ? ~0ull : ((1ull << lo) - 1ull);
    uint64_t mhi = (1ull << hi) - 1ull;  // hi <= 63
    return cnt + (uint32_t)__popcll(m0 & mlo) + (uint32_t)__popcll(m1 & mhi);
}

// set bits of the 128-bit mask m1:m0 below bit `off` (off in 0..127)
__device__ __forceinline__ uint32_t count_lt(uint64_t m0, uint64_t m1, uint32_t off) {
    uint32_t lo = off < 64u ? off : 64u, hi = off < 64u ? 0u : off - 64u;
    uint64_t mlo = lo == 64u ? ~0ull : ((1ull << lo) - 1ull);
    uint64_t mhi = (1ull << hi) - 1ull;  // hi <= 63
    return (uint32_t)__popcll(m0 & mlo) + (uint32_t)__popcll(m1 & mhi);
}

// number of N rows strictly below `row`
__device__ __forceinline__ uint32_t n_rows_lt(const IndexView& ix, uint32_t row) {
    uint32_t lo = 0, hi = ix.num_n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (ix.nrows[mid] < row) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// FMI_GetCharAtBWTPos as a letter id
__device__ __forceinline__ uint32_t bwt_code(const IndexView& ix, uint32_t row) {
    const FMBlock* b = ix.fm + (row >> kFmRowsLog2);
    uint32_t o = row & (kFmRows - 1u), hs = o >> 6, bit = o & 63u;
    uint64_t e = b->ex[hs];
    if ((e >> bit) & 1ull) return row == ix.dollar_row ? 0u : 1u;
    return 2u + (uint32_t)((b->p0[hs] >> bit) & 1ull) + 2u * (uint32_t)((b->p1[hs] >> bit) & 1ull);
}

// The FM blocks of rows `top` and `bot+1`, kept in registers across retries: after a parent step the widened
// interval usually still starts / ends in the same 128-row block, so the retry costs no memory access.
struct BlkCache {
    Blk t, b;
    uint32_t it, ib;  // block indices held (0xFFFFFFFF = none)
};

__device__ __forceinline__ void cache_blocks(const IndexView& ix, BlkCache& bc, uint32_t top, uint32_t bot) {
    uint32_t bt = top >> kFmRowsLog2, bb = (bot + 1u) >> kFmRowsLog2;
    if (bt != bc.it) {
        if (bt == bc.ib) bc.t = bc.b; else bc.t = load_blk(ix.fm, bt);
        bc.it = bt;
    }
    if (bb != bc.ib) {
        if (bb == bc.it) bc.b = bc.t; else bc.b = load_blk(ix.fm, bb);
        bc.ib = bb;
    }
}

// The two rank queries of FMI_FollowLetter for letter id c on [top,bot] (blocks already in `bc`):
//   nt  = C[c] + occ(c, rows < top)        new top
//   nb1 = C[c] + occ(c, rows <= bot)       one past the new bottom
// so nb1 - nt = number of rows of [top,bot] whose BWT letter is c (0: the extension does not occur).
__device__ __forceinline__ void occ_pair(const IndexView& ix, const BlkCache& bc, uint32_t c, uint32_t top,
                                         uint32_t bot, uint32_t& nt, uint32_t& nb1) {
    if (c >= 2u) {
        nt = occ_lt(bc.t, c - 2u, top & (kFmRows - 1u));
        nb1 = occ_lt(bc.b, c - 2u, (bot + 1u) & (kFmRows - 1u));
    } else if (ix.num_n == 0) {
        nt = nb1 = 1u;
    } else {  // N: C[N] = 1 (only '$' is smaller); rank through the sorted list of N rows
        nt = 1u + n_rows_lt(ix, top);
        nb1 = 1u + n_rows_lt(ix, bot + 1u);
    }
}

// FMI_FollowLetter: returns true and updates [top,bot] when the extended string occurs
__device__ __forceinline__ bool follow(const IndexView& ix, uint32_t c, uint32_t& top, uint32_t& bot) {
    BlkCache bc;
    bc.it = bc.ib = 0xFFFFFFFFu;
    cache_blocks(ix, bc, top, bot);
    uint32_t nt, nb1;
    occ_pair(ix, bc, c, top, bot, nt, nb1);
    if (nt >= nb1) return false;
    top = nt;
    bot = nb1 - 1u;
    return true;
}

// GetEnclosingLCPInterval: parent LCP-interval of [top,bot]; returns its depth, -1 at the root.
// rt = record of row top, rb = record of row bot (the same record for a single row).
__device__ __forceinline__ int parent_from(const uint4& rt, const uint4& rb, uint32_t& top, uint32_t& bot) {
    uint32_t a = rt.x, b = rb.z;   // LCP[top] + 1, LCP[bot+1] + 1
    uint32_t d = a > b ? a : b;
    if (d == 0u) return -1;
    if (a == d) top = rt.y;        // closest row above with a smaller LCP   (lcparray.c:519)
    if (b == d) bot = rb.w - 1u;   // closest row below with a smaller LCP   (lcparray.c:520-521)
    return (int)(d - 1u);
}

__device__ __forceinline__ int parent(const IndexView& ix, uint32_t& top, uint32_t& bot) {
    const uint4* R = reinterpret_cast<const uint4*>(ix.rec);
    uint4 rt = R[top], rb = R[bot];  // both boundary records: one round trip (one record for a single row)
    return parent_from(rt, rb, top, bot);
}

__device__ __forceinline__ uint32_t ascii_code_q(uint32_t ch) {
    uint32_t x = ch & 0xDFu;
    return x == 'A' ? 2u : x == 'C' ? 3u : x == 'G' ? 4u : x == 'T' ? 5u : 1u;
}

// ------------------------------------------------------------------------------------------
// K8: the MEM search kernel
// ------------------------------------------------------------------------------------------
struct SearchArgs {
    IndexView ix;
    const uint64_t* qwords;    // query characters viewed as 8-byte words
    const uint64_t* offsets;   // [num_queries+1]
    uint32_t num_queries;
    uint32_t strands;          // 1 or 2
    uint32_t min_len;
    int32_t pad0;
    uint32_t pad1;
    uint32_t pad;
    uint64_t capacity;         // raw records that fit
    unsigned long long* total; // running number of MEMs
    RawKey* raw_key;
    slamem_mem* raw_mem;
    uint32_t* block_counts;    // [num_blocks + 1]
    struct RawRow* inline_rows; // v3: kInlineMems slots per work item, addressed directly (no atomics): slot kk of item g at
                                //     [kk * inline_stride + g] (slot-major: K9 reads every item's first slot, rarely another)
    uint64_t inline_stride;
    const struct ItemDesc* items; // v3: work items in emission order
    uint64_t num_items;
    uint8_t* item_attempt;      // v3: attempt whose records are the valid ones
    const uint8_t* item_alive;  // v3: nullptr, or 0 for items the k-mer presence filter proved empty
    const uint32_t* work_ids;   // v3: nullptr (work = all items), or the ids of the items that survived the prefilter
    const uint32_t* work_count; // v3: device word holding their number
    unsigned long long* stats;  // diagnostic instantiations only: SC_COUNT counters
    uint64_t query_words;       // v3: 8-byte words of the query buffer that may be read
    const uint64_t* pq;         // v3: the strands as packed letter ids (k_pack_queries), two zero words in front
    uint64_t* pq_out;           //     (the same buffer, written by k_pack_queries)
    uint32_t implicit_items;    // v3: 1: no item tables (item_of / item_pk_of)
    uint32_t* seed_left_ids;    // K8s: the strands it leaves to K8 (K8's work list) ...
    unsigned int* seed_left_count;  // ... and their number
    unsigned int* seed_long_flag;   // K8s sets it when a record is longer than a slice (a call that took the seed path without asking starts again)
    unsigned int* seed_wide;        // K8s, over a sample of the reads (256 to 511 blocks of the grid): [0] reads left to K8 for their length that the four-word form holds, [1] that the six-word form holds, [2] set when a read needed this launch's form (longer than the next narrower one holds), [3] reads in the sample
    uint32_t seed_step;         // K8s: 0, or the stride of the windows of the first round (experiments: SLAMEM_SEED_STEP)
    const uint64_t* item_pk;    // v3: per work item, the word offset of its strand block in pq
    unsigned int* work_cursor;  // v3: next unassigned position of the work list (zeroed per launch)
    int32_t direct_min_depth;   // v3: a single-row match at least this deep is extended by comparing with the text (<0: off)
    uint32_t use_jump;          // v3: take the first K letters of a scan through the K-mer jump table
    uint32_t pad2;
    uint32_t skip_w;            // v3: letters verified on the diagonal behind a disagreeing letter (min_len - 1); 0 = no skipping
    const uint64_t* pq2;        // v3: the strands at 2 bits per letter (32 letters per word), at half the offsets of pq
    uint64_t* pq2_out;
    uint8_t* item_flags;        // v3: 1 = the strand holds a letter that is not A,C,G,T (or is a slice): no skipping
    uint32_t skip_s1;           // v3: stride of the probed k-mers, min_len - k + 1
    uint32_t pad4;
    const struct SliceState* slice_state;  // v3: start states of the slices of long records (k_slice_states), or nullptr
    const uint32_t* item_block;            //     strand block number of every item (with slice_state)
    // kCarry (slamem_stream_*): K8 without its tail.  A launch that is followed by the next batch's does not wait for the last
    // strands of its lanes: when the work list is empty every lane writes its state to carry_out and the kernel ends; the next
    // launch takes those lanes in first (carry_in) and lets them write to the PREVIOUS batch's output side (prev).
    struct OutCtx {
        struct RawRow* inline_rows;
        uint64_t inline_stride;
        RawKey* raw_key;
        slamem_mem* raw_mem;
        unsigned long long* total;
        uint64_t capacity;
        uint32_t* block_counts;
        uint8_t* item_attempt;
    } prev;
    // kDefer (repeat-rich texts, reads): enumeration jobs leave the strand's chain of dependent steps.  The lane that meets an
    // interval of several rows / an ancestor still >= min_len deep puts the job into a queue and walks on; k_enum_jobs runs the
    // queue with every wave of the chip; K9 gives the MEMs their places from (records before the job, MEMs of the jobs before).
    struct DeferCtx {
        struct JobRec* queue;      // [pool_cap]
        unsigned int* njobs;
        uint32_t* pool;            // per deferring strand: {jobs, MEMs the lane reported itself, MEMs of job 0, 1, ...} -> prefix sums (K9)
        unsigned int* pool_next;
        uint32_t pool_cap;
        uint32_t queue_cap;        // places of the queue: pool_cap (a job per counter at most) + a chunk per wave
        uint32_t epoch;            // stamp of this launch in the queue's records: the waves reserve queue places in chunks, what they leave unused holds no (or an older) stamp
        uint32_t* raw_seg;         // beside raw_key / raw_mem: where in `pool` the record's offset is (0xFFFFFFFF: its ordinal is final)
        uint2* list;               // {strand block, its place in the pool}
        unsigned int* nlist;
        uint8_t* inline_valid;     // per item: 0, or 1 + the inline slots that hold MEMs (those reported before the first job)
    } defer;
    const struct CarryRec* carry_in;       // lanes of the previous launch that were not finished (nullptr: none)
    const unsigned int* carry_in_count;
    struct CarryRec* carry_out;            // nullptr: run every strand to its end (a stand-alone launch, or the last of a stream)
    unsigned int* carry_out_count;
};

// v3 raw record: the BWT row is resolved to SA[row] by K9, so the search kernel never waits for a locate.
struct RawRow { uint32_t row, pos, len; };
constexpr uint32_t kInlineMems = 4;     // MEMs per work item stored in place; more go to the overflow list
constexpr uint32_t kFetch = 64;         // v3: work items a wave takes from the global cursor at a time

// Work item of K8 v3 = one strand of one query record, or -- for long records (genome against genome) -- one
// slice of kSliceLen positions of it.  The strand's rightmost slice is scanned from the record's end; every other slice
// [a,b) starts at b from the state the full scan has there, which k_slice_states works out beforehand from a warm-up of
// kWarmUp positions (see there; -mam: k_find_mams_sliced guesses and verifies instead).  (K8 still carries the older
// scheme -- scan from b + warm-up, restart with four times the warm-up when a match of the slice reaches the scan's
// start, attempt tag in the records -- for a slice that comes without a state.)
constexpr uint32_t kSliceLen = 4096;
static_assert(kSliceLen == kSearchSliceLen, "stream.hip tells the slice count of a batch from this length");
constexpr uint32_t kWarmUp = 1024;
constexpr uint32_t kMaxAttempt = 7;
struct __attribute__((aligned(16))) ItemDesc {
    uint64_t base;       // byte offset of the query record
    uint32_t len;        // its length
    uint32_t slice_rev;  // slice index | reverse strand << 31
};
// Implicit items (SearchArgs::implicit_items; a batch of reads on the seed path: no record is cut into slices, item g IS strand
// block g): the descriptor and the strand's place in the packed copy come from the offsets -- the two tables of 24 bytes per
// strand that k_item_fill writes (and the scan behind the second) are not made for the few strands K8s leaves.  The place:
// read q's blocks start at 2 + strands * 2 * (offset within the batch / 32 + q) words -- at or behind where the blocks before it end
// (a strand takes 2 * ceil(len / 32) words), within the pq_bytes the work space reserves.
__device__ __forceinline__ ItemDesc item_of(const SearchArgs& A, uint64_t it) {
    if (!A.implicit_items) return A.items[it];
    const uint64_t q = A.strands == 2u ? it >> 1 : it;
    const uint64_t o0 = A.offsets[q];
    return ItemDesc{o0, (uint32_t)(A.offsets[q + 1] - o0), A.strands == 2u ? (uint32_t)(it & 1ull) << 31 : 0u};
}
__device__ __forceinline__ uint64_t item_pk_of(const SearchArgs& A, uint64_t it, const ItemDesc& d) {
    if (!A.implicit_items) return A.item_pk[it];
    const uint64_t q = A.strands == 2u ? it >> 1 : it;
    const uint64_t rel = d.base - A.offsets[0];  // (a caller's offsets may start anywhere: slamem_stream_* hands over windows of one array)
    return 2ull + (uint64_t)A.strands * 2ull * ((rel >> 5) + q) + (uint64_t)(d.slice_rev >> 31) * 2ull * ((d.len + 31u) >> 5);
}

// Letters of one strand of a query record, read through a window of kBytes (16 or 32) held in registers: aligned
// 16-byte loads, 5-10 per strand of 150 letters.  (With 8-byte words, a word per 8 letters, the line was fetched
// again for almost every word: the random index traffic evicts it from L1 and L2 between two uses.)
// The buffer is 16-byte aligned (checked by the host side); a half window is read only if it overlaps the record.
template <uint32_t kBytes>
struct QueryCursorT {
    const uint64_t* words;  // the query buffer
    uint64_t base;          // byte offset of the record
    uint32_t len;
    uint32_t rev;           // reverse-complement view
    uint64_t cidx;          // offset of the window held, ~0 = none
    uint64_t q0, q1, q2, q3;
    __device__ __forceinline__ void init(const uint64_t* q, uint64_t b, uint32_t l, uint32_t r) {
        words = q; base = b; len = l; rev = r; cidx = ~0ull;
        q0 = q1 = q2 = q3 = 0;
    }
    __device__ __forceinline__ void forget() { cidx = ~0ull; }
    __device__ __forceinline__ uint32_t would_load(uint32_t j) const {  // diagnostics: does at(j) fetch a new window?
        uint64_t addr = base + (rev ? (uint64_t)(len - 1u - j) : (uint64_t)j);
        return (addr & ~(uint64_t)(kBytes - 1u)) != cidx ? 1u : 0u;
    }
    // letter id of position j of the scanned strand
    __device__ __forceinline__ uint32_t at(uint32_t j) {
        uint64_t addr = base + (rev ? (uint64_t)(len - 1u - j) : (uint64_t)j);  // byte offset in the buffer
        uint64_t chunk = addr & ~(uint64_t)(kBytes - 1u);
        if (chunk != cidx) {
            const uint4* p = reinterpret_cast<const uint4*>(words) + (chunk >> 4);
            uint4 a = p[0];  // holds the letter, or (32-byte window) at least overlaps the record -- see below
            q0 = u64_of(a.x, a.y); q1 = u64_of(a.z, a.w);
            if (kBytes == 32u) {
                // only halves that overlap the record's characters are read: the buffer may end right behind it
                uint64_t lo = base, hi = base + len;  // record = [lo, hi)
                bool h0 = chunk + 16 > lo && chunk < hi, h1 = chunk + 32 > lo && chunk + 16 < hi;
                uint4 b = make_uint4(0, 0, 0, 0);
                if (!h0) { q0 = 0; q1 = 0; }
                if (h1) b = p[1];
                q2 = u64_of(b.x, b.y); q3 = u64_of(b.z, b.w);
            }
            cidx = chunk;
        }
        uint32_t o = (uint32_t)(addr & (uint64_t)(kBytes - 1u));
        // (picked with masks, not with ?: on the members: the optimiser turns a conditional over adjacent members into an
        //  indexed load, and an indexed load keeps the whole cursor in scratch memory)
        uint64_t m8 = (o & 8u) ? ~0ull : 0ull;
        uint64_t q = q0 ^ ((q0 ^ q1) & m8);
        if (kBytes == 32u) {
            uint64_t m16 = (o & 16u) ? ~0ull : 0ull, qb = q2 ^ ((q2 ^ q3) & m8);
            q = q ^ ((q ^ qb) & m16);
        }
        uint32_t c = ascii_code_q((uint32_t)(q >> ((o & 7u) * 8u)) & 0xFFu);
        return (rev && c >= 2u) ? 7u - c : c;  // A<->T, C<->G; N stays N  (sequence.c:419-426)
    }
};
typedef QueryCursorT<32> QueryCursor;  // the search kernels (a 16-byte window saves 8 VGPRs and costs twice the loads)

// 16-byte chunks of the query buffer whose letters one block of K8a packs into LDS (2 bits per letter + one "not
// A,C,G,T" bit) when the strands of its 256 items lie in a span this short (128 reads of up to 320 letters).  The windows
// of both strands of a read are then cut out of the packed copy with a few shifts -- no per-letter loop (it cost ~80
// instructions per letter: K8a was bound by instruction issue as much as by requests), and the reads' own lines are
// fetched once, coalesced, instead of twice through probes that evict them from L2.  0 = never
#ifndef SLAMEM_PF_STAGE_CHUNKS
#define SLAMEM_PF_STAGE_CHUNKS 2560
#endif
constexpr uint32_t kPfStageChunks = SLAMEM_PF_STAGE_CHUNKS;

// Letters of a strand in ascending position order: the scan of K8a for the strands that are NOT in a packed span (long
// records, slices).  Eight bytes in a register are consumed with a shift, sixteen more wait in registers, one aligned
// 16-byte load per 16 letters.
struct QueryStream {
    const uint4* p;          // next 16-byte chunk (forward strand: ascending addresses, reverse strand: descending)
    uint64_t cur, n1, n2, n3;  // bytes being consumed / the 8-byte pieces that follow, in consumption order
    uint32_t left;           // bytes of cur not consumed yet
    uint32_t nq;             // pieces waiting in n1..n3
    uint32_t chunks;         // 16-byte chunks not loaded yet that still overlap the record
    uint32_t rev;
    uint32_t loads;          // 16-byte loads issued (read by the diagnostic instantiation only; dead code otherwise)
    __device__ __forceinline__ uint4 chunk(int off) const { return p[off]; }
    __device__ __forceinline__ void advance(int by) { p += by; }
    __device__ __forceinline__ void init(const uint64_t* words, uint64_t base, uint32_t len, uint32_t r, uint32_t start) {
        rev = r;
        loads = 1;
        uint64_t addr = base + (r ? (uint64_t)(len - 1u - start) : (uint64_t)start);  // byte offset of the first letter
        p = reinterpret_cast<const uint4*>(words) + (addr >> 4);
        chunks = r ? (uint32_t)((addr >> 4) - (base >> 4)) : (uint32_t)(((base + len - 1u) >> 4) - (addr >> 4));
        uint4 a = chunk(0);
        uint64_t lo = u64_of(a.x, a.y), hi = u64_of(a.z, a.w);
        uint32_t o = (uint32_t)(addr & 15ull);
        n1 = n2 = n3 = 0;
        if (!r) {
            advance(1);
            if (o < 8u) { cur = lo >> (8u * o); left = 8u - o; n1 = hi; nq = 1; }
            else { cur = hi >> (8u * (o - 8u)); left = 16u - o; nq = 0; }
        } else {  // the letters come from descending addresses: take them from the top of the register
            advance(-1);
            if (o >= 8u) { cur = hi << (8u * (15u - o)); left = o - 7u; n1 = lo; nq = 1; }
            else { cur = lo << (8u * (7u - o)); left = o + 1u; nq = 0; }
        }
    }
    // letter id of the next position (the caller never asks for more letters than the strand has)
    __device__ __forceinline__ uint32_t next() {
        if (left == 0u) {
            if (nq) { cur = n1; n1 = n2; n2 = n3; nq--; }
            else {
                // 32 bytes at a time when two more chunks overlap the record (a chunk that does not may lie behind the buffer)
                uint4 a = chunk(0), b = make_uint4(0, 0, 0, 0);
                const bool two = chunks >= 2u;
                if (two) b = rev ? chunk(-1) : chunk(1);
                uint64_t alo = u64_of(a.x, a.y), ahi = u64_of(a.z, a.w), blo = u64_of(b.x, b.y), bhi = u64_of(b.z, b.w);
                if (!rev) { cur = alo; n1 = ahi; n2 = blo; n3 = bhi; advance(two ? 2 : 1); }
                else { cur = ahi; n1 = alo; n2 = bhi; n3 = blo; advance(two ? -2 : -1); }
                nq = two ? 3u : 1u;
                chunks -= two ? 2u : 1u;
                loads += two ? 2u : 1u;
            }
            left = 8u;
        }
        left--;
        uint32_t b;
        if (!rev) { b = (uint32_t)cur & 0xFFu; cur >>= 8; } else { b = (uint32_t)(cur >> 56); cur <<= 8; }
        uint32_t c = ascii_code_q(b);
        return (rev && c >= 2u) ? 7u - c : c;  // A<->T, C<->G; N stays N  (sequence.c:419-426)
    }
};

__device__ __forceinline__ void emit3_at(const SearchArgs& A, uint32_t g, uint32_t kk, uint32_t tag, uint32_t row,
                                         uint32_t pos, uint32_t len);
// the lane-level emission of the -mam kernels: through the v3 output path (the BWT row is stored, K9 resolves SA[row];
// no returned atomic for the first MEMs of an item)
__device__ __forceinline__ void emit(const SearchArgs& A, uint32_t blockid, uint32_t& k, uint32_t row, uint32_t j,
                                     uint32_t len, uint32_t tag = 0u) {
    emit3_at(A, blockid, k, tag, row, j, len);
    k++;
}

// Rows of the pending interval [top,bot] (match of length `depth` starting at query position `pos`) and of
// every ancestor interval still >= min_len deep, filtered for left-maximality (slamem.c:139-193).
//   same_left = number of rows of [top,bot] whose BWT letter equals the query's next letter to the left
//               (it falls out of the rank queries of the NEXT backward step, so the common single-row case
//               needs no extra BWT access: the reference reads FMI_GetCharAtBWTPos per row, slamem.c:141,166)
//   pub       = upper bound on the depth of the parent of [top,bot]; tightened to the exact value when read.
//               While pub < min_len no ancestor can qualify and GetEnclosingLCPInterval (slamem.c:192) is skipped.
__device__ __forceinline__ void emit_levels(const SearchArgs& A, uint32_t blockid, uint32_t& k, uint32_t top,
                                            uint32_t bot, int depth, uint32_t pos, uint32_t left, uint32_t same_left,
                                            int& pub, uint32_t tag = 0u) {
    const IndexView& ix = A.ix;
    const int L = (int)A.min_len;
    uint32_t size = bot - top + 1u;
    if (same_left < size) {
        if (size == 1u) emit(A, blockid, k, top, pos, (uint32_t)depth, tag);
        else
            for (uint32_t row = top; row <= bot; row++)  // slamem.c:140
                if (bwt_code(ix, row) != left) emit(A, blockid, k, row, pos, (uint32_t)depth, tag);
    }
    if (pub < L) return;
    uint32_t t = top, b = bot, pt = top, pb = bot;
    int msz = parent(ix, t, b);
    pub = msz;
    while (msz >= L) {
        for (uint32_t row = t; row != pt; row++)  // new rows above (slamem.c:140)
            if (bwt_code(ix, row) != left) emit(A, blockid, k, row, pos, (uint32_t)msz, tag);
        for (uint32_t row = b; row != pb; row--)  // new rows below, bottom-up (slamem.c:165)
            if (bwt_code(ix, row) != left) emit(A, blockid, k, row, pos, (uint32_t)msz, tag);
        pt = t;
        pb = b;
        msz = parent(ix, t, b);  // slamem.c:192
    }
}

// -mam (slamem.c:131,657): the reference's scan with its MAM test -- a position whose interval is not a single row is
// skipped by a `continue` that also skips the bookkeeping of slamem.c:197-198, so the interval that a later failed
// extension falls back to (:122-123) is the one saved at an EARLIER position.  Output is defined by that
// behaviour, so the scan is kept in the reference's shape: emission at the position itself, explicit fall-back
// interval; one lane per strand, whole strands (the stale interval makes slices of a strand depend on their past).
__global__ void __launch_bounds__(256) k_find_mams(SearchArgs A) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t nblocks = (uint64_t)A.num_queries * A.strands;
    if (g >= nblocks) return;
    uint32_t qi = (uint32_t)(A.strands == 2 ? g >> 1 : g);
    uint32_t rev = A.strands == 2 ? (uint32_t)(g & 1u) : 0u;
    uint64_t o0 = A.offsets[qi], o1 = A.offsets[qi + 1];
    uint32_t len = (uint32_t)(o1 - o0);
    const IndexView& ix = A.ix;
    const int L = (int)A.min_len;
    QueryCursor qc;
    qc.init(A.qwords, o0, len, rev);
    uint32_t top = 0, bot = ix.n, prev_top = 0, prev_bot = ix.n;  // slamem.c:110-113
    int depth = 0;
    uint32_t k = 0;
    for (uint32_t j = len; j-- > 0u;) {
        uint32_t c = qc.at(j);
        uint32_t size = 0;
        for (;;) {  // slamem.c:121-128
            uint32_t t = top, b = bot;
            if (follow(ix, c, t, b)) { top = t; bot = b; size = b - t + 1u; break; }
            top = prev_top;
            bot = prev_bot;
            depth = parent(ix, top, bot);
            if (depth < 0) break;
            prev_top = top;
            prev_bot = bot;
        }
        depth++;
        if (depth >= L) {
            if (size != 1u) continue;  // slamem.c:131 (prev_top / prev_bot keep their old values)
            uint32_t left = j ? qc.at(j - 1u) : 0xFFu;  // slamem.c:137-138
            int pub = 0x3FFFFFFF;
            emit_levels(A, (uint32_t)g, k, top, bot, depth, j, left, bwt_code(ix, top) == left ? 1u : 0u, pub);
        }
        prev_top = top;  // slamem.c:197-198
        prev_bot = bot;
    }
    A.block_counts[g] = k;
}

// Start states of the slices of long records.  The scan of slice [a,b) needs the state the full scan has after
// position b: the interval and length of the longest match that starts at b.  A scan started at e = b + warm-up from the
// root finds min(true length, e - b) there (the truncation property), so (k_slice_states, a lane per slice):
//   * an extension failed on the way (depth < e - b): the state is the full scan's;
//   * none failed and the interval has several rows (a repeat longer than the warm-up): four times the warm-up;
//   * none failed and the interval is ONE row: q[b..e) occurs once in the text, at r = SA[row] -- the full scan is on that
//     row too, and its length is e - b plus the letters that agree further right, found by comparing the strand with the
//     text itself, at most kSliceLen letters of it: by then the comparison is past the next slice's start b' = b + kSliceLen.
// If the letters still agree there (the state is left OPEN), k_slice_chain (a lane per strand, right to left) finishes
// it from the right neighbour's state: when that is one row on the SAME diagonal (text position - strand position), the
// match from b is the neighbour's match from b' plus the kSliceLen letters in between.  Every slice of a genome compared
// with itself is such a case: the chain costs one step per slice where independent comparisons would cost len^2 / 8192
// letters (measured: 1.48 s for 4.6 Mbp).  Only when the neighbour is on another diagonal does the lane compare on.
// K8 starts a slice at b from the stored state (no warm-up in the search kernel); -mam takes it as its guess when no
// extension failed (k_find_mams_sliced).
struct __attribute__((aligned(16))) SliceState {
    uint32_t top, bot;
    int32_t depth;
    uint32_t flags;   // kSsValid | kSsOneRow (diag is set) | kSsNoFail (no extension failed in the warm-up) | kSsOpen
    int64_t diag;     // one row: text position of the match - strand position (the diagonal)
    uint32_t seen;    // open: letters of the strand from b on that are known to agree (the comparison stopped there)
    uint32_t pad;
};
enum : uint32_t { kSsValid = 1u, kSsOneRow = 2u, kSsNoFail = 4u, kSsOpen = 8u };

// letters that agree between the strand from position qp on and the text from position tp on, at most `limit` (letter codes
// of K1's packed words in the text groups; past the text the code is 0, which no strand letter has)
__device__ __forceinline__ uint32_t agree_forward(const IndexView& ix, QueryCursor& qc, uint32_t qp, uint32_t len, uint64_t tp,
                                                  uint32_t limit) {
    const uint32_t q0 = qp;
    uint64_t gi = ~0ull, letters = 0;
    while (qp < len && tp < (uint64_t)ix.n && qp - q0 < limit) {
        if ((tp >> 4) != gi) { gi = tp >> 4; letters = ix.tgrp[gi].letters; }
        const uint32_t tc = (uint32_t)(letters >> (60u - 4u * (uint32_t)(tp & 15u))) & 15u;
        if (tc != qc.at(qp)) break;
        qp++; tp++;
    }
    return qp - q0;
}

__global__ void __launch_bounds__(256) k_slice_states(SearchArgs A, SliceState* __restrict__ out, uint32_t warm_up) {
    const uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= A.num_items) return;
    const ItemDesc d = A.items[it];
    const uint32_t len = d.len, rev = d.slice_rev >> 31, sl = d.slice_rev & 0x7FFFFFFFu;
    const uint32_t a = sl * kSliceLen;
    if (len - a <= kSliceLen) return;  // the strand's rightmost slice (or an unsliced strand): K8 starts at the record's end
    const uint32_t b = a + kSliceLen;
    const IndexView& ix = A.ix;
    SliceState s;
    s.top = 0; s.bot = ix.n; s.depth = 0; s.flags = 0; s.diag = 0; s.seen = 0; s.pad = 0;
    const uint64_t x = it - 1u - A.item_block[it];  // boundary to the right of this slice (see MamPass)
    if (A.item_alive && !A.item_alive[it]) { out[x] = s; return; }  // proven empty by the prefilter: never scanned
    QueryCursor qc;
    qc.init(A.qwords, d.base, len, rev);
    uint32_t w = warm_up;
    for (;;) {
        const uint32_t e = (len - b <= w) ? len : b + w;
        uint32_t top = 0, bot = ix.n;
        int depth = 0;
        for (uint32_t j = e; j-- > b;) {  // slamem.c:121-129
            const uint32_t c = qc.at(j);
            for (;;) {
                uint32_t t = top, bb = bot;
                if (follow(ix, c, t, bb)) { top = t; bot = bb; break; }
                depth = parent(ix, top, bot);
                if (depth < 0) break;
            }
            depth++;
        }
        const bool nofail = depth >= (int)(e - b);
        if (e != len && nofail && (top != bot || !ix.tgrp)) { w = w < 0x20000000u ? w * 4u : 0xFFFFFFFFu; continue; }
        s.flags = kSsValid | (nofail ? kSsNoFail : 0u);
        if (top == bot && depth > 0) {
            const uint64_t r = ix.sa[top];  // the text position that faces strand position b
            s.flags |= kSsOneRow;
            s.diag = (int64_t)r - (int64_t)b;
            if (e != len && nofail) {  // the match may go on to the right of e
                const uint32_t n = agree_forward(ix, qc, e, len, r + (e - b), kSliceLen);
                depth += (int)n;
                if (n == kSliceLen) { s.flags |= kSsOpen; s.seen = (uint32_t)depth; }
            }
        }
        s.top = top; s.bot = bot; s.depth = depth;
        break;
    }
    out[x] = s;
}

// the open states of a strand, right to left (one lane per strand: the item of its rightmost slice)
__global__ void __launch_bounds__(256) k_slice_chain(SearchArgs A, SliceState* __restrict__ st) {
    const uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= A.num_items) return;
    const ItemDesc d = A.items[it];
    const uint32_t len = d.len, rev = d.slice_rev >> 31, sl = d.slice_rev & 0x7FFFFFFFu;
    const uint32_t cnt = len ? (len + kSliceLen - 1u) / kSliceLen : 1u;
    if (cnt < 3u || sl != cnt - 1u) return;  // (the slice next to the rightmost one is never open: its comparison reaches the end)
    const IndexView& ix = A.ix;
    const uint64_t x0 = it - A.item_block[it];  // boundary between the rightmost slice and its left neighbour
    QueryCursor qc;
    qc.init(A.qwords, d.base, len, rev);
    for (uint32_t k = 1; k + 1u < cnt; k++) {  // state k belongs to the slice k+1 from the right; its right neighbour's is k-1
        SliceState s = st[x0 + k];
        if (!(s.flags & kSsOpen)) continue;
        const SliceState r = st[x0 + k - 1u];
        if ((r.flags & kSsValid) && (r.flags & kSsOneRow) && !(r.flags & kSsOpen) && r.diag == s.diag) {
            s.depth = (int)kSliceLen + r.depth;  // the neighbour's match from its b, and the letters in between
        } else {  // (a dead or multi-row neighbour, or one whose longest match lies on another diagonal): compare on
            const uint32_t b = (cnt - 1u - k) * kSliceLen;
            const uint32_t qp = b + s.seen;
            s.depth = (int)s.seen + (int)agree_forward(ix, qc, qp, len, (uint64_t)((int64_t)qp + s.diag), 0xFFFFFFFFu);
        }
        s.flags &= ~kSsOpen;
        st[x0 + k] = s;
    }
}

// -mam over SLICES of long strands (genome against genome: one lane per whole strand took 9.8 s for a 4.6 Mbp pair).
// The scan's state between two positions is (top, bot, prev_top, prev_bot, depth) -- with the reference's stale fall-back
// interval there is no property that tells when a scan started further right has the same state as the full scan, so the
// slices are SPECULATED and VERIFIED:
//   pass 0   every slice [a,b) of a strand starts at e = b + warm-up from the root state without emitting, notes the state
//            it has when it reaches b (in_used), emits [a,b) from there and notes the state it ends with (out).  A warm-up
//            in which no extension failed (depth = e - b) cannot have met the true state: it is repeated four times as long.
//   check    the boundary between slice i (right) and i+1 (left) is consistent if in_used == out: the left slice did exactly
//            what the full scan does, PROVIDED the right slice did.  The rightmost slice starts at the strand's end: by
//            induction every slice to the right of the first inconsistent boundary is final.
//   rerun    the slice left of the first inconsistent boundary of each strand is scanned again from `out` (now final),
//            with attempt tag 1 (each slice is rerun at most once: its start state is final then); check again.
// The scan is a deterministic function of the state, so the result is the full scan's, whatever the warm-up guessed.
struct __attribute__((aligned(16))) MamState {
    uint32_t top, bot, prev_top, prev_bot;
    int32_t depth;
    uint32_t pad0, pad1, pad2;
};
struct MamPass {
    const uint32_t* run_list;   // nullptr: pass 0, every item
    const uint32_t* run_count;
    MamState* in_used;          // per boundary: boundary x = (item index of its right slice) - (strand block number)
    MamState* out;
    const uint32_t* item_block; // strand block number of every item
    const SliceState* guess;    // k_slice_states + k_slice_chain: taken where no extension failed and the match is one row
    const uint8_t* alive;       // nullptr, or 0 for strands that the presence filter proved to hold no match >= min_len
    uint32_t slice_len, warm_up;
};

__device__ __forceinline__ bool same_state(const MamState& a, const MamState& b) {
    return a.top == b.top && a.bot == b.bot && a.prev_top == b.prev_top && a.prev_bot == b.prev_bot && a.depth == b.depth;
}

// one position of the reference's loop without the emission (slamem.c:121-129); returns the size of the new interval
__device__ __forceinline__ uint32_t mam_step(const IndexView& ix, uint32_t c, MamState& s) {
    uint32_t size = 0;
    for (;;) {
        uint32_t t = s.top, b = s.bot;
        if (follow(ix, c, t, b)) { s.top = t; s.bot = b; size = b - t + 1u; break; }
        s.top = s.prev_top;
        s.bot = s.prev_bot;
        s.depth = parent(ix, s.top, s.bot);
        if (s.depth < 0) break;
        s.prev_top = s.top;
        s.prev_bot = s.bot;
    }
    s.depth++;
    return size;
}

__global__ void __launch_bounds__(256) k_find_mams_sliced(SearchArgs A, MamPass P) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool rerun = P.run_list != nullptr;
    uint64_t it;
    if (rerun) { if (g >= *P.run_count) return; it = P.run_list[g]; }
    else { if (g >= A.num_items) return; it = g; }
    const ItemDesc d = A.items[it];
    const uint32_t len = d.len, rev = d.slice_rev >> 31, c_idx = d.slice_rev & 0x7FFFFFFFu;
    const uint32_t cnt = len ? (len + P.slice_len - 1u) / P.slice_len : 1u;
    const uint32_t i = cnt - 1u - c_idx;  // 0 = the strand's rightmost slice
    const uint32_t a = c_idx * P.slice_len, b = (len - a <= P.slice_len) ? len : a + P.slice_len;
    const uint64_t xr = it - 1u - P.item_block[it];  // boundary to the right (i > 0), to the left: xr + 1 (i < cnt - 1)
    if (P.alive && cnt == 1u && !P.alive[it]) {  // no match of min_len letters anywhere in the strand: no MAM either
        A.block_counts[it] = 0;                  // (slices of long strands are scanned regardless: their neighbours need their states)
        return;
    }
    const IndexView& ix = A.ix;
    const int L = (int)A.min_len;
    QueryCursor qc;
    qc.init(A.qwords, d.base, len, rev);
    MamState s;
    s.top = 0; s.bot = ix.n; s.prev_top = 0; s.prev_bot = ix.n; s.depth = 0;  // slamem.c:110-113
    s.pad0 = s.pad1 = s.pad2 = 0;
    uint32_t attempt = 0;
    if (i > 0u) {
        if (rerun) {
            s = P.out[xr];
            attempt = (uint32_t)A.item_attempt[it] + 1u;
        } else {
            bool guessed = false;
            if (P.guess) {
                // a warm-up in which no extension failed and whose match is one row: the full scan is on that row too (it
                // emitted it at b, so its fall-back interval is the row), with the length that the comparison with the text
                // gave (k_slice_states / k_slice_chain) -- a guess like any other, verified by k_mam_check
                const SliceState gs = P.guess[xr];
                const uint32_t need = kSsValid | kSsOneRow | kSsNoFail;
                if ((gs.flags & (need | kSsOpen)) == need && gs.depth >= L) {
                    s.top = s.bot = s.prev_top = s.prev_bot = gs.top;
                    s.depth = gs.depth;
                    guessed = true;
                }
            }
            uint32_t w = P.warm_up;
            while (!guessed) {
                const uint32_t e = (len - b <= w) ? len : b + w;
                s.top = 0; s.bot = ix.n; s.prev_top = 0; s.prev_bot = ix.n; s.depth = 0;
                for (uint32_t j = e; j-- > b;) {
                    uint32_t size = mam_step(ix, qc.at(j), s);
                    if (!(s.depth >= L && size != 1u)) { s.prev_top = s.top; s.prev_bot = s.bot; }  // slamem.c:131 / :197-198
                }
                if (e == len || s.depth < (int)(e - b)) break;
                w = w < 0x20000000u ? w * 4u : 0xFFFFFFFFu;
            }
        }
        P.in_used[xr] = s;
    }
    uint32_t k = 0;
    for (uint32_t j = b; j-- > a;) {
        uint32_t size = mam_step(ix, qc.at(j), s);
        if (s.depth >= L) {
            if (size != 1u) continue;  // slamem.c:131 (prev_top / prev_bot keep their old values)
            uint32_t left = j ? qc.at(j - 1u) : 0xFFu;  // slamem.c:137-138
            int pub = 0x3FFFFFFF;
            emit_levels(A, (uint32_t)it, k, s.top, s.bot, s.depth, j, left, bwt_code(ix, s.top) == left ? 1u : 0u, pub, attempt << 28);
        }
        s.prev_top = s.top;  // slamem.c:197-198
        s.prev_bot = s.bot;
    }
    if (i + 1u < cnt) P.out[xr + 1u] = s;
    A.block_counts[it] = k;
    A.item_attempt[it] = (uint8_t)attempt;
}

// the slices to scan again: left of the FIRST inconsistent boundary of their strand (see above)
__global__ void __launch_bounds__(256) k_mam_check(SearchArgs A, MamPass P, uint32_t* __restrict__ list, unsigned int* __restrict__ count) {
    const uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= A.num_items) return;
    const ItemDesc d = A.items[it];
    const uint32_t len = d.len, c_idx = d.slice_rev & 0x7FFFFFFFu;
    const uint32_t cnt = len ? (len + P.slice_len - 1u) / P.slice_len : 1u;
    const uint32_t i = cnt - 1u - c_idx;
    if (i == 0u) return;
    const uint64_t xr = it - 1u - P.item_block[it];
    if (same_state(P.in_used[xr], P.out[xr])) return;
    for (uint32_t r = 1; r < i; r++)
        if (!same_state(P.in_used[xr - r], P.out[xr - r])) return;  // not the first one
    list[atomicAdd(count, 1u)] = (uint32_t)it;
}

// ------------------------------------------------------------------------------------------
// K8 v3: the same scan as a persistent, desynchronised state machine with ONE memory phase per trip.
//
// Round 1's first kernel (one lane per strand, all 64 strands of a wave walked through position j in lockstep; removed
// in round 2) showed why: per-lane rare events (a failed extension, an emitted MEM, a record that starts) are per-wave
// COMMON events, and each one is a dependent round trip that stalls the whole wave with a handful of loads in flight:
// measured 66 requests in flight per CU, TA 75 % busy, 1100-cycle request latency -- latency-bound far below the
// random-line ceiling.
//
// Here every loop trip issues all of its loads up front from addresses known since the previous trip -- FM block
// of `top`, FM block of `bot+1`, the two row records, the next query word -- waits once, and then every lane
// advances its OWN strand by what arrived: a successful extension moves to the next position, a failed one
// applies the parent step and retries in the next trip (state REC when the records were not fetched).
//   * MEMs are stored without any returned atomic: the first kInlineMems of a strand go to a directly addressed
//     slot (block*kInlineMems + k), the BWT row is kept and resolved to SA[row] by K9;
//   * a finished lane takes the next strand of its wave's chunk (ballot + prefix count), record offsets come from
//     an LDS copy made once per wave;
//   * only genuinely rare work (intervals of several rows, ancestors that are still >= min_len deep, the letter N,
//     more than kInlineMems MEMs in one strand) takes divergent dependent loads.
// Results: the reference's MEMs in the reference's per-strand emission order.
// ------------------------------------------------------------------------------------------
// store MEM number kk of strand block g (kk is assigned by the caller)
__device__ __forceinline__ void emit3_at(const SearchArgs& A, uint32_t g, uint32_t kk, uint32_t tag, uint32_t row,
                                         uint32_t pos, uint32_t len) {
    if (kk < kInlineMems) {
        A.inline_rows[(uint64_t)kk * A.inline_stride + g] = RawRow{row, pos, len};
    } else {
        if (kk >> 28) atomicOr(reinterpret_cast<unsigned int*>(A.total) + 9, 1u);  // the ordinal would run into the tag: reported as an error
        unsigned long long slot = atomicAdd(A.total, 1ull);
        if (slot < A.capacity) {
            A.raw_key[slot] = RawKey{g, kk | tag};  // tag = attempt << 28: records of abandoned attempts are skipped
            A.raw_mem[slot] = slamem_mem{row, pos, len};  // ref_pos holds the ROW until K9
        }
    }
}

// A lane's state between two trips of K8 (kCarry): 64 bytes
struct __attribute__((aligned(16))) CarryRec {
    uint32_t st_flags;  // st | pend << 8 | dmis << 9 | dcool << 10
    uint32_t g, j, top, bot, k, qlen, dir_r;
    int32_t depth, pub;
    uint64_t qp;        // the strand's first word in the packed copy
    uint32_t prev_top, prev_bot, pad0, pad1;
};
static_assert(sizeof(CarryRec) == 64, "carry record is one line");

// A deferred enumeration job (kDefer): what wave_enumerate needs, and where its MEM count goes
struct __attribute__((aligned(16))) JobRec {
    uint32_t g, kbase, t, b;
    uint32_t depth_pos;   // depth | pos << 16   (reads only: both below 65536)
    uint32_t left_flags;  // left letter | level0 << 8 | walk up << 9
    uint32_t segabs;      // place of the job's count in the pool
    uint32_t epoch;       // DeferCtx::epoch of the launch that wrote the record
};
static_assert(sizeof(JobRec) == 32, "job record is 32 bytes");

// emit3_at with the output side chosen per lane: `old` lanes (carried in from the previous launch) write to A.prev
template <bool kCarry>
__device__ __forceinline__ void emit3_sel(const SearchArgs& A, bool old, uint32_t g, uint32_t kk, uint32_t tag, uint32_t row,
                                          uint32_t pos, uint32_t len) {
    if (!kCarry) { emit3_at(A, g, kk, tag, row, pos, len); return; }
    if (kk < kInlineMems) {
        RawRow* ir = old ? A.prev.inline_rows : A.inline_rows;
        ir[(uint64_t)kk * (old ? A.prev.inline_stride : A.inline_stride) + g] = RawRow{row, pos, len};
    } else {
        unsigned long long* tot = old ? A.prev.total : A.total;
        if (kk >> 28) atomicOr(reinterpret_cast<unsigned int*>(tot) + 9, 1u);
        unsigned long long slot = atomicAdd(tot, 1ull);
        if (slot < (old ? A.prev.capacity : A.capacity)) {
            (old ? A.prev.raw_key : A.raw_key)[slot] = RawKey{g, kk | tag};
            (old ? A.prev.raw_mem : A.raw_mem)[slot] = slamem_mem{row, pos, len};
        }
    }
}

// kDefer: a MEM the lane reports itself AFTER its strand's first deferred job: its ordinal counts only what the lane reported
// (the jobs' MEMs before it are not known yet); the record goes to the list with the place of its offset in the pool
__device__ __forceinline__ void emit_provisional(const SearchArgs& A, uint32_t g, uint32_t kk, uint32_t row, uint32_t pos, uint32_t len,
                                                 uint32_t segabs) {
#ifdef SLAMEM_DIAG_NO_PROV   // (timing experiments only: wrong results)
    return;
#endif
    if (kk >> 28) atomicOr(reinterpret_cast<unsigned int*>(A.total) + 9, 1u);
    const unsigned long long slot = atomicAdd(A.total, 1ull);
    if (slot < A.capacity) {
        A.raw_key[slot] = RawKey{g, kk};
        A.raw_mem[slot] = slamem_mem{row, pos, len};
        A.defer.raw_seg[slot] = segabs;
    }
}

// The rows a wave reports in ONE step of an enumeration job: lane `lane` reports (row, pos, len) as MEM number
// k + (reporting lanes below it) of strand block g when `ok`.  The first kInlineMems of a strand go to its inline slots; all
// the others of the step take their places in the overflow list with ONE atomic for the wave.
// kChunk: the instantiation for repeat-rich texts.  The list's counter is ONE word that 4096 waves hit; an atomic per
// emitting step runs at ~150 M/s there and WAS the kernel on such texts (a 10^5-copy family at -l 50: 12.6 M steps, 83 ms per
// million reads; -l 20: 479 ms per 100,000 reads; profiles/r03_repeat_load.jsonl).  Here a wave reserves kOvfChunk places at
// a time and hands them out by itself (ovf_base / ovf_left, wave-uniform, kept in LDS between jobs); places that stay unused
// keep the mark the whole list is filled with before the launch (block 0xFFFFFFFF), K9 skips them.  It is a separate
// instantiation because the headline kernel's register allocation does not survive the extra code (+2.4 ms of 20.9).
constexpr uint32_t kOvfChunk = 128;
template <bool kCarry, bool kChunk>
__device__ __forceinline__ uint32_t wave_emit_step(const SearchArgs& A, bool old, uint32_t lane, uint32_t g, uint32_t k, uint32_t tag,
                                                   bool ok, uint32_t row, uint32_t pos, uint32_t len,
                                                   unsigned long long& ovf_base, uint32_t& ovf_left) {
    const unsigned long long m = __ballot(ok);
    if (m == 0ull) return k;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t kk = k + (uint32_t)__popcll(m & below);
    const bool inl = ok && kk < kInlineMems;
    const unsigned long long mo = __ballot(ok && !inl);
    if (inl) {
        RawRow* ir = (kCarry && old) ? A.prev.inline_rows : A.inline_rows;
        ir[(uint64_t)kk * ((kCarry && old) ? A.prev.inline_stride : A.inline_stride) + g] = RawRow{row, pos, len};
    }
    if (mo != 0ull) {
        unsigned long long* tot = (kCarry && old) ? A.prev.total : A.total;
        const uint64_t cap = (kCarry && old) ? A.prev.capacity : A.capacity;
        const int leader = __ffsll((long long)mo) - 1;
        unsigned long long first;  // place of the step's first record
        if (kChunk && !(kCarry && old)) {
            const uint32_t need = (uint32_t)__popcll(mo);
            if (ovf_left < need) {  // (wave-uniform) a new chunk; what is left of the old one stays marked as unused
                const uint32_t chunk = need > kOvfChunk ? need : kOvfChunk;
                unsigned long long base = 0ull;
                if ((int)lane == leader) base = atomicAdd(tot, (unsigned long long)chunk);
                // (readlane, not a shuffle: the result is wave-uniform and the compiler knows it)
                ovf_base = u64_of((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, leader),
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), leader));
                ovf_left = chunk;
            }
            first = ovf_base;
            ovf_base += need;
            ovf_left -= need;
        } else {
            unsigned long long base = 0ull;
            if ((int)lane == leader) base = atomicAdd(tot, (unsigned long long)__popcll(mo));
            first = u64_of((uint32_t)__shfl((int)(uint32_t)base, leader), (uint32_t)__shfl((int)(uint32_t)(base >> 32), leader));
        }
        if (ok && !inl) {
            if (kk >> 28) atomicOr(reinterpret_cast<unsigned int*>(tot) + 9, 1u);  // the ordinal would run into the tag: reported as an error
            const unsigned long long slot = first + (unsigned long long)__popcll(mo & below);
            if (slot < cap) {
                ((kCarry && old) ? A.prev.raw_key : A.raw_key)[slot] = RawKey{g, kk | tag};
                ((kCarry && old) ? A.prev.raw_mem : A.raw_mem)[slot] = slamem_mem{row, pos, len};  // ref_pos holds the ROW until K9
            }
        }
    }
    return k + (uint32_t)__popcll(m);
}

// One enumeration job of ONE strand, executed by the WHOLE wave (every argument is wave-uniform): all rows of
// [t,b] at depth `msz` (only if `level0`), then of every ancestor interval still >= L deep, the new rows above
// ascending and the new rows below descending (slamem.c:139-193).  The 64 lanes test 64 rows at a time for
// left-maximality (BWT letter != left); __ballot + prefix popcount give each surviving row its MEM number, so the
// strand's emission order is exactly the reference's.  This is what makes repeats (intervals of thousands of
// rows) cost rows/64 steps instead of rows.  Returns the strand's new MEM count; *first_parent = depth of the
// parent of [t,b] (the exact value of `pub`).
template <bool kCarry, bool kChunk>
__device__ __forceinline__ uint32_t wave_enumerate(const SearchArgs& A, bool old, uint32_t lane, uint32_t g, uint32_t k,
                                                   uint32_t tag, uint32_t t, uint32_t b, int msz, bool level0, bool walk_up,
                                                   uint32_t pos, uint32_t left, int L, int* first_parent,
                                                   unsigned long long& ovf_base, uint32_t& ovf_left,
                                                   uint32_t* row_steps = nullptr, uint32_t* levels = nullptr) {
    const IndexView& ix = A.ix;
    uint32_t pt = level0 ? b + 1u : t, pb = b;  // rows already reported: [pt, pb]
    bool first = true;
    *first_parent = -2;
    for (;;) {
        for (uint32_t base = t; base < pt; base += 64u) {  // new rows above, ascending (slamem.c:140)
            uint32_t row = base + lane;
            bool ok = row < pt && bwt_code(ix, row) != left;
            k = wave_emit_step<kCarry, kChunk>(A, old, lane, g, k, tag, ok, row, pos, (uint32_t)msz, ovf_base, ovf_left);
            if (row_steps) (*row_steps)++;
        }
        for (uint32_t done = 0; done < b - pb; done += 64u) {  // new rows below, bottom-up (slamem.c:165)
            uint32_t off = done + lane;
            uint32_t row = b - off;
            bool ok = off < b - pb && bwt_code(ix, row) != left;
            k = wave_emit_step<kCarry, kChunk>(A, old, lane, g, k, tag, ok, row, pos, (uint32_t)msz, ovf_base, ovf_left);
            if (row_steps) (*row_steps)++;
        }
        if (!walk_up) break;
        pt = t;
        pb = b;
        msz = parent(ix, t, b);  // same address in every lane: one line, broadcast (slamem.c:192)
        if (levels) (*levels)++;
        if (first) { *first_parent = msz; first = false; }
        if (msz < L) break;
    }
    return k;
}

// wave_enumerate for the kChunk instantiations (repeat-rich texts: enumeration is most of the kernel there, and a job is a chain
// of levels).  Same order, same results; ONE memory round trip per level instead of three: the first 64 rows above, the first 64
// rows below and the parent's records are independent of each other (the parent depends on [t,b] only), so their loads go out
// together, and a row's three plane words are read at once instead of the exception word first.  Levels with more than 64 new
// rows on a side take further steps as before.
__device__ __forceinline__ uint32_t bwt_code_all(const IndexView& ix, uint32_t row) {
    const FMBlock* blk = ix.fm + (row >> kFmRowsLog2);
    const uint32_t o = row & (kFmRows - 1u), hs = o >> 6, bit = o & 63u;
    const uint64_t e = blk->ex[hs], p0 = blk->p0[hs], p1 = blk->p1[hs];
    const uint32_t acgt = 2u + (uint32_t)((p0 >> bit) & 1ull) + 2u * (uint32_t)((p1 >> bit) & 1ull);
    return ((e >> bit) & 1ull) ? (row == ix.dollar_row ? 0u : 1u) : acgt;
}
template <bool kCarry>
__device__ __forceinline__ uint32_t wave_enumerate_merged(const SearchArgs& A, bool old, uint32_t lane, uint32_t g, uint32_t k,
                                                          uint32_t tag, uint32_t t, uint32_t b, int msz, bool level0, bool walk_up,
                                                          uint32_t pos, uint32_t left, int L, int* first_parent,
                                                          unsigned long long& ovf_base, uint32_t& ovf_left) {
    const IndexView& ix = A.ix;
    const uint4* R = reinterpret_cast<const uint4*>(ix.rec);
    uint32_t pt = level0 ? b + 1u : t, pb = b;  // rows already reported: [pt, pb]
    bool first = true;
    *first_parent = -2;
    for (;;) {
        // ---- one memory phase: rows t+lane (above), b-lane (below), the records of [t,b] -----------------------------
        const uint32_t ra = t + lane, rb = b - lane;
        const bool wa = ra < pt, wb = lane < b - pb;
        uint32_t ca = left, cb = left;
        if (wa) ca = bwt_code_all(ix, ra);
        if (wb) cb = bwt_code_all(ix, rb);
        uint4 rt = make_uint4(0, 0, 0, 0), rbm = rt;
        if (walk_up) { rt = R[t]; rbm = R[b]; }  // (same address in every lane: one line each, broadcast)
        // ---- new rows above, ascending (slamem.c:140) ------------------------------------------------------------------
        if (t < pt) k = wave_emit_step<kCarry, true>(A, old, lane, g, k, tag, wa && ca != left, ra, pos, (uint32_t)msz, ovf_base, ovf_left);
        for (uint32_t base = t + 64u; base < pt && base > t; base += 64u) {
            const uint32_t row = base + lane;
            const bool ok = row < pt && bwt_code(ix, row) != left;
            k = wave_emit_step<kCarry, true>(A, old, lane, g, k, tag, ok, row, pos, (uint32_t)msz, ovf_base, ovf_left);
        }
        // ---- new rows below, bottom-up (slamem.c:165) ------------------------------------------------------------------
        if (b != pb) k = wave_emit_step<kCarry, true>(A, old, lane, g, k, tag, wb && cb != left, rb, pos, (uint32_t)msz, ovf_base, ovf_left);
        for (uint32_t done = 64u; done < b - pb; done += 64u) {
            const uint32_t off = done + lane;
            const uint32_t row = b - off;
            const bool ok = off < b - pb && bwt_code(ix, row) != left;
            k = wave_emit_step<kCarry, true>(A, old, lane, g, k, tag, ok, row, pos, (uint32_t)msz, ovf_base, ovf_left);
        }
        if (!walk_up) break;
        pt = t;
        pb = b;
        msz = parent_from(rt, rbm, t, b);  // (slamem.c:192)
        if (first) { *first_parent = msz; first = false; }
        if (msz < L) break;
    }
    return k;
}

// ---- K8e: the queue of deferred enumeration jobs (kDefer) -------------------------------------------------------------------
// Every wave of the chip takes jobs from the queue, FOUR at a time: a group of 16 lanes per job.  A job is a chain of levels (the
// interval, then its ancestors down to min_len: one memory round trip each), and on a repeat-rich text most rows of a level
// are NOT reported -- the copies of a family mostly agree on the letter to the left -- so the rows are not looked at one per
// lane: a lane takes 64 rows at once as the three plane words of their half FM block and finds the rows whose letter differs
// from the query's with a few bitwise operations (a group: 1,024 rows above and 1,024 below per step).  Order as in
// wave_enumerate (slamem.c:139-193): the new rows above ascending, then the new rows below descending, then the parent.  All
// groups of a wave step together (a group without a job idles through the step); MEM numbers come from sums over the group,
// places in the list from the wave's chunk.  The job's MEM count goes to its place in the pool.
__global__ void __launch_bounds__(256) k_enum_jobs(SearchArgs A) {
    const IndexView& ix = A.ix;
    const uint4* R = reinterpret_cast<const uint4*>(ix.rec);
    const int L = (int)A.min_len;
    const uint32_t lane = threadIdx.x & 63u, gl = lane & 15u;
    const uint32_t ngroups = gridDim.x * (blockDim.x >> 4), group = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const uint32_t nres = *A.defer.njobs;  // places of the queue given out (in chunks: some hold no record of this launch)
    const uint32_t njobs = nres < A.defer.queue_cap ? nres : A.defer.queue_cap;
    unsigned long long ovf_base = 0ull;  // the wave's chunk of the list (wave-uniform)
    uint32_t ovf_left = 0u;
    uint32_t q = group;                  // the group's next job
    bool busy = false;
    // the group's job (the same in its 16 lanes)
    uint32_t g = 0, k = 0, kbase = 0, t = 0, b = 0, pt = 0, pb = 0, pos = 0, left = 0, segabs = 0, aoff = 0, boff = 0;
    int msz = 0;
    bool walk_up = false;
#ifdef SLAMEM_DIAG_TRIPS
    uint32_t dg_steps = 0, dg_busy = 0;
#endif
    for (;;) {
#ifdef SLAMEM_DIAG_TRIPS
        dg_steps++; dg_busy += (uint32_t)__popcll(__ballot(busy)) >> 4;
#endif
        if (!busy && q < njobs) {
            const uint4* jr = reinterpret_cast<const uint4*>(A.defer.queue + q);
            const uint4 j0 = jr[0], j1 = jr[1];
            q += ngroups;
            g = j0.x; kbase = k = j0.y; t = j0.z; b = j0.w;
            msz = (int)(j1.x & 0xFFFFu); pos = j1.x >> 16; left = j1.y & 0xFFu;
            walk_up = (j1.y & 0x200u) != 0u;
            pt = (j1.y & 0x100u) ? b + 1u : t;  // rows already reported: [pt, pb]
            pb = b;
            segabs = j1.z;
            aoff = 0; boff = 0;
            busy = j1.w == A.defer.epoch;  // (else: a place nobody filled in this launch -- the group idles through this step)
        }
        if (__ballot(busy) == 0ull) {  // (wave-uniform) nobody has a job: the end, unless a group's next place just held none
            if (__ballot(q < njobs) == 0ull) break;
            continue;
        }
        // ---- one memory phase: this lane's half block above, its half block below, the records of [t,b] -------------------
        const uint32_t na = pt - t, nb = b - pb;                 // new rows above [t, pt) / below (pb, b] at this level
        const bool do_a = busy && aoff < na;
        const uint32_t a_lo = t + aoff;                          // first row above not looked at yet
        const uint32_t a_end = (a_lo & ~63u) + 1024u;            // the group's 16 half blocks end here (wraps only behind 2^32 rows)
        const bool a_last = !do_a || a_end >= pt || a_end < a_lo;  // the rows above are through with this step
        const bool do_b = busy && a_last && boff < nb;
        const uint32_t b_hi = b - boff;                          // last row below not looked at yet
        const uint32_t b_blk = b_hi & ~63u;                      // first row of its half block
        const bool b_reach = b_blk < 960u || b_blk - 960u <= pb + 1u;  // the group's 16 half blocks reach down to pb + 1
        const bool b_last = a_last && (!do_b || b_reach);        // ... and the rows below: the level ends
        const uint32_t ha = (a_lo & ~63u) + 64u * gl;            // this lane's half block above: rows [ha, ha + 64)
        const bool wa = do_a && ha < pt && ha >= (a_lo & ~63u);
        const bool wb = do_b && b_blk >= 64u * gl && b_blk - 64u * gl + 63u > pb;
        const uint32_t hb = b_blk - 64u * gl;                    // this lane's half block below: rows [hb, hb + 64)
        uint64_t ea = 0, a0 = 0, a1 = 0, eb = 0, b0 = 0, b1 = 0;
        if (wa) { const FMBlock* blk = ix.fm + (ha >> kFmRowsLog2); const uint32_t hs = (ha & (kFmRows - 1u)) >> 6; ea = blk->ex[hs]; a0 = blk->p0[hs]; a1 = blk->p1[hs]; }
        if (wb) { const FMBlock* blk = ix.fm + (hb >> kFmRowsLog2); const uint32_t hs = (hb & (kFmRows - 1u)) >> 6; eb = blk->ex[hs]; b0 = blk->p0[hs]; b1 = blk->p1[hs]; }
        uint4 rt = make_uint4(0, 0, 0, 0), rbm = rt;
        if (busy && b_last && walk_up) { rt = R[t]; rbm = R[b]; }
        // (every load of the step is out before anything is looked at)
        asm volatile("" : "+v"(ea), "+v"(a0), "+v"(a1), "+v"(eb), "+v"(b0), "+v"(b1));
        asm volatile("" : "+v"(rt.x), "+v"(rt.y), "+v"(rt.z), "+v"(rt.w), "+v"(rbm.x), "+v"(rbm.y), "+v"(rbm.z), "+v"(rbm.w));
        // rows of a half block whose BWT letter differs from `left` (left: 2..5 = A,C,G,T; 1 = N; 0xFF at a strand's end: every row)
        auto differs = [&](uint32_t row0, uint64_t e, uint64_t p0, uint64_t p1) -> uint64_t {
            const uint32_t c2 = left - 2u;
            const uint64_t f0 = (c2 & 1u) ? ~0ull : 0ull, f1 = (c2 & 2u) ? ~0ull : 0ull;
            uint64_t same = left >= 2u && left <= 5u ? (~(p0 ^ f0) & ~(p1 ^ f1) & ~e) : 0ull;  // rows that hold the letter itself
            if (left == 1u) {  // N: the exception rows but the one of '$'
                same = e;
                if (ix.dollar_row >= row0 && ix.dollar_row - row0 < 64u) same &= ~(1ull << (ix.dollar_row - row0));
            }
            return ~same;
        };
        uint64_t ma = 0, mb = 0;
        if (wa) {
            const uint32_t lo = a_lo > ha ? a_lo - ha : 0u, hi = pt - ha < 64u ? pt - ha : 64u;  // rows [ha + lo, ha + hi)
            ma = differs(ha, ea, a0, a1) & (hi >= 64u ? ~0ull : (1ull << hi) - 1ull) & ~((1ull << lo) - 1ull);
        }
        if (wb) {
            const uint32_t lo = pb + 1u > hb ? pb + 1u - hb : 0u, hi = b_hi - hb + 1u < 64u ? b_hi - hb + 1u : 64u;  // rows [hb + lo, hb + hi)
            mb = differs(hb, eb, b0, b1) & (hi >= 64u ? ~0ull : (1ull << hi) - 1ull) & ~((1ull << lo) - 1ull);
        }
        // ---- MEM numbers: above first (lanes, then bits, ascending), then below (lanes ascending = rows descending) ---------
        const uint32_t ca = (uint32_t)__popcll(ma), cb = (uint32_t)__popcll(mb);
        uint32_t sa = ca, sb = cb, sw = ca + cb;  // inclusive sums over the lanes of the group (sa, sb) and of the wave (sw)
#pragma unroll
        for (uint32_t d = 1; d < 16u; d <<= 1) {
            const uint32_t xa = __shfl_up(sa, d, 16), xb = __shfl_up(sb, d, 16);
            if (gl >= d) { sa += xa; sb += xb; }
        }
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t xw = __shfl_up(sw, d);
            if (lane >= d) sw += xw;
        }
        const uint32_t ta = __shfl(sa, 15, 16), tb = __shfl(sb, 15, 16);       // the group's totals
        const uint32_t need = (uint32_t)__builtin_amdgcn_readlane((int)sw, 63);   // the wave's
        if (need != 0u) {
            if (ovf_left < need) {  // (wave-uniform) a new chunk; what is left of the old one stays marked as unused
                const uint32_t chunk = need > kOvfChunk ? need : kOvfChunk;
                unsigned long long base = 0ull;
                if (lane == 0u) base = atomicAdd(A.total, (unsigned long long)chunk);
                ovf_base = u64_of((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base),
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)));
                ovf_left = chunk;
            }
            unsigned long long slot = ovf_base + (sw - ca - cb);
            uint32_t kk = k + (sa - ca);
            for (uint64_t m = ma; m != 0ull; m &= m - 1ull, slot++, kk++) {
                const uint32_t row = ha + (uint32_t)__builtin_ctzll(m);
                if (kk >> 28) atomicOr(reinterpret_cast<unsigned int*>(A.total) + 9, 1u);
                if (slot < A.capacity) {
                    A.raw_key[slot] = RawKey{g, kk};
                    A.raw_mem[slot] = slamem_mem{row, pos, (uint32_t)msz};  // ref_pos holds the ROW until K9
                    A.defer.raw_seg[slot] = segabs;
                }
            }
            kk = k + ta + (sb - cb);
            for (uint64_t m = mb; m != 0ull; slot++, kk++) {
                const uint32_t bit = 63u - (uint32_t)__builtin_clzll(m);
                m &= ~(1ull << bit);
                if (kk >> 28) atomicOr(reinterpret_cast<unsigned int*>(A.total) + 9, 1u);
                if (slot < A.capacity) {
                    A.raw_key[slot] = RawKey{g, kk};
                    A.raw_mem[slot] = slamem_mem{hb + bit, pos, (uint32_t)msz};
                    A.defer.raw_seg[slot] = segabs;
                }
            }
            k += ta + tb;
            ovf_base += need;
            ovf_left -= need;
        }
        // ---- advance ----------------------------------------------------------------------------------------------------
        if (busy) {
            if (do_a) aoff = a_end - t;                          // (at least na when the rows above are through)
            if (do_b) boff = b_reach ? nb : b - (b_blk - 960u) + 1u;
            if (b_last) {  // the level is through
                bool done = !walk_up;
                if (!done) {
                    pt = t; pb = b;
                    msz = parent_from(rt, rbm, t, b);  // (slamem.c:192)
                    aoff = 0; boff = 0;
                    done = msz < L;
                }
                if (done) {
                    if (gl == 0u) A.defer.pool[segabs] = k - kbase;
                    busy = false;
                }
            }
        }
    }
#ifdef SLAMEM_DIAG_TRIPS
    if (lane == 0u) { atomicMax(A.defer.njobs + 11, dg_steps); atomicAdd(A.defer.njobs + 12, dg_steps); atomicAdd(A.defer.njobs + 13, dg_busy); }
#endif
}

// K9 for the strands with deferred jobs: the jobs' MEM counts of a strand -> what comes before each job (and before what the lane
// reported behind it), and the strand's total.  One lane per strand (a read has at most a job per position).
__global__ void __launch_bounds__(256) k_defer_prefix(SearchArgs A) {
    const uint32_t n = *A.defer.nlist;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint2 e = A.defer.list[i];
        uint32_t* c = A.defer.pool + e.y;
        const uint32_t jobs = c[0], own = c[1];
        uint32_t run = 0;
        for (uint32_t s2 = 0; s2 < jobs; s2++) { const uint32_t v = c[2u + s2]; c[2u + s2] = run; run += v; }
        c[2u + jobs] = run;
        A.block_counts[e.x] = own + run;
    }
}

// Counters of a diagnostic launch (template parameter kStats; the timed kernels are the kStats = false instantiations,
// which carry none of this): how many loads of each kind the lanes issue.  One 64-bit word each, at SearchArgs::stats.
enum : uint32_t {
    SC_FM_TOP = 0, SC_FM_BOT, SC_REC_FAIL_LINES, SC_REC_PEND_LINES, SC_REC_FLUSH_LINES, SC_QUERY_LOADS, SC_LANE_TRIPS,
    SC_WAVE_TRIPS, SC_POSITIONS, SC_ENUM_JOBS, SC_ENUM_ROW_STEPS, SC_PF_PROBES, SC_PF_QUERY_LOADS, SC_PF_ITEMS,
    SC_DIR_SA, SC_DIR_GROUPS, SC_DIR_RECS, SC_DIR_QLOADS, SC_DIR_LETTERS, SC_JUMP_LINES,
    SC_SKIP_GROUPS, SC_SKIP_QLOADS, SC_SKIP_PROBES, SC_SKIP_OK, SC_T_FIRST, SC_T_DRAIN, SC_T_LAST, SC_T_WAVE_SUM, SC_ENUM_LEVELS, SC_T_ENUM_SUM, SC_STATE_TRIPS /* 11 words: lane trips per state */, SC_STATE_WAVES = SC_STATE_TRIPS + 11 /* 11 words: wave trips in which some lane is in the state */,
    SC_SEED_WINDOWS = SC_STATE_WAVES + 11, SC_SEED_COMPARES, SC_SEED_NMASKS, SC_SEED_MEMS, SC_SEED_LEFT, SC_SEED_READS, SC_SEED_QBYTES, SC_SEED_WHY /* 7 words: why a read / strand was left */, SC_SEED_ONCE = SC_SEED_WHY + 7, SC_COUNT
};
template <bool kStats>
__device__ __forceinline__ void stat_flush(unsigned long long* dst, uint32_t v) {
    if (!kStats) return;
    // wave-reduced: one atomic per wave and counter
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63u) == 0u && v) atomicAdd(dst, (unsigned long long)v);
}

// K7q: the strands of the batch as 4-bit letter ids in SCAN coordinates -- position p of the forward strand is letter p
// of the record, position p of the reverse strand is the complement of letter len-1-p (ReverseComplementSequence,
// sequence.c:413-430, done once here instead of per letter in the scan) -- 16 letters per 64-bit word, first letter in
// the top nibble (the layout of the packed text, so that a query window and a text group compare with one XOR).  Every
// strand block starts on a 16-byte boundary.  16 lanes per work item, one word each per step; ASCII -> id through a
// 256-byte table in LDS.
// kLanes lanes per work item.  long_only = false: the items of the work list (the survivors of the prefilter, or all)
// that are whole strands; long_only = true: every slice of the records that were cut into slices, whether it survived or
// not -- a neighbouring slice's scan starts in it (warm-up).
template <uint32_t kLanes>
__global__ void __launch_bounds__(256) k_pack_queries(SearchArgs A, bool long_only) {
    __shared__ uint8_t lut[512];  // [0,256): id; [256,512): id of the complement
    for (uint32_t i = threadIdx.x; i < 256u; i += 256u) {
        uint32_t c = ascii_code_q(i);
        lut[i] = (uint8_t)c;
        lut[256u + i] = (uint8_t)(c >= 2u ? 7u - c : c);  // A<->T, C<->G; N stays N  (sequence.c:419-426)
    }
    __syncthreads();
    const uint32_t sub = threadIdx.x & (kLanes - 1u);
    const uint64_t count = long_only ? A.num_items : (A.work_ids ? (uint64_t)*A.work_count : A.num_items);
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) / kLanes;
    // grid-stride over the list (the table above is set up once per block)
    for (uint64_t e = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kLanes; e < count; e += stride) {
    const uint64_t item = (!long_only && A.work_ids) ? (uint64_t)A.work_ids[e] : e;
    // the descriptor and the strand's place in the packed copy in ONE round trip (as a struct load the length came first,
    // for the test below, and the rest a round trip later: five dependent memory phases per item, now three)
    ItemDesc d;
    uint64_t pk_word;
    if (A.implicit_items) {
        if (!long_only && A.item_alive && A.item_alive[item] == 0) continue;  // (proved empty by the presence filter: K8 passes over it)
        d = item_of(A, item); pk_word = item_pk_of(A, item, d);
    }
    else {
        uint4 dv = *reinterpret_cast<const uint4*>(A.items + item);
        pk_word = A.item_pk[item];
        asm volatile("" : "+v"(dv.x), "+v"(dv.y), "+v"(dv.z), "+v"(dv.w), "+v"(pk_word));
        d.base = u64_of(dv.x, dv.y); d.len = dv.z; d.slice_rev = dv.w;
    }
    if (long_only != (d.len > kSliceLen)) continue;
    const uint32_t rev = d.slice_rev >> 31, sl = d.slice_rev & 0x7FFFFFFFu;
    const uint32_t a = sl * kSliceLen;
    const uint32_t b = d.len - a < kSliceLen ? d.len : a + kSliceLen;
    uint64_t* out = A.pq_out + pk_word;
    uint32_t* out2 = A.pq2_out ? reinterpret_cast<uint32_t*>(A.pq2_out + (pk_word >> 1)) : nullptr;
    const uint32_t w0 = a >> 4;
    // the last slice also writes the zero padding up to the strand's 16-byte-aligned end
    const uint32_t w1 = b == d.len ? 2u * ((d.len + 31u) >> 5) : (b >> 4);
    const uint8_t* q = reinterpret_cast<const uint8_t*>(A.qwords);
    const uint8_t* tab = lut + (rev ? 256u : 0u);
    for (uint32_t w = w0 + sub; w < w1; w += kLanes) {
        uint32_t whi = 0, wlo = 0;  // the word's two halves: letters 0..7 and 8..15
        const uint32_t p0 = w * 16u;
        if (p0 < d.len) {
            const uint32_t cnt = d.len - p0 < 16u ? d.len - p0 : 16u;
            // bytes of strand positions p0 .. p0+cnt-1: ascending addresses (forward) or descending (reverse)
            const uint64_t first = rev ? d.base + (uint64_t)(d.len - p0 - cnt) : d.base + p0;  // lowest address read
            const uint64_t al = first & ~7ull;
            const uint32_t sh = (uint32_t)(first - al) * 8u;
            const uint64_t* src = reinterpret_cast<const uint64_t*>(q + al);
            const uint64_t last = first + cnt - 1u;  // highest address read
            uint64_t x0 = src[0], x1 = (al + 8u <= last) ? src[1] : 0ull, x2 = (al + 16u <= last) ? src[2] : 0ull;
            uint64_t lo = sh ? (x0 >> sh) | (x1 << (64u - sh)) : x0;   // bytes first .. first+7
            uint64_t hi = sh ? (x1 >> sh) | (x2 << (64u - sh)) : x1;   // bytes first+8 .. first+15
            const uint32_t d0 = (uint32_t)lo, d1 = (uint32_t)(lo >> 32), d2 = (uint32_t)hi, d3 = (uint32_t)(hi >> 32);
            if (cnt == 16u) {  // full word: constant shifts
#define SLAMEM_ID(dw, k) ((uint32_t)tab[((dw) >> (8 * (k))) & 0xFFu])
                if (!rev) {
                    whi = SLAMEM_ID(d0, 0) << 28 | SLAMEM_ID(d0, 1) << 24 | SLAMEM_ID(d0, 2) << 20 | SLAMEM_ID(d0, 3) << 16 |
                          SLAMEM_ID(d1, 0) << 12 | SLAMEM_ID(d1, 1) << 8 | SLAMEM_ID(d1, 2) << 4 | SLAMEM_ID(d1, 3);
                    wlo = SLAMEM_ID(d2, 0) << 28 | SLAMEM_ID(d2, 1) << 24 | SLAMEM_ID(d2, 2) << 20 | SLAMEM_ID(d2, 3) << 16 |
                          SLAMEM_ID(d3, 0) << 12 | SLAMEM_ID(d3, 1) << 8 | SLAMEM_ID(d3, 2) << 4 | SLAMEM_ID(d3, 3);
                } else {  // strand position i is byte 15-i
                    whi = SLAMEM_ID(d3, 3) << 28 | SLAMEM_ID(d3, 2) << 24 | SLAMEM_ID(d3, 1) << 20 | SLAMEM_ID(d3, 0) << 16 |
                          SLAMEM_ID(d2, 3) << 12 | SLAMEM_ID(d2, 2) << 8 | SLAMEM_ID(d2, 1) << 4 | SLAMEM_ID(d2, 0);
                    wlo = SLAMEM_ID(d1, 3) << 28 | SLAMEM_ID(d1, 2) << 24 | SLAMEM_ID(d1, 1) << 20 | SLAMEM_ID(d1, 0) << 16 |
                          SLAMEM_ID(d0, 3) << 12 | SLAMEM_ID(d0, 2) << 8 | SLAMEM_ID(d0, 1) << 4 | SLAMEM_ID(d0, 0);
                }
#undef SLAMEM_ID
            } else {
                for (uint32_t i = 0; i < cnt; i++) {
                    uint32_t bi = rev ? cnt - 1u - i : i;  // byte index (from `first`) of strand position p0+i
                    uint32_t dw = bi < 4u ? d0 : bi < 8u ? d1 : bi < 12u ? d2 : d3;
                    uint32_t id = tab[(dw >> (8u * (bi & 3u))) & 0xFFu];
                    if (i < 8u) whi |= id << (28u - 4u * i); else wlo |= id << (28u - 4u * (i - 8u));
                }
            }
        }
        out[w] = u64_of(wlo, whi);
        if (out2) {
            // the same 16 letters at 2 bits each ((id - 2) & 3) in the half of the 64-bit word they belong to -- letters
            // 0..15 of a 32-letter word are its high half -- and a flag for the item if one of them is not A,C,G,T
            uint32_t c = (whi + 0x66666666u) & 0x33333333u, e = (wlo + 0x66666666u) & 0x33333333u;
            c = (c | (c >> 2)) & 0x0F0F0F0Fu; c = (c | (c >> 4)) & 0x00FF00FFu; c = (c | (c >> 8)) & 0x0000FFFFu;
            e = (e | (e >> 2)) & 0x0F0F0F0Fu; e = (e | (e >> 4)) & 0x00FF00FFu; e = (e | (e >> 8)) & 0x0000FFFFu;
            out2[2u * (w >> 1) + ((w & 1u) ^ 1u)] = (c << 16) | e;
            const uint32_t cntw = p0 < d.len ? (d.len - p0 < 16u ? d.len - p0 : 16u) : 0u;
            const uint32_t vhi = ((whi >> 1) | (whi >> 2) | (whi >> 3)) & 0x11111111u, vlo = ((wlo >> 1) | (wlo >> 2) | (wlo >> 3)) & 0x11111111u;
            const uint32_t nvalid = (uint32_t)__popc(vhi) + (uint32_t)__popc(vlo);
            if (nvalid != cntw || long_only) A.item_flags[item] = 1;  // (slices of long records: no skipping, for now)
        }
    }
    }
}

// Letters of one strand from the packed copy (k_pack_queries): a window of 32 letters (one aligned 16-byte load) in
// registers, 5 loads per strand of 150 letters.
struct PackedCursor {
    const uint4* p;   // the strand's first word (16-byte aligned)
    uint32_t cidx;    // window held, ~0 = none
    uint64_t q0, q1;
    __device__ __forceinline__ void init(const uint64_t* base) { p = reinterpret_cast<const uint4*>(base); cidx = ~0u; q0 = q1 = 0; }
    __device__ __forceinline__ void forget() { cidx = ~0u; }
    __device__ __forceinline__ uint32_t would_load(uint32_t pos) const { return (pos >> 5) != cidx ? 1u : 0u; }
    __device__ __forceinline__ uint32_t at(uint32_t pos) {
        uint32_t chunk = pos >> 5;
        if (chunk != cidx) {
            uint4 a = p[chunk];
            q0 = u64_of(a.x, a.y); q1 = u64_of(a.z, a.w);
            cidx = chunk;
        }
        uint64_t m16 = (pos & 16u) ? ~0ull : 0ull;
        uint64_t w = q0 ^ ((q0 ^ q1) & m16);
        return (uint32_t)(w >> (60u - 4u * (pos & 15u))) & 15u;
    }
};

// States of a lane.  EXT / REC / FLUSH: the index walk.  DSA / DIR / DEND: direct extension of a match that has become
// ONE row, i.e. one text position r = SA[row]: the query is compared with the text itself, one TextGroup (16 letters +
// the parent-depth classes that tell where an ancestor interval may have to be reported) per trip; where the run ends
// (a letter disagrees, a class says stop, the slice or the text begins) the TextRec of the position gives back the row
// and -- after a disagreeing letter -- the parent interval on which the letter is retried.  The reference takes these
// letters one FMI_FollowLetter at a time (slamem.c:121); the output is the same: as long as the letters agree the single
// row's BWT letter IS the query letter (nothing is left-maximal) and only an ancestor >= min_len could emit.
// JQ / JT: the first K letters of a scan through the K-mer jump table (query words, then the table entry).
// SKV / SKQ / SKP: skipping the stretch of chance matches behind a disagreeing letter.  A direct run that ends on a
// disagreeing letter at strand position p (text position s) is followed, in the reference's scan, by ~log4(n) positions
// whose longest matches are chance k-mers (1.9 failed extensions each: most of the kernel's lines).  They cannot emit
// anything, and the scan is back on the same diagonal afterwards, IF (SKV) the next w = min_len-1 letters of the diagonal
// agree, Q[p-w..p) = T[s-w..s), and the suffix at s-w shares fewer than w letters with its neighbours in suffix order
// (parent-depth class), and (SKQ, SKP) no min_len-mer window of the strand that covers p occurs in the text -- every such
// window contains one of the k-mers that start at multiples of s1 = min_len-k+1 in [p-min_len+1, p+s1-1], and the
// occurrence bitmap says that none of them occurs.  Then every position j in [p-w, p] has a longest match shorter than
// min_len (it would be such a window), so nothing is emitted there; and at j = p-w the longest match is exactly the
// diagonal's w letters, in ONE row: the direct run goes on from (j, text position s-w, depth w) without touching the
// index.  Anything else (a second disagreeing letter, a present k-mer, an N, a deep class, too little room) takes the
// normal route (DEND, then the index walk).
enum : uint32_t { ST_EXT = 0, ST_REC = 1, ST_FLUSH = 2, ST_DSA = 3, ST_DIR = 4, ST_DEND = 5, ST_JQ = 6, ST_JT = 7,
                  ST_SKV = 8, ST_SKQ = 9, ST_SKP = 10 };

// kMam: the instantiation for -mam on reads (slamem.c:131): a position is reported only when its interval is ONE row, and a
// position whose interval is deeper than min_len but NOT one row skips the bookkeeping of slamem.c:197-198 -- so the interval
// that a later failed extension falls back to (:122-123) is the one noted at an earlier position.  That is two more
// registers of state (prev_top, prev_bot), another address for the records of a failed extension, and one more condition on
// `pend`; everything else of the state machine is the -mem one (a direct run is a run of one-row positions: the fall-back
// interval is the row all along).  Long strands in slices stay with k_find_mams_sliced (their states are verified there).
#ifndef SLAMEM_V3_WAVES
#define SLAMEM_V3_WAVES 1
#endif
// kSkip: the instantiation that carries the skipping states (SKV / SKQ / SKP).  Measured (profiles/r02_skip_experiment.txt):
// exact, 27 % fewer lines and 32 % fewer lane trips, but the same time -- the kernel is bound by instruction issue and
// every trip pays for the union of the states its lanes are in -- so the default kernel is the one without them
// (SLAMEM_SKIP=1 selects this one).
template <bool kStats, bool kSkip, bool kSliced, bool kMam, bool kCarry, bool kChunk = false, bool kDefer = false>
__global__ void __launch_bounds__(256, (kSkip || kMam || kCarry || kChunk) ? 4 : SLAMEM_V3_WAVES) k_find_mems_v3(SearchArgs A) {
    __shared__ ItemDesc lds_item[4][kFetch];
    __shared__ uint64_t lds_pk[4][kFetch];
    __shared__ uint32_t lds_id[4][kFetch];
    __shared__ uint32_t lds_q2[kSkip ? 6 : 1][kSkip ? 256 : 1];  // skipping: the lane's three 2-bit strand words as six 32-bit halves
    const IndexView& ix = A.ix;
    const int L = (int)A.min_len;
    const uint4* R = reinterpret_cast<const uint4*>(ix.rec);
    // the work list: every item, or only those that survived the prefilter (dense, so no lane idles on dead items).
    // Waves take it in pieces of kFetch items from a global cursor (one atomic per piece), so that every wave stays
    // busy until the list is empty whatever its strands cost -- with a fixed share per wave, lanes idled at the end
    // of every share (measured: 58 % lane use once the direct extension made strand costs uneven).
    const uint32_t nitems = A.work_ids ? *A.work_count : (uint32_t)A.num_items;  // (the host refuses batches near 2^32 items)
    // (an empty list -- K8s left no strand: every batch of plain reads -- and no lanes to take in: 4,096 waves have nothing to
    //  queue up for at the cursor's one address)
    if (!kCarry && !kStats && nitems == 0u) return;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t next = 0, chunk_first = 0, chunk_end = 0;  // wave-uniform: the piece being handed out
    uint32_t seen = 0;                                  // wave-uniform: how far this wave has seen the cursor get
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);  // waves that take work from the list
    bool drained = false;                               // wave-uniform: the cursor is past the end of the list
    // direct extension: the class threshold of this launch (flag <=> class >= cL; a parent depth >= L implies it)
    const int dmin = A.direct_min_depth;  // < 0: off
    const uint32_t cL = depth_class(L);
    const uint64_t k1 = 0x1111111111111111ull;
    const uint64_t cls_add = (uint64_t)(8u - (cL & 7u)) * k1;
    const bool cls_hi = cL >= 8u;
    const uint32_t jK = (ix.kjump && (int)ix.kjump_k < L && A.use_jump) ? ix.kjump_k : 0u;
    const uint32_t skw = kSkip ? A.skip_w : 0u, sks1 = A.skip_s1, kd = ix.kbits_k;  // skipping (skw = 0: off)
    const uint32_t sknp = skw ? (kd - 1u + sks1 - 1u) / sks1 + 1u : 0u;  // probes per disagreeing letter
    const uint32_t cW = depth_class((int)skw);
    const uint64_t clsw_add = (uint64_t)(8u - (cW & 7u)) * k1;
    const bool clsw_hi = cW >= 8u;

    // diagnostic instantiation only: when the first wave started, when the first wave found the list empty, when the last
    // wave left (100 MHz wall clock), and the sum over waves of the time they ran
    const unsigned long long t_wave0 = kStats ? wall_clock64() : 0ull;
    if (kStats && (threadIdx.x & 63u) == 0u) atomicMin(A.stats + SC_T_FIRST, t_wave0);
    uint32_t n_kt = 0, n_kb = 0, n_rec_fail = 0, n_rec_pend = 0, n_rec_flush = 0, n_trips = 0, n_wtrips = 0, n_pos = 0,
             n_enum = 0, n_qloads = 0, n_dsa = 0, n_dgrp = 0, n_drec = 0, n_dlet = 0, n_jump = 0, n_skv = 0, n_skq = 0,
             n_skp = 0, n_skok = 0, n_erow = 0, n_elev = 0;
    unsigned long long t_enum = 0ull;  // diagnostic instantiation: wall clock this wave spent in enumeration jobs
    uint32_t n_state[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, n_wstate[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // lane / wave trips per state

    bool active = false, pend = false, dmis = false, dcool = false;
    bool old = false;  // kCarry: the lane came in from the previous launch and writes to that batch's output side
    uint32_t st = ST_EXT;
    uint32_t g = 0, j = 0, top = 0, bot = 0, k = 0;
    uint32_t a_pos = 0, b_pos = 0, attempt = 0, qlen = 0;  // emitted slice [a_pos, b_pos) of the strand, warm-up attempt
    uint32_t prev_top = 0, prev_bot = 0;  // kMam: the interval a failed extension falls back to (slamem.c:122-123)
    uint32_t dir_r = 0;  // direct extension: text position where the current match starts
    int depth = 0, pub = -1;
    PackedCursor qc;
    qc.init(A.pq);
    // FM block of `top`, kept across trips: after a parent step the widened interval usually still lies in the
    // same 128-row block, so the retry fetches nothing
    Blk kt;
    kt.a = kt.b = kt.c = kt.d = make_uint4(0, 0, 0, 0);
    uint32_t tag_t = 0xFFFFFFFFu;
    // kChunk: the chunk of the overflow list this wave hands out (wave_emit_step), in LDS between the enumeration jobs.  Every
    // lane stores the same wave-uniform value: a store by lane 0 alone would be invisible to the compiler's per-thread view of
    // the other lanes, which could keep a stale copy
    __shared__ unsigned long long lds_ovf_base[kChunk ? 4 : 1];
    __shared__ uint32_t lds_ovf_left[kChunk ? 4 : 1];
    if (kChunk) { lds_ovf_base[wv] = 0ull; lds_ovf_left[wv] = 0u; }

    if (kCarry && A.carry_in) {  // the unfinished lanes of the previous launch first (the grid always covers them)
        const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
        if (t < *A.carry_in_count) {
            const uint4* r = reinterpret_cast<const uint4*>(A.carry_in + t);
            const uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
            st = r0.x & 0xFFu; pend = (r0.x >> 8) & 1u; dmis = (r0.x >> 9) & 1u; dcool = (r0.x >> 10) & 1u;
            g = r0.y; j = r0.z; top = r0.w;
            bot = r1.x; k = r1.y; qlen = r1.z; dir_r = r1.w;
            depth = (int)r2.x; pub = (int)r2.y;
            qc.init(reinterpret_cast<const uint64_t*>(u64_of(r2.z, r2.w)));
            if (kMam) { prev_top = r3.x; prev_bot = r3.y; }
            a_pos = 0; b_pos = qlen; attempt = 0;
            active = true; old = true;
        }
    }

    // kDefer: jobs of this lane's strand that went to the queue, and the strand's place in the pool (0: none yet, ~0: refused --
    // the pool was full: the strand's jobs run in the wave as in the other instantiations)
    uint32_t seg = 0, jobbase = 0;
#ifdef SLAMEM_DIAG_TRIPS
    uint32_t dg_trips = 0, dg_wtrips = 0;
#endif
    // (kDefer) a MEM the lane reports itself behind its strand's first queued job waits in these registers for the end of the trip,
    // where the wave gives the trip's records places from its chunk of the list (one atomic per 128 places: with an atomic per
    // record on the list's one counter -- and another per queued job -- the waves of a repeat-rich batch spent their time in the
    // queue of that counter: 13 us per trip); a second one in the same trip (a strand's last position) takes the counter
    bool em_on = false;
    uint32_t em_row = 0, em_pos = 0, em_len = 0, em_k = 0;
#define SLAMEM_EMIT(row_, pos_, len_)                                                                                         \
    do {                                                                                                                      \
        if (kDefer && seg != 0u) {                                                                                             \
            if (!em_on) { em_on = true; em_row = (row_); em_pos = (pos_); em_len = (len_); em_k = k; }                         \
            else emit_provisional(A, g, k, (row_), (pos_), (len_), jobbase + 2u + seg);                                        \
        } else emit3_sel<kCarry>(A, old, g, k, attempt << 28, (row_), (pos_), (len_));                                         \
    } while (0)
    __shared__ uint32_t lds_q_base[kDefer ? 4 : 1], lds_q_left[kDefer ? 4 : 1];  // the wave's chunk of the job queue
    if (kDefer) { lds_q_base[wv] = 0u; lds_q_left[wv] = 0u; }

    for (;;) {
        // ---- hand the next items to idle lanes ----------------------------------------------------------------
        unsigned long long idle = __ballot(!active);
        if (idle != 0ull && next >= chunk_end && !drained) {  // fetch the next piece of the work list
            // guided self-scheduling: pieces of kFetch items while the list is long, smaller ones (down to 8) towards its
            // end -- the items a wave has fetched but not yet started are captive to it, and at the end of the list they
            // are what the other, drained, waves wait for (measured on a 1 M-read batch: +38 % with fixed pieces of 64)
            uint32_t want = (nitems - seen) / (2u * nwaves) & ~7u;
            want = want < 8u ? 8u : want > kFetch ? kFetch : want;
            // a list shorter than one full piece per wave (the strands K8s left, a small batch): whole pieces at once, so that
            // few, full waves do the work and the others leave -- a wave trip costs the same instructions whatever its lanes do,
            // and with the guided sizes every one of 4,096 waves ran a quarter full: 29 % lane use on a repeat-rich text (round 4)
            // (a list of fewer than 16 strands per wave is a matter of latency, not of issue slots: spread thin, as the guided sizes do)
            if (seen == 0u && nitems < kFetch * nwaves && nitems >= 16u * nwaves) want = kFetch;
            uint32_t base = 0;
            if (lane == 0u) base = atomicAdd(A.work_cursor, want);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= nitems) {
                drained = true;
                if (kStats && lane == 0u) atomicMin(A.stats + SC_T_DRAIN, (unsigned long long)wall_clock64());
            } else {
                chunk_first = next = base;
                chunk_end = nitems - base < want ? nitems : base + want;
                seen = chunk_end;
                const uint32_t i = chunk_first + lane;
                if (i < chunk_end) {  // descriptors -> LDS: one coalesced read per wave instead of a round trip per item
                    uint32_t id = A.work_ids ? A.work_ids[i] : (uint32_t)i;
                    lds_id[wv][lane] = id;
                    ItemDesc idesc = item_of(A, id);
                    if (A.implicit_items && A.item_alive && A.item_alive[id] == 0) idesc.len = 0;  // (the list K8s wrote, less what the presence filter proved empty)
                    lds_item[wv][lane] = idesc;
                    lds_pk[wv][lane] = item_pk_of(A, id, idesc);
                }
                // written and read by lanes of this wave only
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (idle != 0ull && next < chunk_end) {
            uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            uint32_t cand = next + rank;
            if (!active && cand < chunk_end) {
                g = lds_id[wv][cand - chunk_first];
                ItemDesc d = lds_item[wv][cand - chunk_first];
                qc.init(A.pq + lds_pk[wv][cand - chunk_first]);
                qlen = d.len;
                uint32_t sl = d.slice_rev & 0x7FFFFFFFu;
                a_pos = sl * kSliceLen;
                b_pos = d.len - a_pos < kSliceLen ? d.len : a_pos + kSliceLen;
                attempt = 0;
                j = d.len - b_pos < kWarmUp ? d.len : b_pos + kWarmUp;  // scan start e (one past the first position)
                top = 0; bot = ix.n; depth = 0; pub = -1; pend = false; st = ST_EXT; k = 0; dcool = false;
                if (kDefer) { seg = 0; jobbase = 0; }
                if (kCarry) old = false;
                if (kMam) { prev_top = 0; prev_bot = ix.n; }
                // the first K letters through the jump table: no position that shallow can emit (K < L), and the scan
                // must have more than K letters before the slice ends
                if (jK != 0u && j - a_pos > jK) st = ST_JQ;
                // kSliced: the instantiation for batches with a record longer than a slice.  Every slice but the strand's
                // rightmost starts at b_pos from the state the full scan has there (k_slice_states): exact from the first
                // position, so "attempt" is set to the one that never restarts.  (The instantiation without it is the
                // headline workload's kernel: its code is not touched by the slice logic.)
                if (kSliced && A.slice_state && d.len - a_pos > kSliceLen) {
                    const uint4 ss = *reinterpret_cast<const uint4*>(A.slice_state + ((uint64_t)g - 1u - A.item_block[g]));
                    if (ss.w) { j = b_pos; top = ss.x; bot = ss.y; depth = (int)ss.z; pub = depth - 1; attempt = kMaxAttempt; st = ST_EXT; }
                }
                if (d.len == 0) {  // empty record: nothing to scan
                    A.block_counts[g] = 0;
                    A.item_attempt[g] = 0;
                } else active = true;
            }
            next += (uint32_t)__popcll(idle);
            // the LDS slots are reused by the next fetch: every lane has copied its descriptor by then (same wave, in order)
        }
        if (kCarry && A.carry_out && drained && next >= chunk_end && __ballot(active && old) == 0ull) {
            // the list is empty and this wave has handed out all it fetched: its lanes do not run their strands to the end
            // here (K8's tail, ~1 ms of a chip that empties), they go on in the next launch.  Lanes that came in from the
            // previous launch are never passed on a second time (the wave stays until they are through: a strand's time)
            const unsigned long long am = __ballot(active);
            if (am != 0ull) {
                unsigned int base = 0;
                if (lane == 0u) base = atomicAdd(A.carry_out_count, (unsigned int)__popcll(am));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (active) {
                    uint4* w = reinterpret_cast<uint4*>(A.carry_out + base + (uint32_t)__popcll(am & ((1ull << lane) - 1ull)));
                    const uint64_t qp = reinterpret_cast<uint64_t>(qc.p);
                    w[0] = make_uint4(st | (pend ? 0x100u : 0u) | (dmis ? 0x200u : 0u) | (dcool ? 0x400u : 0u), g, j, top);
                    w[1] = make_uint4(bot, k, qlen, dir_r);
                    w[2] = make_uint4((uint32_t)depth, (uint32_t)pub, (uint32_t)qp, (uint32_t)(qp >> 32));
                    w[3] = make_uint4(prev_top, prev_bot, 0u, 0u);
                }
            }
            break;
        }
        if (__ballot(active) == 0ull) {
            if (drained && next >= chunk_end) break;
            continue;
        }
        if (kStats) {
            n_wtrips += lane == 0u;
#pragma unroll
            for (uint32_t q = 0; q < 11u; q++)
                if (__ballot(active && st == q) != 0ull && lane == 0u) n_wstate[q]++;
        }
        // enumeration job of this trip (rare): rows that the lane does not emit itself
        bool e_on = false, e_level0 = false, e_up = false;
        uint32_t e_pos = 0, e_left = 0;
        bool consumed = false, finished = false;

        if (active) {
            // ---- memory phase: every load of this trip, no use in between -------------------------------------
            // two generic 16-byte slots, addressed by state: the row records of `top` and `bot`; or the suffix-array
            // quad of `top`; or a text group and the query words that face it; or the text-ordered record
            const uint4* a1 = nullptr;
            const uint2* a2 = nullptr;
            const uint2* a3 = nullptr;    // (SKQ only: a third 8-byte word)
            const uint8_t* a4 = nullptr;  // (SKQ only: the item's flag)
            uint32_t pk_bits = 0, pk_mask = 0, flag8 = 0;  // (SKP only: bit positions of the probes, which exist)
            uint4 rt = make_uint4(0, 0, 0, 0);
            uint2 rb0 = make_uint2(0, 0), rb1 = rb0;
            bool want_rec = false;
            uint32_t c = 0, dm = 0;
            int qs = 0;
            Blk kb;  // block of bot+1 when it differs from top's (wide intervals only: not kept across trips)
            kb.a = kb.b = kb.c = kb.d = make_uint4(0, 0, 0, 0);
            if (st == ST_EXT) {
                // records together with the blocks when a pending position's parent may still be >= min_len deep
                want_rec = pend && pub >= L;
                if (kStats && want_rec) n_rec_pend += 1u + ((top >> 2) != (bot >> 2));
            } else if (st == ST_DSA) {
                a1 = reinterpret_cast<const uint4*>(ix.sa + (top & ~3u));  // the aligned quad that holds SA[top]
                if (kStats) n_dsa++;
            } else if (st == ST_DIR || (kSkip && st == ST_SKV)) {
                // DIR: the letters below text position dir_r face the strand positions below j.  SKV: the same comparison one
                // letter further down (behind the disagreeing letter), `pub` letters of the diagonal already looked at
                const uint32_t back = (kSkip && st == ST_SKV) ? 1u + (uint32_t)pub : 0u;
                const uint32_t dirp = dir_r - back, jp = j - back;
                const uint32_t gi = (dirp - 1u) >> 4;  // dirp >= 1 here
                dm = dirp - (gi << 4);                 // letters of the group below dirp: 1..16
                a1 = reinterpret_cast<const uint4*>(ix.tgrp + gi);
                qs = (int)jp - (int)dm;                // strand position that faces the group's first letter (may be < 0)
                a2 = reinterpret_cast<const uint2*>(reinterpret_cast<const uint64_t*>(qc.p) + (qs >> 4));
                if (kStats) { if (st == ST_SKV) n_skv++; else n_dgrp++; }
            } else if (st == ST_DEND) {
                a1 = reinterpret_cast<const uint4*>(ix.prec + dir_r);
                if (kStats) n_drec++;
            } else if (st == ST_JQ) {
                qs = (int)(j - jK);  // the scan's first K letters: strand positions j-K .. j-1
                a2 = reinterpret_cast<const uint2*>(reinterpret_cast<const uint64_t*>(qc.p) + (qs >> 4));
            } else if (st == ST_JT) {
                a2 = ix.kjump + dir_r;  // (dir_r holds the key between the two trips; one 8-byte entry)
                if (kStats) n_jump++;
            } else if (kSkip && st == ST_SKQ) {
                // the strand at 2 bits per letter around p = j-1: the probed k-mers all contain p, so they lie in
                // [p-k+1, p+k): at most 62 letters from the start of the word that holds p-k+1 -- two words, a third for safety
                const uint32_t y0 = j - kd;  // = p - (k-1)
                const uint64_t* p2 = A.pq2 + ((reinterpret_cast<const uint64_t*>(qc.p) - A.pq) >> 1) + (y0 >> 5);
                a2 = reinterpret_cast<const uint2*>(p2);
                a3 = reinterpret_cast<const uint2*>(p2 + 2);
                a4 = A.item_flags + g;
                if (kStats) n_skq++;
            } else if (kSkip && st == ST_SKP) {
                // (the probes are issued further down, beside the FM blocks: they land in the same registers)
            } else {
                want_rec = true;  // ST_REC, ST_FLUSH
            }
            if (kStats) {
                n_trips++;
                if (st < 11u) n_state[st]++;
                if (st == ST_REC) n_rec_fail += 1u + ((top >> 2) != (bot >> 2));
                if (st == ST_FLUSH) n_rec_flush += 1u + ((top >> 2) != (bot >> 2));
            }
            if (want_rec) {
                // (kMam, state REC: the records of the fall-back interval; with a pending position it IS [top,bot])
                const bool of_prev = kMam && st == ST_REC;
                a1 = R + (of_prev ? prev_top : top);
                a2 = reinterpret_cast<const uint2*>(R + (of_prev ? prev_bot : bot));
            }
            if (a1) rt = *a1;
            if (a2) { rb0 = a2[0]; if (st != ST_JT) rb1 = a2[1]; }
            if (a3) { const uint2 t3 = a3[0]; rt.x = t3.x; rt.y = t3.y; }
            if (a4) flag8 = *a4;
            if (st == ST_EXT) {
                if (kStats) n_qloads += qc.would_load(j - 1u);
                c = qc.at(j - 1u);  // issues the query-window load (if any) behind the ones above
                // The FM blocks LAST.  The compiler rearranges the registers of the second block as soon as they are loaded --
                // s_waitcnt inside this branch, taken in 94 % of the wave trips (some lane's interval spans two blocks) --
                // and with the blocks first, the record / text / query loads of the other lanes were issued only after
                // those waits: two memory phases per trip.  In this order every load of the trip is in flight before the
                // first wait (checked in the ISA: twelve loads back to back): K8 21.76 -> 20.65 ms.
                uint32_t bt = top >> kFmRowsLog2, bb = (bot + 1u) >> kFmRowsLog2;
                if (kStats) { n_kt += bt != tag_t; n_kb += bb != bt; }
                if (bt != tag_t) { kt = load_blk(ix.fm, bt); tag_t = bt; }
                if (bb != bt) kb = load_blk(ix.fm, bb);
            } else if (kSkip && st == ST_SKP) {
                // the probed k-mers: starts p-(k-1), p-(k-1)+s1, ... and p itself -- every one contains p, consecutive starts
                // are at most s1 = min_len-k+1 apart, so every min_len-mer window that covers p contains one of them
                // (the lane's 2-bit words wait in LDS since the last trip: a k-mer of 16 letters or fewer lies in two
                //  consecutive 32-bit halves)
                const uint32_t y0 = j - kd;
                const uint32_t o0 = y0 & 31u;
                // (at most four probes -- the host side checks: their words land in the record slots rt / rb0 / rb1, which
                //  this state does not use otherwise, so they share no register with the FM blocks of the other lanes)
#pragma unroll
                for (uint32_t i = 0; i < 4u; i++) {
                    if (i < sknp) {
                        uint32_t d = i * sks1;
                        if (d > kd - 1u) d = kd - 1u;
                        const uint32_t o = o0 + d;  // letter offset from the start of the first word (< 32 + k)
                        const uint32_t h0 = lds_q2[o >> 4][threadIdx.x], h1 = lds_q2[(o >> 4) + 1u][threadIdx.x];
                        const uint32_t sh = 2u * (o & 15u);
                        const uint32_t v32 = sh ? (h0 << sh) | (h1 >> (32u - sh)) : h0;
                        const uint32_t key = v32 >> (32u - 2u * kd);
                        pk_bits |= (key & 63u) << (6u * i);
                        pk_mask |= 1u << i;
                        const uint2 wv = *reinterpret_cast<const uint2*>(ix.kbits + (key >> 6));
                        if (i == 0) { rt.x = wv.x; rt.y = wv.y; } else if (i == 1) { rt.z = wv.x; rt.w = wv.y; }
                        else if (i == 2) rb0 = wv; else rb1 = wv;
                        if (kStats) n_skp++;
                    }
                }
            }
            if (kSkip) {  // (as for kMam below: nothing of these is touched before every load of the trip is out)
                asm volatile("" : "+v"(rt.x), "+v"(rt.y), "+v"(rt.z), "+v"(rt.w), "+v"(flag8));
                asm volatile("" : "+v"(rb0.x), "+v"(rb0.y), "+v"(rb1.x), "+v"(rb1.y));
            }
            if (kMam || kChunk) {
                // keeps the compiler from copying parts of the record out of its load's registers right behind the load (an
                // s_waitcnt between the loads of one trip: seen in these instantiations' ISA): the values "change" here
                asm volatile("" : "+v"(rt.x), "+v"(rt.y), "+v"(rt.z), "+v"(rt.w));
            }
            const uint4 rb = make_uint4(rb0.x, rb0.y, rb1.x, rb1.y);

            // ---- compute phase -------------------------------------------------------------------------------
            bool strand_end = false;  // position 0 has been consumed: flush what is pending, finish
            bool pub_exact = false;   // strand end reached through DEND: pub is the exact parent depth
            if (st == ST_FLUSH) {  // the strand ended on a match whose parent may qualify too: its depth has arrived
                st = ST_EXT;
                uint32_t t2 = top, b2 = bot;
                pub = parent_from(rt, rb, t2, b2);
                if (pub < L && top == bot) { SLAMEM_EMIT(top, 0u, (uint32_t)depth); k++; }
                else { e_on = true; e_level0 = true; e_up = pub >= L; e_pos = 0u; e_left = 0xFFu; }
                pend = false;
                finished = true;
            } else if (st == ST_REC) {  // a deep match ended at this letter: widen, retry next trip
                st = ST_EXT;
                dcool = false;
                if (kMam) { top = prev_top; bot = prev_bot; }  // slamem.c:122-123
                int d = parent_from(rt, rb, top, bot);
                if (d < 0) { depth = 0; pub = -1; consumed = true; }
                else { depth = d; pub = d - 1; if (kMam) { prev_top = top; prev_bot = bot; } }
            } else if (st == ST_JQ) {
                // K letters, first on top -> 2 bits each (the layout of the table's keys, index_build.hip k_kjump_keys)
                const uint64_t w0 = u64_of(rb0.x, rb0.y), w1 = u64_of(rb1.x, rb1.y);
                const uint32_t sh = ((uint32_t)qs & 15u) * 4u;
                uint64_t x = sh ? (w0 << sh) | (w1 >> (64u - sh)) : w0;
                const uint64_t v = ((x >> 1) | (x >> 2) | (x >> 3)) & k1;  // nibble >= 2: one of A,C,G,T
                const uint64_t topk = ~0ull << (4u * (16u - jK));
                if ((v & topk) == (k1 & topk)) {
                    x = (x & topk) | (0x2222222222222222ull & ~topk);
                    x = (x - 0x2222222222222222ull) & topk;
                    x = (x & 0x0303030303030303ull) | ((x & 0x3030303030303030ull) >> 2);
                    x = (x & 0x000F000F000F000Full) | ((x & 0x0F000F000F000F00ull) >> 4);
                    x = (x & 0x000000FF000000FFull) | ((x & 0x00FF000000FF0000ull) >> 8);
                    x = (x & 0xFFFFull) | ((x >> 16) & 0xFFFF0000ull);
                    dir_r = (uint32_t)x >> (2u * (16u - jK));
                    st = ST_JT;
                } else st = ST_EXT;  // an N among them: letter by letter
            } else if (st == ST_JT) {
                st = ST_EXT;
                if (rb0.x <= rb0.y) {  // the K-mer occurs: K successful extensions from the root (slamem.c:110-129)
                    top = rb0.x; bot = rb0.y; depth = (int)jK; pub = (int)jK - 1; j -= jK;
                    if (kMam) { prev_top = top; prev_bot = bot; }  // (K < min_len: every one of those positions noted its interval)
                    if (kStats) n_pos += jK;
                }
            } else if (kSkip && st == ST_SKQ) {
                if (flag8) { dmis = true; st = ST_DEND; }  // a letter that is not A,C,G,T somewhere in the strand
                else {  // (a 64-bit word holds its first 16 letters in its high half)
                    lds_q2[0][threadIdx.x] = rb0.y; lds_q2[1][threadIdx.x] = rb0.x;
                    lds_q2[2][threadIdx.x] = rb1.y; lds_q2[3][threadIdx.x] = rb1.x;
                    lds_q2[4][threadIdx.x] = rt.y; lds_q2[5][threadIdx.x] = rt.x;
                    st = ST_SKP;
                }
            } else if (kSkip && st == ST_SKP) {
                uint32_t present = 0;
                present |= (pk_mask >> 0) & (uint32_t)(u64_of(rt.x, rt.y) >> ((pk_bits >> 0) & 63u));
                present |= (pk_mask >> 1) & (uint32_t)(u64_of(rt.z, rt.w) >> ((pk_bits >> 6) & 63u));
                present |= (pk_mask >> 2) & (uint32_t)(u64_of(rb0.x, rb0.y) >> ((pk_bits >> 12) & 63u));
                present |= (pk_mask >> 3) & (uint32_t)(u64_of(rb1.x, rb1.y) >> ((pk_bits >> 18) & 63u));
                if (present & 1u) { dmis = true; st = ST_DEND; }  // one of the k-mers occurs: the normal route
                else {
                    // certified.  The position in front of the disagreeing letter is left-maximal: its one row (text position
                    // dir_r) is emitted if it is long enough (no ancestor can qualify: its class was checked when the run stopped)
                    const bool in_slice = j >= a_pos && j < b_pos;
                    if (depth >= L && in_slice) { SLAMEM_EMIT(dir_r, j, (uint32_t)depth | 0x80000000u); k++; }
                    j -= skw + 1u; dir_r -= skw + 1u; depth = (int)skw;
                    pend = false;
                    st = ST_DIR;
                    if (kStats) { n_skok++; n_pos += skw + 1u; n_dlet += skw + 1u; }
                }
            } else if (st == ST_DSA) {
                const uint32_t o = top & 3u;
                dir_r = o == 0u ? rt.x : o == 1u ? rt.y : o == 2u ? rt.z : rt.w;
                dmis = true;  // read only when the text begins here: '$' on the left never equals a query letter
                st = dir_r == 0u ? ST_DEND : ST_DIR;
            } else if (st == ST_DIR || (kSkip && st == ST_SKV)) {
                // the group's letters face strand positions qs .. qs+15, letter i in nibble 15-i.  Letters dm-1, dm-2, ... are
                // compared with the strand positions below
                const bool ver = kSkip && st == ST_SKV;
                const uint64_t T = u64_of(rt.x, rt.y), C = u64_of(rt.z, rt.w);
                const uint64_t w0 = u64_of(rb0.x, rb0.y), w1 = u64_of(rb1.x, rb1.y);
                const uint32_t sh = ((uint32_t)qs & 15u) * 4u;
                const uint64_t Q = sh ? (w0 << sh) | (w1 >> (64u - sh)) : w0;
                uint64_t x = T ^ Q;
                x |= x >> 1; x |= x >> 2;
                const uint64_t mis = x & k1;                                   // nibble LSB set: the letters differ
                const uint64_t lo3 = C & (7ull * k1), h8 = (C >> 3) & k1;
                const uint64_t ge = ((lo3 + (ver ? clsw_add : cls_add)) >> 3) & k1;
                // DIR: class >= class(min_len): an ancestor may qualify.  SKV: class >= class(w): maybe >= w letters shared
                const uint64_t flag = (ver ? clsw_hi : cls_hi) ? (h8 & ge) : (h8 | ge);
                const uint32_t nlo = 16u - dm;
                if (!ver) {
                    const uint32_t W = j - a_pos;                               // letters this item may still consume (>= 1)
                    const uint32_t take = dm < W ? dm : W;
                    const uint32_t nhi = nlo + take;                            // nibbles of the letters in play
                    uint64_t mask = nhi >= 16u ? ~0ull : ((1ull << (4u * nhi)) - 1ull);
                    mask &= ~((1ull << (4u * nlo)) - 1ull);
                    const uint64_t stop = (mis | flag) & mask;
                    uint32_t kc, flagged_stop = 0;
                    const bool hit = stop != 0ull;
                    if (hit) {
                        const uint32_t nib = (uint32_t)__builtin_ctzll(stop) >> 2;
                        kc = nib - nlo;
                        dmis = ((mis >> (4u * nib)) & 1ull) != 0ull;
                        flagged_stop = (uint32_t)((flag >> (4u * nib)) & 1ull);
                        dcool = !dmis;  // stopped by a class flag: the index walk takes over until the interval changes
                    } else {
                        kc = take;
                        dmis = false;
                    }
                    j -= kc; depth += (int)kc; dir_r -= kc;
                    if (kStats) { n_dlet += kc; n_pos += kc; }
                    if (!hit && j != a_pos) {
                        if (dir_r == 0u) { dmis = true; st = ST_DEND; }  // the text begins: the next letter cannot match
                        else st = ST_DIR;                               // the whole group agreed: next group
                    } else if (kSkip && hit && dmis && skw != 0u && flagged_stop == 0u && j - a_pos >= skw + 1u && dir_r >= skw + 2u &&
                               j - 1u + kd <= qlen) {  // (j >= k follows from the room on the left: w + 1 = min_len >= k)
                        pub = 0;       // a disagreeing letter, nothing pending can have ancestors, room on both sides: try to
                        st = ST_SKV;   // skip the chance matches behind it
                    } else {
                        st = ST_DEND;
                    }
                } else {
                    const uint32_t v = (uint32_t)pub, need = skw + 1u - v;      // w letters to agree, then one for its class
                    const uint32_t take = dm < need ? dm : need;
                    const uint32_t must = v + take <= skw ? take : take - 1u;   // how many of them must agree
                    const uint32_t nhm = nlo + must;
                    uint64_t maskm = nhm >= 16u ? ~0ull : ((1ull << (4u * nhm)) - 1ull);
                    maskm &= ~((1ull << (4u * nlo)) - 1ull);
                    bool bad = (mis & maskm) != 0ull;
                    if (must != take) bad = bad || ((flag >> (4u * (nlo + take - 1u))) & 1ull) != 0ull;
                    if (bad) { dmis = true; st = ST_DEND; }   // the normal route
                    else {
                        pub = (int)(v + take);
                        st = v + take == skw + 1u ? ST_SKQ : ST_SKV;
                    }
                }
            } else if (st == ST_DEND) {
                // rt = {row, parent top, parent bottom, parent depth + 1} of the suffix that starts at dir_r
                st = ST_EXT;
                top = bot = rt.x;
                const int pdepth = (int)rt.w - 1;
                pub = pdepth;  // exact
                const bool in_slice = j >= a_pos && j < b_pos;
                pend = depth >= L && in_slice;
                if (j == 0u) {
                    strand_end = true;
                    pub_exact = true;
                } else if (dmis && rt.w != 0u && !(pend && pdepth >= L)) {
                    // the letter to the left differs from the text's: what EXT + REC would do -- the pending row is
                    // left-maximal (slamem.c:141), then the interval widens to its parent and the letter is retried
                    if (pend) { SLAMEM_EMIT(top, j, (uint32_t)depth); k++; pend = false; }
                    top = rt.y; bot = rt.z; depth = pdepth; pub = pdepth - 1;
                    dcool = false;
                }
                // otherwise the index walk continues from the single row (pending position, exact parent depth)
                if (kMam) { prev_top = top; prev_bot = bot; }  // the run's positions were one row each: noted; or its parent (:126-127)
            } else {  // ST_EXT
                uint32_t bt = top >> kFmRowsLog2, bb = (bot + 1u) >> kFmRowsLog2;
                uint32_t nt, nb1;
                if (c >= 2u) {
                    // the rows of each block that hold letter c (two 64-bit masks), once; then two prefix counts.  The
                    // bottom query takes its masks and its rank sample from top's block when it is the same block
                    // (picked with selects on values: a reference to "kb or kt" would force both blocks into scratch)
                    const uint32_t c2 = c - 2u;
                    const uint64_t f0 = (c2 & 1u) ? ~0ull : 0ull, f1 = (c2 & 2u) ? ~0ull : 0ull;
                    const uint64_t mt0 = ~(u64_of(kt.b.x, kt.b.y) ^ f0) & ~(u64_of(kt.c.x, kt.c.y) ^ f1) & ~u64_of(kt.d.x, kt.d.y);
                    const uint64_t mt1 = ~(u64_of(kt.b.z, kt.b.w) ^ f0) & ~(u64_of(kt.c.z, kt.c.w) ^ f1) & ~u64_of(kt.d.z, kt.d.w);
                    const uint32_t ct = c2 == 0 ? kt.a.x : c2 == 1 ? kt.a.y : c2 == 2 ? kt.a.z : kt.a.w;
                    uint64_t mb0 = mt0, mb1 = mt1;
                    uint32_t cb = ct;
                    if (bb != bt) {
                        mb0 = ~(u64_of(kb.b.x, kb.b.y) ^ f0) & ~(u64_of(kb.c.x, kb.c.y) ^ f1) & ~u64_of(kb.d.x, kb.d.y);
                        mb1 = ~(u64_of(kb.b.z, kb.b.w) ^ f0) & ~(u64_of(kb.c.z, kb.c.w) ^ f1) & ~u64_of(kb.d.z, kb.d.w);
                        cb = c2 == 0 ? kb.a.x : c2 == 1 ? kb.a.y : c2 == 2 ? kb.a.z : kb.a.w;
                    }
                    nt = ct + count_lt(mt0, mt1, top & (kFmRows - 1u));
                    nb1 = cb + count_lt(mb0, mb1, (bot + 1u) & (kFmRows - 1u));
                } else if (ix.num_n == 0) {
                    nt = nb1 = 1u;
                } else {
                    nt = 1u + n_rows_lt(ix, top);
                    nb1 = 1u + n_rows_lt(ix, bot + 1u);
                }
                // rows of the previous position wait for this letter (left-maximality), slamem.c:139-193
                if (pend) {
                    uint32_t size = bot - top + 1u, same_left = nb1 - nt;
                    bool lvl0 = same_left < size, anc = false;
                    if (pub >= L) {  // read the exact parent depth from the records fetched with this trip
                        uint32_t t2 = top, b2 = bot;
                        pub = parent_from(rt, rb, t2, b2);
                        anc = pub >= L;
                    }
                    if (!anc && (!lvl0 || size == 1u)) {  // the common case: at most this one row, no ancestors
                        if (lvl0) { SLAMEM_EMIT(top, j, (uint32_t)depth); k++; }
                        pend = false;
                    } else {  // several rows and/or ancestors: the wave does it together, below; this trip only emits
                        e_on = true; e_level0 = lvl0; e_up = anc; e_pos = j; e_left = c;
                    }
                }
                // a slice that does not start the strand ends here: position a_pos-1 was only needed as the left
                // letter of position a_pos
                if (a_pos != 0u && j == a_pos) finished = true;
                if (!e_on && !finished) {
                    if (nt < nb1) {  // the extension occurs (slamem.c:121)
                        top = nt;
                        bot = nb1 - 1u;
                        pub++;  // parent depth of cW <= parent depth of W + 1
                        depth++;
                        consumed = true;
                    } else if (want_rec) {  // (kMam: a pending position is one row and was noted: the fall-back interval is [top,bot])
                        int d = parent_from(rt, rb, top, bot);
                        dcool = false;
                        if (d < 0) { depth = 0; pub = -1; consumed = true; }  // root, letter absent (slamem.c:125)
                        else { depth = d; pub = d - 1; if (kMam) { prev_top = top; prev_bot = bot; } }  // widened; retry the letter next trip
                    } else {
                        st = ST_REC;  // fetch the records in the next trip
                    }
                }
            }
            if (consumed) {
                if (kStats) n_pos++;
                j--;  // position j is done: it matched `depth` characters
                bool in_slice = !kSliced || (j >= a_pos && j < b_pos);
                pend = depth >= L && depth > 0 && in_slice && (!kMam || top == bot);  // slamem.c:130 (:131)
                if (kMam && !(depth >= L && top != bot)) { prev_top = top; prev_bot = bot; }  // slamem.c:197-198, skipped by :131
                // scan start of this attempt; a match that reaches it may be truncated: redo with a longer warm-up
                // (without a slice this never fires, but compiling it out moves a register copy next to the record load
                //  above -- an s_waitcnt between the loads of one trip, +1 ms: the instruction stream is checked, not assumed)
                uint32_t e = (attempt >= kMaxAttempt || qlen - b_pos < (kWarmUp << (2u * attempt))) ? qlen
                                                                                                 : b_pos + (kWarmUp << (2u * attempt));
                if (in_slice && e < qlen && (uint32_t)depth == e - j) {
                    attempt++;
                    j = (attempt >= kMaxAttempt || qlen - b_pos < (kWarmUp << (2u * attempt))) ? qlen
                                                                                          : b_pos + (kWarmUp << (2u * attempt));
                    top = 0; bot = ix.n; depth = 0; pub = -1; pend = false; k = 0; dcool = false;
                    qc.forget();
                } else if (j == 0u) {
                    strand_end = true;
                } else if (st == ST_EXT && top == bot && depth >= dmin && dmin >= 0 && !dcool && j > a_pos &&
                           !(e < qlen && (uint32_t)depth == e - j)) {
                    st = ST_DSA;  // one row, too deep to be a chance match: compare with the text itself from here on
                }
            }
            if (strand_end) {  // strand finished; rows still pending have nothing to their left (slamem.c:138)
                finished = true;
                if (pend && pub >= L && !pub_exact) { st = ST_FLUSH; finished = false; }  // parent depth needed: next trip
                else if (pend && pub >= L) { e_on = true; e_level0 = true; e_up = true; e_pos = 0u; e_left = 0xFFu; }
                else if (pend && top == bot) { SLAMEM_EMIT(top, 0u, (uint32_t)depth); k++; pend = false; }
                else if (pend) { e_on = true; e_level0 = true; e_up = false; e_pos = 0u; e_left = 0xFFu; }
            }
        }

        if (kDefer) {
            // the job goes to the queue and the lane walks on: `pub` is already the exact depth of the parent whenever ancestors are
            // to be reported (it was read with this trip's records), and the strand's MEM numbers are put right by K9
            bool put = e_on && jobbase != 0xFFFFFFFFu;
            if (put && jobbase == 0u) {  // the strand's first job: counters for every job it can still have (one per position left, and the end)
                // (positions j .. a_pos can each have one, the strand's end another; two words in front, one behind for the total)
                const uint32_t need = (j - a_pos) + 6u;
                const uint32_t base = atomicAdd(A.defer.pool_next, need) + 1u;  // (place 0 is never given out: 0 = no place yet)
                if (base <= A.defer.pool_cap && need <= A.defer.pool_cap - base) {
                    jobbase = base;
                    A.defer.inline_valid[g] = (uint8_t)(1u + (k < kInlineMems ? k : kInlineMems));
                } else { jobbase = 0xFFFFFFFFu; put = false; }
            }
            const unsigned long long pm = __ballot(put);
            if (pm != 0ull) {
                // places in the queue from the wave's chunk of 64 (every lane stores the same wave-uniform values: see lds_ovf_base)
                const uint32_t cnt = (uint32_t)__popcll(pm);
                uint32_t q_base = lds_q_base[wv], q_left = lds_q_left[wv];
                uint32_t q_new = 0;  // a new chunk when the old one does not hold them all: its first places take the rest
                if (q_left < cnt) {
                    unsigned int nb = 0;
                    if (lane == 0u) nb = atomicAdd(A.defer.njobs, 64u);
                    q_new = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
                }
                if (put) {
                    const uint32_t rank = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
                    const uint32_t q = rank < q_left ? q_base + rank : q_new + (rank - q_left);
                    if (q < A.defer.queue_cap) {
                        uint4* w = reinterpret_cast<uint4*>(A.defer.queue + q);
                        w[0] = make_uint4(g, k, top, bot);
                        w[1] = make_uint4((uint32_t)depth | (e_pos << 16), e_left | (e_level0 ? 0x100u : 0u) | (e_up ? 0x200u : 0u), jobbase + 2u + seg,
                                          A.defer.epoch);
                        seg++;
                        pend = false;  // the next trip extends from the same interval with the same letter
                        e_on = false;
                    }  // (else: the queue is full -- never, while every strand's jobs have their counters -- the job runs in the wave)
                }
                if (q_left < cnt) { lds_q_base[wv] = q_new + (cnt - q_left); lds_q_left[wv] = 64u - (cnt - q_left); }
                else { lds_q_base[wv] = q_base + cnt; lds_q_left[wv] = q_left - cnt; }
            }
            // the MEMs of this trip that wait for their places
            const unsigned long long sm = __ballot(em_on);
            if (sm != 0ull) {
                const uint32_t cnt = (uint32_t)__popcll(sm);
                unsigned long long ovf_base = lds_ovf_base[wv];
                uint32_t ovf_left = lds_ovf_left[wv];
                if (ovf_left < cnt) {
                    unsigned long long nb = 0ull;
                    if (lane == 0u) nb = atomicAdd(A.total, (unsigned long long)kOvfChunk);
                    ovf_base = u64_of((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nb), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(nb >> 32)));
                    ovf_left = kOvfChunk;
                }
                if (em_on) {
                    const unsigned long long slot = ovf_base + (unsigned long long)__popcll(sm & ((1ull << lane) - 1ull));
                    if (em_k >> 28) atomicOr(reinterpret_cast<unsigned int*>(A.total) + 9, 1u);
                    if (slot < A.capacity) {
                        A.raw_key[slot] = RawKey{g, em_k};
                        A.raw_mem[slot] = slamem_mem{em_row, em_pos, em_len};
                        // (the record was made before this trip's job, if any, was queued: seg may have moved on by one)
                        A.defer.raw_seg[slot] = jobbase + 2u + seg - ((put && !e_on) ? 1u : 0u);
                    }
                    em_on = false;
                }
                lds_ovf_base[wv] = ovf_base + cnt;
                lds_ovf_left[wv] = ovf_left - cnt;
            }
        }
        // ---- enumeration jobs, one strand at a time, all 64 lanes on its rows (every lane reaches this point) ----
        const unsigned long long t_en0 = (kStats && __ballot(e_on) != 0ull) ? wall_clock64() : 0ull;
        for (unsigned long long em = __ballot(e_on); em != 0ull; em &= em - 1ull) {
            int owner = __ffsll((long long)em) - 1;
            // (kChunk: readlane instead of a shuffle -- the job's parameters are wave-uniform and the compiler knows it, so the loops
            //  of wave_enumerate and the chunk state stay in scalar registers)
#define SLAMEM_OWN(x) (kChunk ? (uint32_t)__builtin_amdgcn_readlane((int)(x), owner) : (uint32_t)__shfl((int)(x), owner))
            uint32_t o_g = SLAMEM_OWN(g), o_k = SLAMEM_OWN(k), o_t = SLAMEM_OWN(top), o_b = SLAMEM_OWN(bot);
            uint32_t o_tag = SLAMEM_OWN(attempt) << 28;
            int o_depth = (int)SLAMEM_OWN(depth);
            uint32_t o_pos = SLAMEM_OWN(e_pos), o_left = SLAMEM_OWN(e_left);
            bool o_l0 = SLAMEM_OWN(e_level0 ? 1 : 0) != 0u, o_up = SLAMEM_OWN(e_up ? 1 : 0) != 0u;
            int fp;
            const bool o_old = kCarry && SLAMEM_OWN(old ? 1 : 0) != 0u;
#undef SLAMEM_OWN
            uint32_t e_steps = 0, e_levels = 0;
            unsigned long long ovf_base = kChunk ? lds_ovf_base[wv] : 0ull;
            uint32_t ovf_left = kChunk ? lds_ovf_left[wv] : 0u;
            uint32_t nk = (kChunk && !kStats)
                              ? wave_enumerate_merged<kCarry>(A, o_old, lane, o_g, o_k, o_tag, o_t, o_b, o_depth, o_l0, o_up, o_pos, o_left, L, &fp,
                                                              ovf_base, ovf_left)
                              : wave_enumerate<kCarry, kChunk>(A, o_old, lane, o_g, o_k, o_tag, o_t, o_b, o_depth, o_l0, o_up, o_pos, o_left, L, &fp,
                                                               ovf_base, ovf_left, kStats ? &e_steps : nullptr, kStats ? &e_levels : nullptr);
            if (kChunk) { lds_ovf_base[wv] = ovf_base; lds_ovf_left[wv] = ovf_left; }
            if (kStats && lane == 0u) { n_erow += e_steps; n_elev += e_levels; }
            if ((int)lane == owner) {
                k = nk;
                if (o_up) pub = fp;
                pend = false;  // the next trip extends from the same interval with the same letter
            }
        }

        if (kStats && t_en0 != 0ull && lane == 0u) t_enum += wall_clock64() - t_en0;

#ifdef SLAMEM_DIAG_TRIPS
        if (kDefer) {
            dg_wtrips++;
            if (active) dg_trips++;
            if (active && finished) {
                atomicMax(A.defer.njobs + 4, dg_trips); atomicAdd(A.defer.njobs + 5, dg_trips); atomicAdd(A.defer.njobs + 6, 1u);
                if (seg) { atomicAdd(A.defer.njobs + 7, dg_trips); atomicAdd(A.defer.njobs + 8, 1u); }
                dg_trips = 0;
            }
        }
#endif
        if (kDefer && active && finished && seg != 0u) {  // K9 adds the jobs' MEMs: {jobs, MEMs the lane reported} head the strand's counters
            A.defer.pool[jobbase] = seg;
            A.defer.pool[jobbase + 1u] = k;
            A.defer.list[atomicAdd(A.defer.nlist, 1u)] = make_uint2(g, jobbase);
        }
        if (active && finished) {
            (kCarry && old ? A.prev.block_counts : A.block_counts)[g] = k;
            (kCarry && old ? A.prev.item_attempt : A.item_attempt)[g] = (uint8_t)attempt;
            active = false;
        }
        if (kStats) n_enum += e_on;
    }
#ifdef SLAMEM_DIAG_TRIPS
    if (kDefer && lane == 0u) { atomicMax(A.defer.njobs + 9, dg_wtrips); atomicAdd(A.defer.njobs + 10, dg_wtrips); }
#endif
    if (kStats) {
        stat_flush<kStats>(A.stats + SC_FM_TOP, n_kt); stat_flush<kStats>(A.stats + SC_FM_BOT, n_kb);
        stat_flush<kStats>(A.stats + SC_REC_FAIL_LINES, n_rec_fail); stat_flush<kStats>(A.stats + SC_REC_PEND_LINES, n_rec_pend);
        stat_flush<kStats>(A.stats + SC_REC_FLUSH_LINES, n_rec_flush); stat_flush<kStats>(A.stats + SC_QUERY_LOADS, n_qloads);
        stat_flush<kStats>(A.stats + SC_LANE_TRIPS, n_trips); stat_flush<kStats>(A.stats + SC_WAVE_TRIPS, n_wtrips);
        stat_flush<kStats>(A.stats + SC_POSITIONS, n_pos); stat_flush<kStats>(A.stats + SC_ENUM_JOBS, n_enum);
        stat_flush<kStats>(A.stats + SC_DIR_SA, n_dsa); stat_flush<kStats>(A.stats + SC_DIR_GROUPS, n_dgrp);
        stat_flush<kStats>(A.stats + SC_DIR_RECS, n_drec); stat_flush<kStats>(A.stats + SC_DIR_LETTERS, n_dlet);
        stat_flush<kStats>(A.stats + SC_JUMP_LINES, n_jump);
        stat_flush<kStats>(A.stats + SC_ENUM_ROW_STEPS, n_erow); stat_flush<kStats>(A.stats + SC_ENUM_LEVELS, n_elev);
#pragma unroll
        for (uint32_t q = 0; q < 11u; q++) {
            stat_flush<kStats>(A.stats + SC_STATE_TRIPS + q, n_state[q]);
            stat_flush<kStats>(A.stats + SC_STATE_WAVES + q, n_wstate[q]);
        }
        stat_flush<kStats>(A.stats + SC_SKIP_GROUPS, n_skv); stat_flush<kStats>(A.stats + SC_SKIP_QLOADS, n_skq);
        stat_flush<kStats>(A.stats + SC_SKIP_PROBES, n_skp); stat_flush<kStats>(A.stats + SC_SKIP_OK, n_skok);
        if ((threadIdx.x & 63u) == 0u) {
            const unsigned long long t1 = wall_clock64();
            atomicMax(A.stats + SC_T_LAST, t1);
            atomicAdd(A.stats + SC_T_WAVE_SUM, t1 - t_wave0);
            atomicAdd(A.stats + SC_T_ENUM_SUM, t_enum);
        }
    }
}

// ---- K8a on a packed span ------------------------------------------------------------------------------------
// 16 letters of the query buffer -> 2-bit codes (A,C,G,T = 0..3, the first letter lowest) and a bit per letter that is
// none of them (same classes as ascii_code_q)
__device__ __forceinline__ void pack_chunk(const uint4& a, uint32_t& pk, uint32_t& nm) {
    pk = 0u; nm = 0u;
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t u = (w[i >> 2] >> (8 * (i & 3))) & 0xDFu;
        const uint32_t x = (u >> 1) & 3u, code = x ^ (x >> 1);           // A 0, C 1, G 2, T 3
        const bool ok = ((0x54474341u >> (8u * code)) & 0xFFu) == u;    // the letter that has this code
        pk |= (ok ? code : 0u) << (2 * i);
        nm |= (ok ? 0u : 1u) << i;
    }
}

// letters [q, q+len) of the packed span (len <= 24) as a number, the FIRST letter in the lowest bits
__device__ __forceinline__ uint64_t span_letters(const uint32_t* pk, const uint16_t* nm, uint32_t q, uint32_t len, bool& has_n) {
    const uint32_t c = q >> 4, o = q & 15u;
    uint64_t v = ((uint64_t)pk[c] | ((uint64_t)pk[c + 1u] << 32)) >> (2u * o);
    if (o) v |= (uint64_t)pk[c + 2u] << (64u - 2u * o);
    const uint64_t m = (uint64_t)nm[c] | ((uint64_t)nm[c + 1u] << 16) | ((uint64_t)nm[c + 2u] << 32);
    has_n = ((m >> o) & ((1ull << len) - 1ull)) != 0ull;
    return v & ((1ull << (2u * len)) - 1ull);
}

// The prefilter of one item whose strand lies in the packed span (rel = span letter of the record's first letter): the
// same windows, the same tests and the same treatment of N as prefilter_item below, each window cut out of the packed
// letters instead of rolled letter by letter.
template <bool kStats>
__device__ __forceinline__ uint8_t prefilter_item_packed(const SearchArgs& A, const ItemDesc& d, const uint32_t* pk,
                                                         const uint16_t* nm, uint32_t rel, uint32_t& n_probe) {
    const IndexView& ix = A.ix;
    const uint32_t k = ix.kfilter_k, L = A.min_len, s = L - k + 1u;
    const uint32_t rev = d.slice_rev >> 31, slen = d.len;
    const uint32_t sl = d.slice_rev & 0x7FFFFFFFu;
    const uint32_t a = sl * kSliceLen;
    if (!(slen >= k && slen - a >= 1u)) return 0;
    const uint32_t b = slen - a < kSliceLen ? slen : a + kSliceLen;
    // letters [p, p+len) of the strand as the build hashes them: the first letter in the highest bits
    auto win = [&](uint32_t p, uint32_t len, bool& has_n) -> uint64_t {
        const uint64_t v = span_letters(pk, nm, rel + (rev ? slen - p - len : p), len, has_n);
        if (rev) return ~v & ((1ull << (2u * len)) - 1ull);  // complemented; its first letter is the forward strand's last
        uint64_t r = __brevll(v) >> (64u - 2u * len);          // pairs in reverse order, the two bits of each swapped
        return ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    };
    auto present = [&](const uint64_t* line, uint64_t h) -> bool {
        const uint64_t bits = kfilter_bits(h);
        return (line[kfilter_word(h)] & bits) == bits;
    };
    uint8_t res = 0;
    bool hn;
    if (s <= 6u) {  // the cascade (see prefilter_item)
        const uint32_t k1 = k - 2u, s1 = L - k1 + 1u;
        const bool three = L >= k + 2u && k + 2u <= 32u && ix.kfilter_levels >= 3u;
        const uint32_t p0 = (a + s1 - 1u) / s1 * s1;
        const uint32_t pmax = (uint32_t)(((uint64_t)b + s1 - 2u < (uint64_t)(slen - k1)) ? b + s1 - 2u : slen - k1);
#pragma unroll 1
        for (uint32_t p = p0; p <= pmax && !res; p += s1) {
            const uint64_t v1 = win(p, k1, hn);
            if (hn) { res = 1; break; }                          // holds an N: cannot be ruled out
            const uint64_t h1 = kfilter_hash(v1 ^ kFilterShortSalt);
            const uint64_t* line = ix.kfilter + kfilter_line(h1, ix.kfilter_log2);
            if (kStats) n_probe++;
            if (!present(line, h1)) continue;
#pragma unroll 1
            for (uint32_t t = 0; t < 3u && !res; t++) {           // the k-mers that start at p-2, p-1, p
                if (p + t < 2u || (uint64_t)p + t - 2u + k > slen) continue;
                const uint32_t q = p + t - 2u;
                const uint64_t vk = win(q, k, hn);
                if (hn) { res = 1; break; }
                if (!present(line, kfilter_hash(vk))) continue;
                if (!three) { res = 1; break; }
#pragma unroll 1
                for (uint32_t u = 0; u < 3u && !res; u++) {       // the (k+2)-mers that start at q-2, q-1, q
                    if (q + u < 2u || (uint64_t)q + u + k > slen) continue;
                    const uint64_t v3 = win(q + u - 2u, k + 2u, hn);
                    if (hn) { res = 1; break; }
                    if (present(line, kfilter_hash(v3 ^ kFilterLongSalt))) res = 1;
                }
            }
        }
    } else {
        const uint32_t p0 = (a + s - 1u) / s * s;
        const uint32_t pmax = (uint32_t)(((uint64_t)b + s - 2u < (uint64_t)(slen - k)) ? b + s - 2u : slen - k);
#pragma unroll 1
        for (uint32_t p = p0; p <= pmax && !res; p += s) {
            const uint64_t vk = win(p, k, hn);
            if (hn) { res = 1; break; }
            const uint64_t* line = ix.kfilter + kfilter_line(kfilter_hash((vk >> 4) ^ kFilterShortSalt), ix.kfilter_log2);
            if (kStats) n_probe++;
            if (!present(line, kfilter_hash(vk))) continue;
            if (k + 2u > 32u || ix.kfilter_levels < 3u) { res = 1; break; }
            // (min_len >= k+6 here) a MEM around this window that starts at m also holds the (k+2)-mer that starts at
            // max(m, p-2), one of p-2, p-1, p -- entered in the same line: a second test that costs no line of HBM
#pragma unroll 1
            for (uint32_t u = 0; u < 3u && !res; u++) {
                if (p + u < 2u || (uint64_t)p + u + k > slen) continue;
                const uint64_t v3 = win(p + u - 2u, k + 2u, hn);
                if (hn || present(line, kfilter_hash(v3 ^ kFilterLongSalt))) res = 1;
            }
        }
    }
    return res;
}

// K8a: presence prefilter.  A MEM of length >= L that starts in the item's slice [a,b) contains a k-mer window
// starting at a multiple of s = L-k+1 inside [a, b+s-2]; if none of those windows occurs in the text (filter says
// "absent": no false negatives) the item cannot emit anything and K8 skips it.  Windows holding an N count as
// present.  One lane per item, early exit at the first present window (the matching strand of a read exits after
// a few probes; the other strand pays ~ len/s probes instead of a full scan).
template <bool kStats>
__device__ __forceinline__ uint8_t prefilter_item(const SearchArgs& A, const ItemDesc& d, uint32_t& n_probe, uint32_t& n_qload) {
    const IndexView& ix = A.ix;
    const uint32_t k = ix.kfilter_k, L = A.min_len;
    const uint32_t s = L - k + 1u;
    uint32_t sl = d.slice_rev & 0x7FFFFFFFu;
    uint32_t a = sl * kSliceLen;
    uint32_t b = d.len - a < kSliceLen ? d.len : a + kSliceLen;
    uint8_t res = 0;
    if (d.len >= k && d.len - a >= 1u && s <= 6u) {
        // Two levels (short minimum lengths, where one level would probe almost every window): a MEM >= L that starts
        // in [a,b) contains a (k-2)-mer window starting at a multiple of s1 = L-(k-2)+1 inside [a, b+s1-2], AND -- if m
        // is where the MEM starts and p that window -- the k-mer starting at max(m, p-2), one of p-2, p-1, p.  So the
        // k-mers are probed only behind a (k-2)-mer that is present: s1/s times fewer probes for an empty strand, and
        // far fewer chance survivors (a random strand needs a (k-2)-mer hit and a k-mer hit next to it).
        const uint32_t k1 = k - 2u, s1 = L - k1 + 1u;
        // (all positions fit 32 bits: a record is shorter than 2^32 letters; the per-letter loop is instruction bound --
        //  a wave64 VALU instruction occupies its SIMD for four cycles -- so no 64-bit arithmetic and no division in it)
        const uint32_t p0 = (a + s1 - 1u) / s1 * s1;
        uint32_t pmax = (uint32_t)(((uint64_t)b + s1 - 2u < (uint64_t)(d.len - k1)) ? b + s1 - 2u : d.len - k1);
        if (p0 <= pmax) {
            QueryStream qs;
            // third level: if the MEM is at least k+2 long it also contains the (k+2)-mer that starts at max(m, s'-2), where
            // s' is the start of the k-mer above -- one of s'-2, s'-1, s'.  For L == k+2 that makes the test exact up to the
            // filter's false positives: a strand survives only if it really shares L letters with the text.
            const bool three = L >= k + 2u && k + 2u <= 32u && ix.kfilter_levels >= 3u;
            const uint32_t kw = three ? k + 2u : k;  // letters kept in the rolling value
            const uint64_t mask = kw >= 32u ? ~0ull : (1ull << (2u * kw)) - 1ull;
            const uint64_t maskk = (1ull << (2u * k)) - 1ull, mask1 = (1ull << (2u * k1)) - 1ull;
            uint64_t km = 0;
            uint32_t run = 0, confirm = 0, confirm2 = 0;
            // line of the (k-2)-mer whose hit started the confirmations under way (every k-mer / (k+2)-mer tested behind
            // it contains that (k-2)-mer, so the build entered it there), and of the one behind the (k+2)-mer tests
            const uint64_t* line1 = ix.kfilter;
            const uint64_t* line2 = ix.kfilter;
            uint32_t x = p0 >= 4u ? p0 - 4u : 0u;
            qs.init(A.qwords, d.base, d.len, d.slice_rev >> 31, x);
            uint64_t xe64 = (uint64_t)pmax + k1 + 1u + (three ? 2u : 0u);
            const uint32_t xend = xe64 > (uint64_t)d.len - 1u ? d.len - 1u : (uint32_t)xe64;
            uint32_t wend = p0 + k1 - 1u;  // letter at which the next probed (k-2)-mer window ends
            const uint32_t wlast = pmax + k1 - 1u;
            for (; x <= xend && !res; x++) {
                uint32_t c = qs.next();
                if (c >= 2u) { km = ((km << 2) | (uint64_t)(c - 2u)) & mask; run++; }
                else { km = 0; run = 0; }
                if (x == wend && wend <= wlast) {               // short window [x+1-k1, x+1)
                    wend += s1;
                    if (run < k1) res = 1;                      // holds an N: cannot be ruled out
                    else {
                        uint64_t h = kfilter_hash((km & mask1) ^ kFilterShortSalt), bits = kfilter_bits(h);
                        if (kStats) n_probe++;
                        line1 = ix.kfilter + kfilter_line(h, ix.kfilter_log2);  // the confirmations are in this line too
                        if ((line1[kfilter_word(h)] & bits) == bits) confirm = 3;  // k-mers ending at x, x+1, x+2
                    }
                }
                if (confirm && !res) {
                    confirm--;
                    if (x + 1u >= k) {                          // the k-mer [x+1-k, x+1) lies inside the strand
                        if (run < k) res = 1;
                        else {
                            uint64_t h = kfilter_hash(km & maskk), bits = kfilter_bits(h);
                            if ((line1[kfilter_word(h)] & bits) == bits) {
                                if (three) { confirm2 = 3; line2 = line1; }  // (k+2)-mers ending at x, x+1, x+2
                                else res = 1;
                            }
                        }
                    }
                }
                if (confirm2 && !res) {
                    confirm2--;
                    if (x + 1u >= k + 2u) {
                        if (run < k + 2u) res = 1;
                        else {
                            uint64_t h = kfilter_hash(km ^ kFilterLongSalt), bits = kfilter_bits(h);
                            if ((line2[kfilter_word(h)] & bits) == bits) res = 1;
                        }
                    }
                }
            }
            if (kStats) n_qload += qs.loads;
        }
    } else if (d.len >= k && d.len - a >= 1u) {
        const uint32_t p0 = (a + s - 1u) / s * s;               // first sampled window start >= a
        // last window start that can serve this slice
        const uint32_t pmax = (uint32_t)(((uint64_t)b + s - 2u < (uint64_t)(d.len - k)) ? b + s - 2u : d.len - k);
        if (p0 <= pmax) {
            QueryStream qs;
            qs.init(A.qwords, d.base, d.len, d.slice_rev >> 31, p0);
            const uint64_t mask = (1ull << (2u * k)) - 1ull;
            uint64_t km = 0;
            uint32_t run = 0, wend = p0 + k - 1u;               // letter at which the next probed window ends
            const uint32_t xend = pmax + k - 1u;
            for (uint32_t x = p0; x <= xend && !res; x++) {
                uint32_t c = qs.next();
                if (c >= 2u) { km = ((km << 2) | (uint64_t)(c - 2u)) & mask; run++; }
                else { km = 0; run = 0; }
                if (x == wend) {                                // window [x+1-k, x+1)
                    wend += s;
                    if (run < k) res = 1;                       // holds an N: cannot be ruled out
                    else {
                        // the k-mer is in the line of each (k-2)-mer it contains: look in that of its first k-2 letters
                        const uint64_t* line = ix.kfilter + kfilter_line(kfilter_hash((km >> 4) ^ kFilterShortSalt), ix.kfilter_log2);
                        uint64_t h = kfilter_hash(km), bits = kfilter_bits(h);
                        if (kStats) n_probe++;
                        if ((line[kfilter_word(h)] & bits) == bits) res = 1;
                    }
                }
            }
            if (kStats) n_qload += qs.loads;
        }
    }
    return res;
}

// (without the second bound the compiler takes 127 registers -- four waves per SIMD -- and K8a waits for memory)
#ifndef SLAMEM_PF_WAVES
#define SLAMEM_PF_WAVES 8
#endif
template <bool kStats>
__global__ void __launch_bounds__(256, SLAMEM_PF_WAVES) k_prefilter(SearchArgs A, uint8_t* __restrict__ alive) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < A.num_items;
    if (!live) g = 0;  // idle lanes stay for the block's barriers and the diagnostic reduction (they store nothing)
    uint32_t n_probe = 0, n_qload = 0;
    ItemDesc d = A.items[g];
    if (!live) d.len = 0;
    uint8_t res;
    if (kPfStageChunks != 0u) {
        // the 16-byte chunks [lo, hi) of the query buffer that the block's strands lie in
        __shared__ uint32_t s_pk[kPfStageChunks + 2u];
        __shared__ uint16_t s_nm[kPfStageChunks + 2u];
        __shared__ unsigned long long s_lo[4], s_hi[4];
        unsigned long long lo = d.len ? d.base >> 4 : ~0ull, hi = d.len ? ((d.base + d.len - 1u) >> 4) + 1ull : 0ull;
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            unsigned long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if ((threadIdx.x & 63u) == 0u) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; w++) {
            lo = s_lo[w] < lo ? s_lo[w] : lo;
            hi = s_hi[w] > hi ? s_hi[w] : hi;
        }
        // packs chunks [lo, lo+n) of the query buffer (every lane of the block helps)
        auto pack_span = [&](unsigned long long first_chunk, uint32_t n) {
            const uint4* src = reinterpret_cast<const uint4*>(A.qwords) + first_chunk;
#pragma unroll 1
            for (uint32_t i = threadIdx.x; i < n; i += 256u) {
                uint32_t pk, nm;
                pack_chunk(src[i], pk, nm);
                s_pk[i] = pk;
                s_nm[i] = (uint16_t)nm;
                if (kStats) n_qload++;
            }
            if (threadIdx.x < 2u) { s_pk[n + threadIdx.x] = 0u; s_nm[n + threadIdx.x] = 0u; }  // a window's reads run two words past it
            __syncthreads();
        };
        if (hi > lo && hi - lo <= (unsigned long long)kPfStageChunks) {  // (the same for every lane of the block)
            // the usual case: the strands of all 256 items (128 reads of up to 320 letters) in one span
            pack_span(lo, (uint32_t)(hi - lo));
            res = prefilter_item_packed<kStats>(A, d, s_pk, s_nm, (uint32_t)(d.base - (lo << 4)), n_probe);
        } else {
            // longer reads: the block works in two or four parts (pairs of waves, single waves -- 64 reads of up to 640
            // letters, 32 of up to 1280), if the span of every part fits; else (long records, slices) the letter loop
            auto span_lo = [&](uint32_t first, uint32_t count) {
                unsigned long long v = ~0ull;
                for (uint32_t w = first; w < first + count; w++) v = s_lo[w] < v ? s_lo[w] : v;
                return v;
            };
            auto span_hi = [&](uint32_t first, uint32_t count) {
                unsigned long long v = 0ull;
                for (uint32_t w = first; w < first + count; w++) v = s_hi[w] > v ? s_hi[w] : v;
                return v;
            };
            uint32_t per_part = 0u;  // waves per part
            for (uint32_t cnt = 2u; cnt >= 1u && !per_part; cnt >>= 1) {
                bool fits = true;
                for (uint32_t first = 0; first < 4u; first += cnt) {
                    const unsigned long long l = span_lo(first, cnt), h = span_hi(first, cnt);
                    if (h > l && h - l > (unsigned long long)kPfStageChunks) fits = false;
                }
                if (fits) per_part = cnt;
            }
            if (per_part && hi > lo) {
                const uint32_t my_wave = threadIdx.x >> 6;
                res = 0;
#pragma unroll 1
                for (uint32_t first = 0; first < 4u; first += per_part) {
                    lo = span_lo(first, per_part);
                    hi = span_hi(first, per_part);
                    if (hi <= lo) continue;  // nothing but empty records (or no items) in these waves
                    __syncthreads();         // the part before has been read
                    pack_span(lo, (uint32_t)(hi - lo));
                    if (my_wave >= first && my_wave < first + per_part)
                        res = prefilter_item_packed<kStats>(A, d, s_pk, s_nm, (uint32_t)(d.base - (lo << 4)), n_probe);
                }
            } else {
                res = prefilter_item<kStats>(A, d, n_probe, n_qload);
            }
        }
    } else {
        res = prefilter_item<kStats>(A, d, n_probe, n_qload);
    }
    if (live) alive[g] = res;
    if (kStats) {
        stat_flush<kStats>(A.stats + SC_PF_PROBES, n_probe);
        stat_flush<kStats>(A.stats + SC_PF_QUERY_LOADS, n_qload);
        stat_flush<kStats>(A.stats + SC_PF_ITEMS, live ? 1u : 0u);
    }
}

// K8a on a LIST of items (the strands K8s left to the index walk): the presence tests of prefilter_item for each, survivors
// appended to a second list.  A read is left with BOTH strands (a bucket that holds more k-mers than it has slots says nothing
// about either), and the strand that holds nothing -- the wrong strand of the read -- is the expensive one for K8: 150 positions
// of failing extensions, a millisecond of dependent steps.  The filter proves nearly all of those empty in a few lines each;
// the list is then made again from alive[].
__global__ void __launch_bounds__(256) k_prefilter_list(SearchArgs A, const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
                                                        uint8_t* __restrict__ alive) {
    const uint32_t count = *list_count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint32_t g = list[i];
        const ItemDesc d = item_of(A, g);
        uint32_t n_probe = 0, n_qload = 0;
        if (!prefilter_item<false>(A, d, n_probe, n_qload)) alive[g] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// K8s: seed-and-compare -- the search for READS (k_seed_mems)
// ------------------------------------------------------------------------------------------
// Replaces, for strands of up to kSeedMaxLenLong letters, the scan of GetMatches (slamem.c:105-199) AND the prefilter: the
// output of the scan is the set of all maximal matches of at least L letters, and every such match contains a
// k-letter window that starts at a multiple of s = L - k + 1 of its strand.  So, per read:
//   1. windows of the forward strand at multiples of s are looked up in the seed table by their canonical form: ONE
//      64-byte line (and the spill list behind a bucket of 13 to 28 k-mers) gives every text position of the window AND of
//      its reverse complement -- the window of the reverse strand that covers the same letters (strand offset len - k - o;
//      those offsets form a residue class mod s too, so the reverse strand's MEMs all hold one).  One lane per window, in two
//      rounds when the windows lie close: every m-th window first; a window of the other kind that lies inside a run a
//      first-round compare measured, and whose k-mer occurs once in the text (a plane of the text units), is not looked up
//      (one strand asked for: first-round hits in the other orientation are compared too, for these marks only);
//   2. a hit (strand, text position p, strand offset o) is a diagonal d = p - o; the MEM around the window is the run of
//      agreeing letters on that diagonal: the strand (bit-planes in LDS) XOR the text (four 32-byte units = the 192 letters
//      from d on), one lane per compare.  A run that holds a first-round window is reported by the first of those, one that
//      holds none by its first window (a hit whose neighbour window hit the same diagonal is dropped before the compare when
//      the two windows overlap or touch); a window that is its own reverse complement is compared on both strands;
//   3. the MEMs of a strand are ranked in the reference's emission order -- start descending, then length descending
//      (slamem.c:114,139-193: the scan runs right to left, a position reports its deepest interval first), MEMs with the same
//      start and length as the rows of the suffix array lie around the interval the walk came up from (slamem.c:140,165:
//      from the text letters behind the match, or TextRec::row) -- and go to the strand's inline slots / the overflow list
//      with their text position (K9 skips the suffix-array read for them).
// Whatever this cannot decide exactly goes to the index walk (K8), read by read (alive[] = 1 for both strands; their numbers
// go straight into K8's work list): a letter that is not A,C,G,T in the read when the text holds such letters too (N equals N
// in the reference: a MEM may span one), a read longer than the planes hold (or with more than 64 windows), a bucket with more
// than 28 k-mers (a repeat family), lists that run over.  Nothing here is approximate: a read is either reported completely
// by this kernel or completely by K8.
// Three forms (plane words a strand: 3, 4, 6 = reads of up to 192, 256, 384 letters), chosen per batch by the average read
// length and by what the last batches against the index left for their length (SearchJob::prep / collect, seed_words_hint).
#ifndef SLAMEM_SEED_READS
#define SLAMEM_SEED_READS 21
#endif
// reads of a wave in the instantiation for reads of up to 192 letters: 21 x 3 plane words are 63 lanes of work in the planes
// step, 21 x 9 first-round windows three full trips (at most 32: a flag word holds a bit per read); the long-read form takes 16
constexpr uint32_t kSeedReads = SLAMEM_SEED_READS, kSeedReadsMid = 16, kSeedReadsLong = 16;
constexpr uint32_t kSeedMaxLen = 192;   // letters of a strand the planes hold in the instantiation for reads of up to 192 letters (three words)
constexpr uint32_t kSeedMaxLenMid = 256;   // ... in the one for 2 x 250 bp runs (four words: a batch takes it when its reads average 193 .. 256) ...
constexpr uint32_t kSeedMaxLenLong = 384;  // ... and in the one for longer reads (six words: a batch takes it when its reads average more than 256)
#ifndef SLAMEM_SEED_JOBS
#define SLAMEM_SEED_JOBS 16
#endif
constexpr uint32_t kSeedJobs = SLAMEM_SEED_JOBS;      // compares of one wave: this many per read
#ifndef SLAMEM_SEED_MEMS
#define SLAMEM_SEED_MEMS 12
#endif
constexpr uint32_t kSeedMems = SLAMEM_SEED_MEMS;     // MEMs of one wave: this many per read
constexpr uint32_t kSigLetters = 12, kSigMask = (1u << kSigLetters) - 1u, kSigKnown = 1u << 24;  // text letters kept behind a MEM (ties, phase 3)

template <uint32_t NW, uint32_t R>
struct SeedWave {
    static constexpr uint32_t kJobs = kSeedJobs * R * NW / 3u, kMems = kSeedMems * R * NW / 3u;  // (longer strands: more windows, more MEMs)
    uint64_t pl[R][2][2][NW + 1];       // [read][strand][plane][word]; the last word stays 0 (a window's second word)
    uint32_t len[R];                    // letters (0: the read takes no part)
    uint32_t nwin[R];                   // windows
    union {
        // first the wave's reads as they are (their bytes, 16-byte chunks of the query buffer), while the planes are made ...
        uint4 raw[R * NW * 4 + 1];
        // ... then the compares and the MEMs
        struct {
            uint32_t job_p[kJobs], job_x[kJobs];
            uint32_t mem_key[kMems], mem_ref[kMems];
            // bits 25-29: the strand (of the wave's 32); the kSigLetters text letters behind the match (plane 0: bits 0-11, plane 1:
            // bits 12-23), bit 24: they are known
            uint32_t mem_sig[kMems];
        };
    };
    uint32_t bad[R];                    // the read holds a letter that is not A,C,G,T (and the text holds such letters too)
    uint64_t pn[R][NW + 1];             // forward strand, bit per letter: not A,C,G,T (a text without such letters: it agrees with nothing)
    uint32_t hasn;                      // some read of the wave has such a letter
    unsigned long long expl[R];         // bit per window of a read: its only occurrence in the text is accounted for (round A)
    union {
        uint16_t ring[128];             // window ids (read << 8 | window) waiting for a full trip
        unsigned long long smask[2 * R];  // ... later: per strand, which MEMs of the wave's list are its own
    };
    uint32_t flags;                     // bit per read: it is left to K8 (both strands)
    uint32_t pad[3];
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// bits [sh, sh + 64) of the 128-bit value hi:lo (sh in 0..63)
__device__ __forceinline__ uint64_t funnel64(uint64_t lo, uint64_t hi, uint32_t sh) { return (lo >> sh) | ((hi << 1) << (63u - sh)); }
// the bits [lo, hi) of a word (any integers; clamped to 0..64)
__device__ __forceinline__ uint64_t bits_below(int x) {  // the bits [0, x)
    const int c = x < 0 ? 0 : x > 64 ? 64 : x;
    return c == 0 ? 0ull : (~0ull >> (64 - c));
}
__device__ __forceinline__ uint64_t bits_range(int lo, int hi) { return bits_below(hi) & ~bits_below(lo); }
__device__ __forceinline__ uint32_t sel4(const uint4& v, uint32_t a) {
    const uint32_t lo = (a & 1u) ? v.y : v.x, hi = (a & 1u) ? v.w : v.z;
    return (a & 2u) ? hi : lo;
}
__device__ __forceinline__ uint32_t sel12(const uint4& b0, const uint4& b1, const uint4& b2, uint32_t e) {
    const uint32_t a = e & 3u, x0 = sel4(b0, a), x1 = sel4(b1, a), x2 = sel4(b2, a);
    return e < 4u ? x0 : e < 8u ? x1 : x2;
}
// bit 7 of every byte of m -> four bits
__device__ __forceinline__ uint32_t byte_tops(uint32_t m) {
    const uint32_t y = m >> 7;
    return (y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xFu;
}

#ifndef SLAMEM_SEED_WAVES
#define SLAMEM_SEED_WAVES 1
#endif
template <bool kStats, uint32_t NW, uint32_t R>
__global__ void __launch_bounds__(256, SLAMEM_SEED_WAVES) k_seed_mems(SearchArgs A, uint8_t* __restrict__ alive) {
    constexpr uint32_t kMaxLen = 64u * NW;
    static_assert(R <= 32u && R * NW <= 128u, "a flag word holds a bit per read; the planes step runs in at most two passes");
    __shared__ SeedWave<NW, R> lds[4];
    const IndexView& ix = A.ix;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    SeedWave<NW, R>& S = lds[wv];
    const uint32_t strands = A.strands, k = ix.seed_k, L = A.min_len, s = L - k + 1u;
    const uint32_t kmask = (1u << k) - 1u;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4u + wv) * R;
    if (r0 >= (uint64_t)A.num_queries) return;
    const uint32_t nr = (uint64_t)A.num_queries - r0 < R ? (uint32_t)((uint64_t)A.num_queries - r0) : R;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int64_t ulast = (int64_t)text_units(ix.n) - 1;
    uint32_t n_win = 0, n_cmp = 0, n_nm = 0, n_mem = 0;

    // ---- the wave's reads: lengths, windows, bit-planes of both strands -------------------------------------------------
    const uint64_t off = lane <= nr ? A.offsets[r0 + lane] : 0ull;
    // The reads' bytes are asked for BEFORE the offsets have arrived, on the guess that every read of the batch is as long as
    // the first (sequencing runs are): where this wave's reads lie then follows from three words every wave reads (scalar
    // loads, cached).  The guess is checked against the offsets when they are there; a wave it does not hold for asks again.
    // A wave's life is a chain of dependent round trips (offsets, bytes, lookups, compares -- twice with two rounds) of about
    // 2 us each, and the chip holds 4,096 waves: one link less is 0.25 ms of the headline's 476,192 waves.
    constexpr uint32_t kRawChunks = R * NW * 4u + 1u, kSpecIters = (kRawChunks + 63u) / 64u;
#ifndef SLAMEM_SEED_NO_SPEC
    constexpr bool kSpec = kSpecIters <= 5u;
#else
    constexpr bool kSpec = false;
#endif
    uint4 sv[kSpec ? kSpecIters : 1u];
    uint64_t g0 = 0, len0 = 0;
    bool spec = false;
    if (kSpec) {
        const uint64_t o0 = A.offsets[0], o1 = A.offsets[1], o_end = A.offsets[A.num_queries];
        len0 = o1 - o0;
        g0 = o0 + r0 * len0;
        const uint64_t g1 = g0 + (uint64_t)nr * len0;
        spec = len0 != 0ull && len0 <= (uint64_t)kMaxLen && g1 <= o_end;  // (at most R * NW * 64 bytes: they fit the buffer; nothing behind the batch is touched)
        const uint32_t gn = spec ? (uint32_t)(((g1 - 1u) >> 4) - (g0 >> 4) + 1u) : 0u;
        const uint4* src = reinterpret_cast<const uint4*>(A.qwords) + (g0 >> 4);
#pragma unroll
        for (uint32_t j = 0; j < kSpecIters; j++) {
            const uint32_t c = lane + 64u * j;
            sv[j] = c < gn ? src[c] : make_uint4(0, 0, 0, 0);
        }
    }
    const uint64_t offn = __shfl_down(off, 1);
    if (kSpec) spec = spec && __ballot(lane <= nr && off != g0 + (uint64_t)lane * len0) == 0ull;
    uint32_t len = 0, nwin = 0;
    bool left = false;  // both strands of the read are left to K8
    if (lane < nr) {
        const uint64_t l64 = offn - off;
        if (l64 > kSliceLen) atomicOr(A.seed_long_flag, 1u);  // (such a record needs slices: what the caller assumed does not hold)
        if (l64 > kMaxLen) left = l64 >= L; else len = (uint32_t)l64;
        if (len >= L) nwin = (len - k) / s + 1u;
        if (nwin > 64u) { left = true; nwin = 0; }
    }
#ifndef SLAMEM_SEED_NO_FORM_HINT
    const uint32_t lg_grid = 31u - (uint32_t)__clz((int)(gridDim.x | 1u));
    if ((blockIdx.x & ((1u << (lg_grid > 8u ? lg_grid - 8u : 0u)) - 1u)) == 0u) {
        // What the next batch against this index should know (the form is chosen by the average length: a batch of mixed
        // lengths may hold many reads beyond it, and each of those costs a walk of the index).  A sample counts -- 256 to 511
        // blocks spread evenly over the batch, every block of a small one: every wave of the grid adding to one word is a queue
        // of its own (5 M reads of 250 letters: 13 ms instead of 5).
        const uint64_t l64 = lane < nr ? offn - off : 0ull;
        if (lane == 0u) atomicAdd(A.seed_wide + 3, nr);
        if (NW < 6u) {
            const bool w4 = NW < 4u && l64 > kMaxLen && l64 <= kSeedMaxLenMid && l64 >= L;
            const bool w6 = l64 > (kMaxLen > kSeedMaxLenMid ? kMaxLen : kSeedMaxLenMid) && l64 <= kSeedMaxLenLong && l64 >= L;
            const uint32_t n4 = (uint32_t)__popcll(__ballot(w4)), n6 = (uint32_t)__popcll(__ballot(w6));
            if (lane == 0u && n4) atomicAdd(A.seed_wide, n4);
            if (lane == 0u && n6) atomicAdd(A.seed_wide + 1, n6);
        }
        if (NW > 3u) {
            const bool needed = l64 > (NW == 4u ? kSeedMaxLen : kSeedMaxLenMid) && l64 <= kMaxLen;
            if (__ballot(needed) != 0ull && lane == 0u) A.seed_wide[2] = 1u;  // (every wave that writes, writes the same)
        }
    }
#endif
    uint32_t wflags = (uint32_t)__ballot(left);  // wave-uniform part of the flags (a bit per read)
    if (lane == 0u) S.flags = 0u;
    const uint32_t plen = nwin ? len : 0u;
    // The wave's reads lie next to each other in the query buffer: all their 16-byte chunks are fetched at once into LDS (a few
    // wide loads in flight together) and the letters are taken from there.  (Letter by letter from global memory, each round
    // of 64 letters waited for its own 64-byte load: the kernel spent half its time there.)  A wave whose reads span more than
    // the buffer holds (a long record among them) reads them from global memory.
    const uint64_t span0 = u64_of((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, 0), (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), 0));
    const uint64_t span1 = u64_of((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, (int)nr), (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), (int)nr));
    const uint64_t chunk0 = span0 >> 4;
    const uint32_t nchunks = span1 > span0 ? (uint32_t)(((span1 - 1u) >> 4) - chunk0 + 1u) : 0u;
    const bool staged = nchunks <= kRawChunks;
    if (kSpec && spec) {  // (the guess held: span0 = g0, the chunks asked for are the wave's)
#pragma unroll
        for (uint32_t j = 0; j < kSpecIters; j++) {
            const uint32_t c = lane + 64u * j;
            if (c < nchunks) S.raw[c] = sv[kSpec ? j : 0u];
        }
    } else if (staged) {
        const uint4* src = reinterpret_cast<const uint4*>(A.qwords) + chunk0;
        for (uint32_t c = lane; c < nchunks; c += 64u) S.raw[c] = src[c];
    }
    wave_sync();
    // One lane per (read, 64-letter word): the lane takes its 64 bytes as 16 words and turns them into the word's two plane
    // words with whole-word bit operations -- the letters' 2-bit codes are bits 1 and 2 of their ASCII bytes ((c >> 1) & 3 = x,
    // code = x ^ (x >> 1): A,C,G,T = 0..3), gathered four letters at a time.  (One letter per lane and three ballots per 64
    // letters of ONE read, the first form of this step, took 3.0 of the kernel's 6.8 ms on the headline batch.)
    if (lane < R) S.bad[lane] = 0u;
    if (lane == 0u) S.hasn = 0u;
    // a text without a letter that is not A,C,G,T (ArenaHeader::num_n = 0): such a letter of a READ agrees with nothing (N equals
    // only N, A.1), so no MEM holds it -- its windows are not looked up, its positions disagree in every compare, and the read
    // stays here; with such letters in the text a MEM may span one (N against N), and the read goes to the index walk
    const bool text_pure = ix.num_n == 0u;
    wave_sync();
    const uint32_t* rawwords = reinterpret_cast<const uint32_t*>(S.raw);
    const uint32_t* qw32 = reinterpret_cast<const uint32_t*>(A.qwords);
    // (the last 4-byte word that may be read: the 8-byte word that holds the batch's last letter ends with it; the offsets may
    //  start anywhere -- slamem_stream_* hands over windows of one array -- so the batch's end, not its size, says where)
    const uint64_t qlast32 = ((A.offsets[A.num_queries] + 7ull) >> 3) * 2ull - 1ull;
    constexpr uint32_t kRawLast = (R * NW * 4u + 1u) * 4u - 1u;
    for (uint32_t pb = 0; pb < nr * NW; pb += 64u) {
        const uint32_t pp = pb + lane, i = pp / NW, j = pp % NW;
        const bool on = pp < nr * NW;
        const uint32_t Li = __shfl(plen, (int)(on ? i : 0u));
        const uint64_t base = __shfl(off, (int)(on ? i : 0u));
        if (on && 64u * j < Li) {
            const uint32_t nv = Li - 64u * j < 64u ? Li - 64u * j : 64u;   // letters of this word
            const uint64_t a = base + 64u * j;                              // its first byte in the query buffer
            const uint32_t sh8 = (uint32_t)(a & 3u);
            uint32_t w[17];
            if (staged) {
                const uint32_t i0 = (uint32_t)((a - (chunk0 << 4)) >> 2);
#pragma unroll
                for (uint32_t t = 0; t < 17u; t++) w[t] = rawwords[i0 + t < kRawLast ? i0 + t : kRawLast];
            } else {
                const uint64_t i0 = a >> 2;
#pragma unroll
                for (uint32_t t = 0; t < 17u; t++) w[t] = qw32[i0 + t < qlast32 ? i0 + t : qlast32];
            }
            uint32_t p0lo = 0, p0hi = 0, p1lo = 0, p1hi = 0, bad = 0;
#pragma unroll
            for (uint32_t t = 0; t < 16u; t++) {
                const uint32_t d = __builtin_amdgcn_alignbyte(w[t + 1u], w[t], sh8);  // bytes a + 4t .. a + 4t + 3
                const uint32_t nb = nv > 4u * t ? (nv - 4u * t < 4u ? nv - 4u * t : 4u) : 0u;
                const uint32_t M = nb >= 4u ? 0x01010101u : (0x01010101u & ((1u << (8u * nb)) - 1u));
                const uint32_t B0 = d, B1 = d >> 1, B2 = d >> 2, B3 = d >> 3, B4 = d >> 4, B6 = d >> 6, B7 = d >> 7;
                // one of A,C,G,T in either case: 0100 0001, 0100 0011, 0100 0111, 0101 0100 with bit 5 free -- with t = "bits 2,1
                // are 1,0" (T's), bit 0 is the opposite of t and bit 4 equals it (of the 16 values of bits 4,2,1,0 exactly those four)
                const uint32_t tt = B2 & ~B1, good = (B0 ^ tt) & ~((B4 ^ tt) | B3 | B7) & B6;
                bad |= ~good & M;
                const uint32_t X = ((B1 ^ B2) & M) | ((B2 & M) << 4);
                const uint32_t G = (X | (X >> 7) | (X >> 14) | (X >> 21)) & 0xFFu;  // plane 0 of the four letters: bits 0-3, plane 1: bits 4-7
                if (t < 8u) { p0lo |= (G & 0xFu) << (4u * t); p1lo |= (G >> 4) << (4u * t); }
                else { p0hi |= (G & 0xFu) << (4u * (t - 8u)); p1hi |= (G >> 4) << (4u * (t - 8u)); }
            }
            S.pl[i][0][0][j] = u64_of(p0lo, p0hi);
            S.pl[i][0][1][j] = u64_of(p1lo, p1hi);
            uint32_t nlo = 0, nhi = 0;
            if (bad && text_pure) {  // (rare) which letters: the same gather over the "not one of A,C,G,T" bits
#pragma unroll
                for (uint32_t t = 0; t < 16u; t++) {
                    const uint32_t d = __builtin_amdgcn_alignbyte(w[t + 1u], w[t], sh8);
                    const uint32_t nb = nv > 4u * t ? (nv - 4u * t < 4u ? nv - 4u * t : 4u) : 0u;
                    const uint32_t M = nb >= 4u ? 0x01010101u : (0x01010101u & ((1u << (8u * nb)) - 1u));
                    const uint32_t B0 = d, B1 = d >> 1, B2 = d >> 2, B3 = d >> 3, B4 = d >> 4, B6 = d >> 6, B7 = d >> 7;
                    const uint32_t tt = B2 & ~B1, X = ~((B0 ^ tt) & ~((B4 ^ tt) | B3 | B7) & B6) & M;
                    const uint32_t G = (X | (X >> 7) | (X >> 14) | (X >> 21)) & 0xFu;
                    if (t < 8u) nlo |= G << (4u * t); else nhi |= G << (4u * (t - 8u));
                }
                S.hasn = 1u;
            } else if (bad) S.bad[i] = 1u;
            S.pn[i][j] = u64_of(nlo, nhi);
        } else if (on) {
            S.pl[i][0][0][j] = 0ull; S.pl[i][0][1][j] = 0ull; S.pn[i][j] = 0ull;
        }
        if (on && j == 0u) { S.pl[i][0][0][NW] = 0ull; S.pl[i][0][1][NW] = 0ull; S.pl[i][1][0][NW] = 0ull; S.pl[i][1][1][NW] = 0ull; S.pn[i][NW] = 0ull; }
    }
    wave_sync();
    // reverse strand: letter x of it is the complement (both plane bits flipped) of letter Li-1-x: the bit-reversed planes
    // (word 0 lowest, words behind the last are 0) come down by 64 * NW - Li letters
    for (uint32_t pb = 0; pb < nr * NW; pb += 64u) {
        const uint32_t pp = pb + lane, i = pp / NW, w = pp % NW;
        const bool on = pp < nr * NW;
        const uint32_t Li = __shfl(plen, (int)(on ? i : 0u));
        if (on) {
            const uint32_t sh = kMaxLen - Li, q = sh >> 6, r = sh & 63u;
#pragma unroll
            for (uint32_t pl = 0; pl < 2u; pl++) {
                const uint64_t* f = S.pl[i][0][pl];
                const uint64_t lo = w + q < NW ? __brevll(f[NW - 1u - (w + q)]) : 0ull, hi = w + q + 1u < NW ? __brevll(f[NW - 2u - (w + q)]) : 0ull;
                S.pl[i][1][pl][w] = Li ? ~funnel64(lo, hi, r) & bits_range(0, (int)Li - 64 * (int)w) : 0ull;
            }
        }
    }
    {   // a letter that is not A,C,G,T (N equals N in the reference, A.1): the index walk knows how
        const bool isbad = lane < nr && plen != 0u && S.bad[lane] != 0u;
        wflags |= (uint32_t)__ballot(isbad);
        if (isbad) nwin = 0;
        if (lane < nr) { S.len[lane] = nwin ? len : 0u; S.nwin[lane] = nwin; }
    }
    if (lane == 0u)
        for (uint32_t i = nr; i < R; i++) { S.len[i] = 0u; S.nwin[i] = 0u; }
    wave_sync();
    // ---- which windows are looked up, and when ----------------------------------------------------------------------------
    // Two rounds when the windows lie close (mstep > 1).  Round A looks up every mstep-th window of a read and compares its hits;
    // a compare that finds the exact run [a, b) around its window also reads the "occurs once" plane of the text under it: a
    // window of the other kind that lies inside [a, b) and whose k-mer occurs once in the text has its ONLY occurrence on this
    // diagonal, where it is accounted for -- its lookup would bring that one hit and nothing else, so it is not made.  Round B
    // looks up what is left: the windows with an error in them and the repeats (headline batch: 27 windows a read, 72 % of
    // them free of errors: 9 lookups in round A, 5-6 in round B).  A match that holds a round-A window is reported in round A
    // by the first of those (forward offset); one that holds none, in round B by its first window.
    // (as many windows apart as still overlap or touch, k / s -- 3 on the headline, 6 for 18-letter seeds at -l 20 -- and at
    //  least every second one while the windows lie within 12 letters of each other)
    const uint32_t mfit = k / s > 6u ? 6u : k / s;
    const uint32_t mstep = A.seed_step ? A.seed_step : mfit >= 3u ? mfit : s <= 12u ? 2u : 1u;

    const bool wave_hasn = __builtin_amdgcn_readfirstlane((int)S.hasn) != 0;
    uint32_t lg_slots = 0;  // lanes a read takes in the queueing of its windows: the power of two that holds the most windows of a read of the wave
    {
        const uint32_t nwl = lane < nr ? S.nwin[lane] : 0u;
#pragma unroll
        for (uint32_t t = 1, lg = 1; t <= 32u; t <<= 1, lg++)
            if (__ballot(nwl > t) != 0ull) lg_slots = lg;
    }
    if (lane < R) S.expl[lane] = 0ull;
    uint32_t njobs = 0, nmems = 0;

    // (the lookups and the compares in two forms, chosen per wave: with the code for reads that hold a letter which is not
    //  A,C,G,T, and without it -- merely compiling that code into the one form cost the headline batch 3.7 %)
    auto search = [&](auto with_n) {
    constexpr bool hasn = decltype(with_n)::value;
    // ---- lookups of one trip: one lane per window (ent = read << 8 | window): the seed table line of its canonical form ----
    auto trip = [&](uint32_t ent, bool act, uint32_t step) {
        const uint32_t rs = ent >> 8, wi = ent & 0xFFu;
        const uint32_t Lr = act ? S.len[rs] : 0u;
        const uint32_t o = wi * s;
        uint32_t want = 0, pal = 0;
        uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0, b2 = b0, b3 = b0;
        if (hasn && act) {  // a window that holds a letter which is not A,C,G,T occurs nowhere in the text: no lookup
            const uint64_t* PN = S.pn[rs];
            if (((uint32_t)funnel64(PN[o >> 6], PN[(o >> 6) + 1u], o & 63u) & kmask) != 0u) act = false;
        }
        if (act) {
            const uint32_t q = o >> 6, sh = o & 63u;
            const uint64_t* P0 = S.pl[rs][0][0];
            const uint64_t* P1 = S.pl[rs][0][1];
            const uint32_t f0 = (uint32_t)funnel64(P0[q], P0[q + 1u], sh) & kmask, f1 = (uint32_t)funnel64(P1[q], P1[q + 1u], sh) & kmask;
            uint32_t bucket, tagbits, orient;
            seed_place(f0, f1, k, ix.seed_log2, bucket, tagbits, orient, pal);  // (seeds of 17 / 18 letters: 64-bit keys, a wave-uniform branch)
            want = tagbits | (orient << 7);
            const uint4* B = reinterpret_cast<const uint4*>(ix.seed + bucket);
            b0 = B[0]; b1 = B[1]; b2 = B[2]; b3 = B[3];
        }
        if (kStats) n_win += (uint32_t)__popcll(__ballot(act));
        // tags that agree (bit 7 aside) -> hits; bit 7 differs: the text holds the window's reverse complement
        const uint32_t wbytes = want * 0x01010101u;
        const uint32_t x0 = b3.x ^ wbytes, x1 = b3.y ^ wbytes, x2 = b3.z ^ wbytes;
        const uint32_t m0 = ~((x0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) & 0x80808080u, m1 = ~((x1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) & 0x80808080u,
                       m2 = ~((x2 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) & 0x80808080u;
        const uint32_t count = b3.w;
        uint32_t hits = byte_tops(m0) | (byte_tops(m1) << 4) | (byte_tops(m2) << 8);
        const uint32_t rev = byte_tops(x0 & 0x80808080u) | (byte_tops(x1 & 0x80808080u) << 4) | (byte_tops(x2 & 0x80808080u) << 8);
        hits &= count >= kSeedSlots ? 0xFFFu : (1u << count) - 1u;
        // (one strand asked for: a hit in the other orientation is no MEM -- but in round A its compare still tells which windows
        //  of the other kind have their only occurrence there, on the strand that is not reported, and need no lookup: half the
        //  reads of a run come from that strand)
        if (strands == 1u && step == 1u) hits &= ~rev;
        if (!act) hits = 0;
        // more than twelve k-mers in the bucket: up to sixteen more in the spill list (looked at behind the twelve)
        const uint32_t xn = act && (count & kSeedSpilled) ? (count >> 24) & 0x1Fu : 0u, xo = (count & 0xFFFFFFu) << 2;
        if (act && count > kSeedSlots && !(count & kSeedSpilled)) {
            atomicOr(&S.flags, 1u << rs);
            if (kStats) atomicAdd(A.stats + SC_SEED_WHY + 1u, 1ull);
            hits = 0;
        }
        // a window that is its own reverse complement lies in both strands: every hit is a compare for each
        const bool both = pal != 0u && strands == 2u;
        if (kStats && both && hits) atomicAdd(A.stats + SC_SEED_WHY + 2u, 1ull);  // (counted, no longer a reason to leave the read)
        // Every hit is a compare -- except one whose neighbour (the lane before: the window `step` windows earlier in the forward
        // strand of the same read) hit the same diagonal while the two windows overlap or touch: they lie in the same match and
        // that window, or one before it, reports the MEM.  (The neighbour's first hit is looked at; a hit this misses is
        // sorted out by its compare.)
        const uint32_t e1 = hits ? (uint32_t)__ffs((int)hits) - 1u : 0u;
        const uint32_t p1 = sel12(b0, b1, b2, e1);
        const uint32_t st1 = (rev >> e1) & 1u;
        const uint32_t dist = step * s;  // letters between the neighbour's window and this one
        uint32_t pn = 0, pp1 = 0, ps1 = 0;  // the neighbour's first hit: position and strand (positions take all 32 bits at 3.1 Gbp)
        if (dist <= k) {
            pp1 = __shfl_up(p1, 1);
            ps1 = __shfl_up(hits ? st1 : 2u, 1);
            const uint32_t pent = __shfl_up(act ? ent : 0xFFFFFFFFu, 1);
            pn = (lane == 0u || wi < step || pent != ent - step || ps1 == 2u) ? 0u : 1u;
        }
        const uint32_t njobs0 = njobs;
        auto push = [&](bool hv, uint32_t p, uint32_t st) {
            const uint32_t w = st ? p + dist : p - dist;
            const bool job = hv && !(pn && pp1 == w && ps1 == st);
            const unsigned long long qb = __ballot(job);
            const uint32_t at = njobs + (uint32_t)__popcll(qb & below);
            if (job && at < SeedWave<NW, R>::kJobs) {
                S.job_p[at] = p;
                S.job_x[at] = (st ? Lr - k - o : o) | (st << 15) | (rs << 16);
            }
            njobs += (uint32_t)__popcll(qb);
        };
        uint32_t rem = hits;
        for (uint32_t it = 0; __ballot(rem != 0u) != 0ull; it++) {
            const bool hv = rem != 0u;
            const uint32_t e = it == 0u ? e1 : hv ? (uint32_t)__ffs((int)rem) - 1u : 0u;
            rem &= rem - 1u;
            const uint32_t p = it == 0u ? p1 : sel12(b0, b1, b2, e);
            push(hv, p, (rev >> e) & 1u);
            if (__ballot(hv && both) != 0ull) push(hv && both, p, 1u);
        }
        for (uint32_t e0 = 0; __ballot(e0 < xn) != 0ull; e0 += 4u) {  // (rare) four entries of the spill list per round
            const bool on = e0 < xn;
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
            if (on) {
                const uint4* P = reinterpret_cast<const uint4*>(ix.spill + xo + e0);
                v0 = P[0]; v1 = P[1];
            }
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {
                const uint32_t p = q == 0u ? v0.x : q == 1u ? v0.z : q == 2u ? v1.x : v1.z;
                const uint32_t d = ((q == 0u ? v0.y : q == 1u ? v0.w : q == 2u ? v1.y : v1.w) ^ want) & 0xFFu;
                const uint32_t st = d >> 7;
                const bool hv = on && e0 + q < xn && (d & 0x7Fu) == 0u && !(strands == 1u && st && step == 1u);
                if (__ballot(hv) == 0ull) continue;
                push(hv, p, st);
                if (__ballot(hv && both) != 0ull) push(hv && both, p, 1u);
            }
        }
        if (njobs > SeedWave<NW, R>::kJobs) {  // (many repeated windows) every read of this trip is left to K8
            if (act) atomicOr(&S.flags, 1u << rs);
            if (kStats && lane == 0u) atomicAdd(A.stats + SC_SEED_WHY + 3u, 1ull);
            njobs = njobs0;
        }
    };
    // the windows of a round, read after read, 64 to a trip (a ring of window ids in LDS takes what does not fill a trip yet)
    auto lookups = [&](uint32_t round) {
        const uint32_t step = round == 0u ? mstep : 1u;
        uint32_t cnt = 0;
        // (as many reads an iteration as fit the wave with a power of two of lanes each: two of the headline's, 27 windows a read,
        //  sixteen at -l 50, 4 windows a read)
        const uint32_t per = 64u >> lg_slots, wi = lane & ((1u << lg_slots) - 1u);
        const uint32_t wmod = wi - mstep * ((wi * (65536u / mstep + 1u)) >> 16);  // wi % mstep (wi < 64, mstep <= 6)
        for (uint32_t r = 0;; r += per) {
            if (r < nr) {
                const uint32_t rr = r + (lane >> lg_slots);
                // (a read that is left to K8 already -- a bucket beyond the spill list, lists that ran over: repeat families do
                //  that -- asks for nothing more)
                const uint32_t nw = (rr >= nr || ((S.flags >> rr) & 1u)) ? 0u : S.nwin[rr];
                const bool pred = wi < nw && (round == 0u ? wmod == 0u : wmod != 0u && ((S.expl[rr] >> wi) & 1ull) == 0ull);
                const unsigned long long pm = __ballot(pred);
                if (pred) S.ring[cnt + (uint32_t)__popcll(pm & below)] = (uint16_t)((rr << 8) | wi);
                cnt += (uint32_t)__popcll(pm);
            }
            const bool last = r >= nr;  // (what is left in the ring)
            if (cnt >= 64u || (last && cnt != 0u)) {
                wave_sync();
                const bool act = lane < cnt;
                const uint32_t ent = act ? S.ring[lane] : 0u;
                const uint32_t over = lane + 64u < cnt ? S.ring[lane + 64u] : 0u;
                wave_sync();
                if (lane + 64u < cnt) S.ring[lane] = (uint16_t)over;
                cnt = cnt > 64u ? cnt - 64u : 0u;
                trip(ent, act, step);
            }
            if (last) break;
        }
        wave_sync();
    };

    // ---- compares: one lane per job: the strand against the text on the hit's diagonal -------------------------------------
    auto compares = [&](uint32_t round) {
        const bool mark = round == 0u && mstep > 1u;
        const uint32_t ms = mstep * s;
        const uint32_t gone = S.flags;  // reads that are left to K8 already: their compares are not made
        for (uint32_t jb = 0; jb < njobs; jb += 64u) {
            const bool listed = jb + lane < njobs;
            const uint32_t p = listed ? S.job_p[jb + lane] : 0u, xx = listed ? S.job_x[jb + lane] : 0u;
            const uint32_t os = xx & 0x7FFFu, st = (xx >> 15) & 1u, jr = xx >> 16;
            const bool has = listed && ((gone >> jr) & 1u) == 0u;
            const uint32_t Lj = S.len[jr];
            const int64_t d = (int64_t)p - (int64_t)os;   // text position of the strand's first letter
            const int64_t u0 = d >> 6;
            const uint32_t sh = (uint32_t)(d & 63);
            bool is_mem = false;
            uint32_t key = 0, ref = 0, sig = 0;
            if (has) {
                constexpr int NU = (int)NW + 1;  // units of the text the strand faces: NW words from any start inside a unit
                int64_t ui[NU];
#pragma unroll
                for (int i = 0; i < NU; i++) { const int64_t u = u0 + i; ui[i] = u < 0 ? 0 : u > ulast ? ulast : u; }
                // (a unit is 32 bytes: planes, letter mask, occurs-once plane -- the units of a compare are 128 contiguous bytes)
                const uint4* T = reinterpret_cast<const uint4*>(ix.tpl);
                uint4 tu[NU], tx[NU];
#pragma unroll
                for (int i = 0; i < NU; i++) { tu[i] = T[2 * ui[i]]; tx[i] = T[2 * ui[i] + 1]; }
                uint64_t uq[NU], nu[NU];
                uint32_t anyn = 0;
#pragma unroll
                for (int i = 0; i < NU; i++) { nu[i] = u64_of(tx[i].x, tx[i].y); uq[i] = u64_of(tx[i].z, tx[i].w); anyn |= (tx[i].x | tx[i].y); }
                if (kStats && anyn) n_nm++;
                const uint64_t* R0 = S.pl[jr][st][0];
                const uint64_t* R1 = S.pl[jr][st][1];
                uint64_t mm[NW];
#pragma unroll
                for (int w = 0; w < (int)NW; w++)
                    mm[w] = (funnel64(u64_of(tu[w].x, tu[w].y), u64_of(tu[w + 1].x, tu[w + 1].y), sh) ^ R0[w]) |
                            (funnel64(u64_of(tu[w].z, tu[w].w), u64_of(tu[w + 1].z, tu[w + 1].w), sh) ^ R1[w]) | funnel64(nu[w], nu[w + 1], sh);
                if (hasn) {  // the strand's letters that are not A,C,G,T disagree with whatever they face
                    const uint64_t* PN = S.pn[jr];
                    if (st == 0u) {
#pragma unroll
                        for (int w = 0; w < (int)NW; w++) mm[w] |= PN[w];
                    } else {  // (the reverse strand: the forward plane bit-reversed, come down by 64 * NW - length letters)
                        const uint32_t shn = kMaxLen - Lj, qn = shn >> 6, rn = shn & 63u;
#pragma unroll
                        for (int w = 0; w < (int)NW; w++) {
                            const uint32_t x = (uint32_t)w + qn;
                            const uint64_t lo64 = x < NW ? __brevll(PN[NW - 1u - x]) : 0ull, hi64 = x + 1u < NW ? __brevll(PN[NW - 2u - x]) : 0ull;
                            mm[w] |= funnel64(lo64, hi64, rn);
                        }
                    }
                }
                // letters that face no text letter, or lie behind the strand, disagree
                const int lo = d < 0 ? (int)(-d) : 0;
                const int64_t room = (int64_t)ix.n - d;
                const int hi = room < (int64_t)Lj ? (int)room : (int)Lj;
#pragma unroll
                for (int w = 0; w < (int)NW; w++) mm[w] |= ~bits_range(lo - 64 * w, hi - 64 * w);
                // the run of agreeing letters around the window [os, os + k): [a, b)
                uint32_t a = 0, b = kMaxLen;
                bool broken = false;
#pragma unroll
                for (int w = 0; w < (int)NW; w++) {
                    const uint64_t lw = mm[w] & bits_range(0, (int)os - 64 * w);
                    if (lw) a = 64u * w + 64u - (uint32_t)__clzll((long long)lw);
                    if (mm[w] & bits_range((int)os - 64 * w, (int)(os + k) - 64 * w)) broken = true;
                }
#pragma unroll
                for (int w = (int)NW - 1; w >= 0; w--) {
                    const uint64_t hw = mm[w] & ~bits_range(0, (int)(os + k) - 64 * w);
                    if (hw) b = 64u * w + (uint32_t)__ffsll((unsigned long long)hw) - 1u;
                }
                // in forward offsets (the windows are laid out along the forward strand): the run [af, bf), the window at of
                const uint32_t af = st ? Lj - b : a, bf = st ? Lj - a : b, of = st ? Lj - k - os : os;
                bool owner;
                if (round == 0u) owner = !(of >= ms && of - ms >= af);  // the first round-A window inside the run
                else {
                    const uint32_t oa = (af + ms - 1u) / ms * ms;       // the first round-A window at or behind af: inside?
                    owner = !(mstep > 1u && oa + k <= bf) && !(of >= s && of - s >= af);
                }
                if (broken) {  // (cannot happen: the table is exact) -- leave the strand to K8
                    atomicOr(&S.flags, 1u << jr);
                    if (kStats) atomicAdd(A.stats + SC_SEED_WHY + 4u, 1ull);
                }
                else if (owner && b - a >= L && (strands == 2u || st == 0u)) {  // (one strand asked for: the other's compares only mark windows)
                    is_mem = true; key = (a << 16) | (b - a); ref = (uint32_t)(d + (int64_t)a);
                    // what the text goes on with behind the match: orders MEMs of one strand with the same start and length
                    // (phase 3) as the rows of the suffix array would
                    if (!anyn && b + kSigLetters <= kMaxLen && d + (int64_t)b + (int64_t)kSigLetters <= (int64_t)ix.n) {
                        const uint32_t wb = b >> 6, sb = b & 63u;
                        uint64_t l0 = 0, h0 = 0, l1 = 0, h1 = 0;
#pragma unroll
                        for (int w = 0; w < (int)NW; w++) {
                            const uint64_t t0 = funnel64(u64_of(tu[w].x, tu[w].y), u64_of(tu[w + 1].x, tu[w + 1].y), sh);
                            const uint64_t t1 = funnel64(u64_of(tu[w].z, tu[w].w), u64_of(tu[w + 1].z, tu[w + 1].w), sh);
                            if ((uint32_t)w == wb) { l0 = t0; l1 = t1; }
                            if ((uint32_t)w == wb + 1u) { h0 = t0; h1 = t1; }
                        }
                        sig = kSigKnown | ((uint32_t)funnel64(l0, h0, sb) & kSigMask) | (((uint32_t)funnel64(l1, h1, sb) & kSigMask) << kSigLetters);
                    }
                }
                if (mark && !broken && bf >= af + k) {
                    // the other windows inside the run whose k-mer occurs once in the text: accounted for
                    uint64_t uqs[NW];
#pragma unroll
                    for (int w = 0; w < (int)NW; w++) uqs[w] = funnel64(uq[w], uq[w + 1], sh);
                    unsigned long long done = 0ull;
                    for (uint32_t wi = (af + s - 1u) / s, wl = (bf - k) / s; wi <= wl; wi++) {
                        const uint32_t x = st ? Lj - k - wi * s : wi * s;  // the window's first letter in the strand
                        uint64_t word = 0ull;
#pragma unroll
                        for (int w = 0; w < (int)NW; w++) word = (x >> 6) == (uint32_t)w ? uqs[w] : word;
                        if ((word >> (x & 63u)) & 1ull) done |= 1ull << wi;
                    }
                    if (done) atomicOr(&S.expl[jr], done);
                }
            }
            if (kStats) {
                const uint32_t nh = (uint32_t)__popcll(__ballot(has));
                n_cmp += nh;
                if (mark && lane == 0u) atomicAdd(A.stats + SC_SEED_ONCE, (unsigned long long)nh);
            }
            const unsigned long long mb = __ballot(is_mem);
            if (is_mem) {
                const uint32_t at = nmems + (uint32_t)__popcll(mb & below);
                if (at < SeedWave<NW, R>::kMems) { S.mem_key[at] = key; S.mem_ref[at] = ref; S.mem_sig[at] = sig | ((2u * jr + st) << 25); }
                else { atomicOr(&S.flags, 1u << jr); if (kStats) atomicAdd(A.stats + SC_SEED_WHY + 5u, 1ull); }
            }
            nmems += (uint32_t)__popcll(mb);
        }
        wave_sync();
    };

    for (uint32_t round = 0; round < (mstep > 1u ? 2u : 1u); round++) {
        njobs = 0;
        lookups(round);
        compares(round);
    }
    };
    if (wave_hasn) search(std::true_type{}); else search(std::false_type{});
    {
        if (nmems > SeedWave<NW, R>::kMems) nmems = SeedWave<NW, R>::kMems;

        // ---- phase 3: the strands' MEMs in the reference's emission order ------------------------------------------------
        // rank of a MEM = MEMs of its strand that come before it (greater start, or equal start and greater length; same start
        // and length: tie_rank).  A strand is named by its bit of the flag word here (2 * read + strand)
        const uint32_t g0 = (uint32_t)(r0 * strands);
        auto rank_of = [&](uint32_t mi, uint32_t key, uint32_t g, uint32_t& rank, uint32_t& cnt, bool& tie) {
            rank = 0; cnt = 0; tie = false;
            for (uint32_t t2 = 0; t2 < nmems; t2++) {
                const uint32_t kk = S.mem_key[t2], gg = S.mem_sig[t2] >> 25;
                const bool same = gg == g;
                cnt += same ? 1u : 0u;
                rank += (same && kk > key) ? 1u : 0u;
                tie = tie || (same && kk == key && t2 != mi);
            }
        };
        // MEMs of a strand with the same start a and length: rows of one interval of the walk at position a, which the
        // reference lists around the interval it came up from (slamem.c:140,165): the rows above that child interval by
        // ascending row, then those below it by descending row; all of them ascending when the interval is the deepest one at a.
        // A deeper interval exists iff another match of the strand covers [a, b] and more to the right; a row lies above it iff
        // its text letter behind the match is smaller than the strand's; rows compare as the text behind their matches does.
        // Adds the tied MEMs that come before this one to rank; true: not decidable from kSigLetters letters (the strand is left to K8)
        auto tie_rank = [&](uint32_t mi, uint32_t key, uint32_t myref, uint32_t g, uint32_t sig, uint32_t& rank) -> bool {
            const uint32_t a = key >> 16, b = a + (key & 0xFFFFu);
            bool child = false;
            for (uint32_t t2 = 0; t2 < nmems; t2++) {
                const uint32_t kk = S.mem_key[t2], a2 = kk >> 16, b2 = a2 + (kk & 0xFFFFu);
                child = child || ((S.mem_sig[t2] >> 25) == g && kk != key && a2 <= a && b2 > b);
            }
            const uint32_t jr = g >> 1, st = g & 1u;
            uint32_t qb = 0;
            if (child) qb = (uint32_t)((S.pl[jr][st][0][b >> 6] >> (b & 63u)) & 1ull) | ((uint32_t)((S.pl[jr][st][1][b >> 6] >> (b & 63u)) & 1ull) << 1);
            auto letter = [&](uint32_t sg, uint32_t t) { return ((sg >> t) & 1u) | (((sg >> (t + kSigLetters)) & 1u) << 1); };
            // copies that go on alike for all the letters kept (long repeats): the rows themselves, from the text-ordered records
            // (the full layout has them: TextRec::row of a text position is the row of the suffix that starts there)
            const bool rows = ix.prec != nullptr;
            const uint32_t my_side = child && letter(sig, 0u) > qb ? 1u : 0u;
            bool bad = (sig & kSigKnown) == 0u && (child || !rows);  // (which side of the child interval: the letter behind the match says)
            for (uint32_t t2 = 0; t2 < nmems; t2++) {
                const uint32_t so = S.mem_sig[t2];
                if (t2 == mi || (so >> 25) != g || S.mem_key[t2] != key) continue;
                const bool known = (so & sig & kSigKnown) != 0u;
                if (!known && (child || !rows)) { bad = true; continue; }
                const uint32_t o_side = child && letter(so, 0u) > qb ? 1u : 0u;
                bool before;
                if (o_side != my_side) before = o_side == 0u;
                else {
                    const uint32_t x = sig ^ so, diff = known ? (x | (x >> kSigLetters)) & kSigMask : 0u;
                    bool smaller;
                    if (diff != 0u) {
                        const uint32_t t = (uint32_t)__ffs((int)diff) - 1u;
                        smaller = letter(so, t) < letter(sig, t);
                    } else if (rows) smaller = ix.prec[S.mem_ref[t2]].row < ix.prec[myref].row;
                    else { bad = true; continue; }
                    before = my_side == 0u ? smaller : !smaller;
                }
                rank += before ? 1u : 0u;
            }
            return bad;
        };
        auto emit_mem = [&](uint32_t fl, uint32_t key, uint32_t ref, uint32_t gb, uint32_t rank, uint32_t cnt) {
            if ((fl >> (gb >> 1)) & 1u) return;
            const uint32_t g = g0 + (strands == 2u ? gb : gb >> 1);
            emit3_at(A, g, rank, 0u, ref, key >> 16, (key & 0xFFFFu) | 0x80000000u);  // bit 31: ref_pos is the text position (K9)
            if (rank == 0u) A.block_counts[g] = cnt;
            if (kStats) n_mem++;
        };
        if (nmems <= 64u) {  // the usual case: one MEM per lane, one loop
            const bool has = lane < nmems;
            const uint32_t key = has ? S.mem_key[lane] : 0u, ref = has ? S.mem_ref[lane] : 0u, g = has ? S.mem_sig[lane] >> 25 : 0xFFFFFFFFu;
            // the MEMs of a strand find each other through a mask per strand (a strand has a few, the wave's list some forty)
            if (lane < 2u * R) S.smask[lane] = 0ull;
            wave_sync();
            if (has) atomicOr(&S.smask[g], 1ull << lane);
            wave_sync();
            const unsigned long long mine = has ? S.smask[g] : 0ull;
            uint32_t rank = 0;
            const uint32_t cnt = (uint32_t)__popcll(mine);
            bool tie = false;
            for (unsigned long long o = mine & ~(1ull << lane); __ballot(o != 0ull) != 0ull; o &= o - 1ull) {
                if (o == 0ull) continue;
                const uint32_t kk = S.mem_key[__ffsll(o) - 1];
                rank += kk > key ? 1u : 0u;
                tie = tie || kk == key;
            }
            if (has && tie && tie_rank(lane, key, ref, g, S.mem_sig[lane], rank)) {
                atomicOr(&S.flags, 1u << (g >> 1));
                if (kStats) atomicAdd(A.stats + SC_SEED_WHY + 6u, 1ull);
            }
            wave_sync();
            if (has) emit_mem(S.flags | wflags, key, ref, g, rank, cnt);
        } else {
            for (uint32_t pass = 0; pass < 2u; pass++) {
                const uint32_t fl = pass ? (S.flags | wflags) : 0u;
                for (uint32_t m0i = 0; m0i < nmems; m0i += 64u) {
                    const uint32_t mi = m0i + lane;
                    const bool has = mi < nmems;
                    const uint32_t key = has ? S.mem_key[mi] : 0u, ref = has ? S.mem_ref[mi] : 0u, g = has ? S.mem_sig[mi] >> 25 : 0xFFFFFFFFu;
                    uint32_t rank, cnt;
                    bool tie;
                    rank_of(mi, key, g, rank, cnt, tie);
                    const bool bad = has && tie && tie_rank(mi, key, ref, g, S.mem_sig[mi], rank);
                    if (!pass) { if (bad) atomicOr(&S.flags, 1u << (g >> 1)); }
                    else if (has) emit_mem(fl, key, ref, g, rank, cnt);
                }
                wave_sync();
            }
        }
    }
    wave_sync();
    const uint32_t fl = S.flags | wflags;
    {   // the strands left to K8: their flags, and their numbers appended to K8's work list (the list's order is the order the
        // waves end in: it decides which lane scans a strand, never what is reported for it)
        const bool isleft = lane < nr * strands && ((fl >> (strands == 2u ? lane >> 1 : lane)) & 1u) != 0u;
        if (lane < nr * strands) alive[r0 * strands + lane] = isleft ? 1 : 0;
        const unsigned long long lm = __ballot(isleft);
        if (lm != 0ull) {
            unsigned int at = 0;
            if (lane == 0u) at = atomicAdd(A.seed_left_count, (unsigned int)__popcll(lm));
            at = (unsigned int)__builtin_amdgcn_readfirstlane((int)at);
            if (isleft) A.seed_left_ids[at + (uint32_t)__popcll(lm & below)] = (uint32_t)(r0 * strands + lane);
        }
    }
    if (kStats) {
        stat_flush<kStats>(A.stats + SC_SEED_NMASKS, n_nm);
        stat_flush<kStats>(A.stats + SC_SEED_MEMS, n_mem);
        if (lane == 0u) {
            atomicAdd(A.stats + SC_SEED_WINDOWS, (unsigned long long)n_win);
            atomicAdd(A.stats + SC_SEED_COMPARES, (unsigned long long)n_cmp);
            atomicAdd(A.stats + SC_SEED_READS, (unsigned long long)nr);
            atomicAdd(A.stats + SC_SEED_QBYTES, (unsigned long long)(__builtin_amdgcn_readlane((int)(uint32_t)off, (int)nr) - __builtin_amdgcn_readlane((int)(uint32_t)off, 0)));
            atomicAdd(A.stats + SC_SEED_LEFT, (unsigned long long)(strands * (uint32_t)__popc(fl)));
            atomicAdd(A.stats + SC_SEED_WHY + 0u, (unsigned long long)__popc(wflags));  // reads left before any lookup: too long, or a letter that is not A,C,G,T (and trips whose compares did not fit)
        }
    }
}

// K9 for v3: inline slots and overflow records -> grouped output, BWT rows resolved to text positions here
__global__ void __launch_bounds__(256) k_place_inline(const RawRow* __restrict__ inl, uint64_t stride, const uint32_t* __restrict__ counts,
                                                      const uint64_t* __restrict__ item_off, uint64_t nitems,
                                                      const uint32_t* __restrict__ sa, uint64_t capacity,
                                                      slamem_mem* __restrict__ out, const uint8_t* __restrict__ inline_valid) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nitems) return;
    // three memory phases whatever the count (count + offset; the rows; their text positions), not two per MEM
    uint32_t cnt = counts[g];
    uint64_t off = item_off[g];
    uint32_t iv = inline_valid ? inline_valid[g] : 0u;  // (kDefer) a strand with deferred jobs: only what it reported before the first
    asm volatile("" : "+v"(cnt), "+v"(off), "+v"(iv));
    if (cnt > kInlineMems) cnt = kInlineMems;
    if (iv && cnt > iv - 1u) cnt = iv - 1u;
    if (cnt == 0u) return;
    static_assert(kInlineMems == 4, "k_place_inline is written for four inline slots");
    const RawRow* in = inl + g;
    const RawRow none = {0u, 0u, 0x80000000u};  // (bit 31: no suffix-array read)
    RawRow r0 = in[0], r1 = none, r2 = none, r3 = none;
    if (cnt > 1u) r1 = in[stride];
    if (cnt > 2u) r2 = in[2u * stride];
    if (cnt > 3u) r3 = in[3u * stride];
    asm volatile("" : "+v"(r0.row), "+v"(r0.pos), "+v"(r0.len), "+v"(r1.row), "+v"(r1.pos), "+v"(r1.len));
    asm volatile("" : "+v"(r2.row), "+v"(r2.pos), "+v"(r2.len), "+v"(r3.row), "+v"(r3.pos), "+v"(r3.len));
    // bit 31 of the length: `row` already is the text position (a MEM emitted from a direct run)
    uint32_t p0 = r0.row, p1 = r1.row, p2 = r2.row, p3 = r3.row;
    if (!(r0.len >> 31)) p0 = sa[r0.row];
    if (cnt > 1u && !(r1.len >> 31)) p1 = sa[r1.row];
    if (cnt > 2u && !(r2.len >> 31)) p2 = sa[r2.row];
    if (cnt > 3u && !(r3.len >> 31)) p3 = sa[r3.row];
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    if (off < capacity) out[off] = slamem_mem{p0, r0.pos, r0.len & 0x7FFFFFFFu};
    if (cnt > 1u && off + 1u < capacity) out[off + 1u] = slamem_mem{p1, r1.pos, r1.len & 0x7FFFFFFFu};
    if (cnt > 2u && off + 2u < capacity) out[off + 2u] = slamem_mem{p2, r2.pos, r2.len & 0x7FFFFFFFu};
    if (cnt > 3u && off + 3u < capacity) out[off + 3u] = slamem_mem{p3, r3.pos, r3.len & 0x7FFFFFFFu};
}

// K9 when no record is cut into slices (item g is strand block g): offsets in two levels.  The MEMs of every 64 consecutive strands
// are summed (k_group_totals, one wave per group), the 1/64 as many sums are scanned, and the placement kernel -- one strand a
// lane, as k_place_inline -- finds a strand's offset from its group's prefix and a prefix over the wave's lanes, writes the
// block offsets and places the inline rows.  (The plain form scans 20 M counts into 20 M offsets -- 160 MB written and read
// again -- before the placement reads both: 0.16 + 0.24 ms on the headline batch.)
// (64-bit sums: a strand may report up to 2^28 MEMs, 64 of them more than 32 bits hold)
__global__ void __launch_bounds__(256) k_group_totals(const uint32_t* __restrict__ counts, uint64_t nitems, uint64_t* __restrict__ totals) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t v = g < nitems ? (uint64_t)counts[g] : 0ull;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += u64_of(__shfl_down((uint32_t)v, d), __shfl_down((uint32_t)(v >> 32), d));
    if ((threadIdx.x & 63u) == 0u && (g >> 6) <= (nitems >> 6)) totals[g >> 6] = v;
}
__global__ void __launch_bounds__(256) k_place_grouped(const uint32_t* __restrict__ counts, uint64_t nitems, const uint64_t* __restrict__ group_prefix,
                                                       uint64_t* __restrict__ block_offsets, const RawRow* __restrict__ inl, uint64_t stride,
                                                       const uint32_t* __restrict__ sa, uint64_t capacity, slamem_mem* __restrict__ out,
                                                       const uint8_t* __restrict__ inline_valid) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool item = g < nitems;
    const uint32_t full = item ? counts[g] : 0u;
    uint32_t iv = (item && inline_valid) ? inline_valid[g] : 0u;  // (kDefer) a strand with deferred jobs: only what it reported before the first
    uint64_t gp = g <= nitems ? group_prefix[g >> 6] : 0ull;
    RawRow r0 = {0u, 0u, 0x80000000u};  // (bit 31: no suffix-array read)
    if (full != 0u) r0 = inl[g];         // (the first phase: count, group prefix and first row go out together)
    asm volatile("" : "+v"(iv), "+v"(gp), "+v"(r0.row), "+v"(r0.pos), "+v"(r0.len));
    uint64_t incl = full;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = u64_of(__shfl_up((uint32_t)incl, d), __shfl_up((uint32_t)(incl >> 32), d));
        if ((int)lane >= d) incl += o;
    }
    const uint64_t off = gp + (incl - full);
    if (g <= nitems) block_offsets[g] = off;  // (g == nitems: the total)
    uint32_t cnt = full > kInlineMems ? kInlineMems : full;
    if (iv && cnt > iv - 1u) cnt = iv - 1u;
    if (cnt == 0u) return;
    static_assert(kInlineMems == 4, "k_place_grouped is written for four inline slots");
    const RawRow none = {0u, 0u, 0x80000000u};
    RawRow r1 = none, r2 = none, r3 = none;
    if (cnt > 1u) r1 = inl[stride + g];
    if (cnt > 2u) r2 = inl[2u * stride + g];
    if (cnt > 3u) r3 = inl[3u * stride + g];
    asm volatile("" : "+v"(r1.row), "+v"(r1.pos), "+v"(r1.len), "+v"(r2.row), "+v"(r2.pos), "+v"(r2.len), "+v"(r3.row), "+v"(r3.pos), "+v"(r3.len));
    // bit 31 of the length: `row` already is the text position (K8s, or a MEM emitted from a direct run)
    uint32_t p0 = r0.row, p1 = r1.row, p2 = r2.row, p3 = r3.row;
    if (!(r0.len >> 31)) p0 = sa[r0.row];
    if (cnt > 1u && !(r1.len >> 31)) p1 = sa[r1.row];
    if (cnt > 2u && !(r2.len >> 31)) p2 = sa[r2.row];
    if (cnt > 3u && !(r3.len >> 31)) p3 = sa[r3.row];
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    if (off < capacity) out[off] = slamem_mem{p0, r0.pos, r0.len & 0x7FFFFFFFu};
    if (cnt > 1u && off + 1u < capacity) out[off + 1u] = slamem_mem{p1, r1.pos, r1.len & 0x7FFFFFFFu};
    if (cnt > 2u && off + 2u < capacity) out[off + 2u] = slamem_mem{p2, r2.pos, r2.len & 0x7FFFFFFFu};
    if (cnt > 3u && off + 3u < capacity) out[off + 3u] = slamem_mem{p3, r3.pos, r3.len & 0x7FFFFFFFu};
}

// (the number of listed records stays on the device -- *listed, written by K8's atomics -- so that K9 is launched right
//  behind K8 without a host round trip in between; the grid is fixed and strides over the list)
__global__ void __launch_bounds__(256) k_place_overflow(const RawKey* __restrict__ key, const slamem_mem* __restrict__ raw,
                                                        const unsigned long long* __restrict__ listed,
                                                        const uint64_t* __restrict__ item_off,
                                                        const uint8_t* __restrict__ item_attempt,
                                                        const uint32_t* __restrict__ sa, uint64_t capacity,
                                                        slamem_mem* __restrict__ out, const uint32_t* __restrict__ raw_seg,
                                                        const uint32_t* __restrict__ pool) {
    uint64_t count = *listed;
    if (count > capacity) count = capacity;  // (records beyond the capacity were never stored: emit3_at)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        RawKey kk = key[i];
        if (kk.block == 0xFFFFFFFFu) continue;                  // a place some wave reserved and did not use (kChunk)
        if ((kk.k >> 28) != item_attempt[kk.block]) continue;  // written by an attempt that was abandoned
        slamem_mem m = raw[i];
        uint64_t pos = item_off[kk.block] + (kk.k & 0x0FFFFFFFu);
        if (raw_seg) {  // (kDefer) a provisional number: the MEMs of the strand's jobs before this record come first
            const uint32_t sg = raw_seg[i];
            if (sg != 0xFFFFFFFFu) pos += pool[sg];
        }
        if (pos >= capacity) continue;
        if (m.length >> 31) m.length &= 0x7FFFFFFFu;  // ref_pos already is the text position
        else m.ref_pos = sa[m.ref_pos];
        out[pos] = m;
    }
}

// ---- work items ------------------------------------------------------------------------------------------
// slices per strand of every record (>= 1 so that empty records still own an (empty) output block)
// (and the 64-bit words one strand of the record occupies in the packed copy: a multiple of two, i.e. 16-byte blocks)
__global__ void __launch_bounds__(256) k_item_counts(const uint64_t* __restrict__ offsets, uint32_t nq, uint32_t slice_len,
                                                     uint32_t* __restrict__ cnt, uint32_t* __restrict__ wps,
                                                     unsigned int* __restrict__ most) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0, w = 0;
    if (q < nq) {
        uint64_t len = offsets[q + 1] - offsets[q];
        c = slice_len ? (uint32_t)((len + slice_len - 1) / slice_len) : 1u;
        if (c == 0) c = 1;
        w = 2u * (uint32_t)((len + 31u) >> 5);
    }
    // the most items any record has (1: no record is cut into slices -- item g is strand block g, and a batch on the seed
    // path needs no scan of these counts)
    if (c > 1u) atomicMax(most, c);
    if (q > nq) return;
    cnt[q] = c;  // cnt[nq] = 0: the scan of nq+1 values leaves the total in first[nq]
    wps[q] = w;
}

// items in emission order: per record the forward strand's slices right to left, then the reverse strand's
// item_pk: word offset of the item's strand block in the packed copy (two zero words lead the buffer); every slice of a
// strand carries the block's offset, positions stay relative to the strand
__global__ void __launch_bounds__(256) k_item_fill(const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ first,
                                                   const uint64_t* __restrict__ wscan, uint32_t nq, uint32_t strands,
                                                   ItemDesc* __restrict__ items, uint64_t* __restrict__ item_pk,
                                                   uint32_t* __restrict__ item_block) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint64_t o0 = offsets[q];
    uint32_t len = (uint32_t)(offsets[q + 1] - o0);
    uint32_t f = first[q], cnt = first[q + 1] - f;
    uint64_t w0 = wscan ? wscan[q] : 0ull, wps = wscan ? wscan[q + 1] - w0 : 0ull;
    for (uint32_t s = 0; s < strands; s++)
        for (uint32_t c = 0; c < cnt; c++) {
            uint64_t it = (uint64_t)strands * f + (uint64_t)s * cnt + (cnt - 1u - c);
            items[it] = ItemDesc{o0, len, c | (s << 31)};
            if (item_pk) item_pk[it] = 2ull + (uint64_t)strands * w0 + (uint64_t)s * wps;
            if (item_block) item_block[it] = (uint32_t)(q * strands + s);
        }
}

// public strand blocks: block (q, s) starts where its first item starts
__global__ void __launch_bounds__(256) k_block_offsets(const uint32_t* __restrict__ first, const uint64_t* __restrict__ item_off,
                                                       uint32_t nq, uint32_t strands, uint64_t nitems,
                                                       uint64_t* __restrict__ block_offsets) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t nblocks = (uint64_t)nq * strands;
    if (b > nblocks) return;
    if (b == nblocks) { block_offsets[b] = item_off[nitems]; return; }
    uint32_t q = (uint32_t)(b / strands), sidx = (uint32_t)(b % strands);
    uint32_t f = first[q], cnt = first[q + 1] - f;
    block_offsets[b] = item_off[(uint64_t)strands * f + (uint64_t)sidx * cnt];
}

// ------------------------------------------------------------------------------------------
// fine-grained batch kernels (one lane per element) -- same device functions as K8
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_follow_batch(IndexView ix, const char* __restrict__ letters, uint32_t* top,
                                                      uint32_t* bot, uint32_t* size_out, uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t t = top[i], b = bot[i];
    if (t > b || b > ix.n) { size_out[i] = 0; return; }
    bool ok = follow(ix, ascii_code_q((uint8_t)letters[i]), t, b);
    if (ok) { top[i] = t; bot[i] = b; size_out[i] = b - t + 1u; }
    else size_out[i] = 0;
}

__global__ void __launch_bounds__(256) k_parent_batch(IndexView ix, uint32_t* top, uint32_t* bot, int32_t* depth_out,
                                                      uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t t = top[i], b = bot[i];
    if (t > b || b > ix.n) { depth_out[i] = -2; return; }
    depth_out[i] = parent(ix, t, b);
    top[i] = t;
    bot[i] = b;
}

__global__ void __launch_bounds__(256) k_locate_batch(IndexView ix, const uint32_t* __restrict__ rows, uint32_t* out,
                                                      uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t r = rows[i];
    out[i] = r <= ix.n ? ix.sa[r] : 0xFFFFFFFFu;
}

__global__ void __launch_bounds__(256) k_bwtchar_batch(IndexView ix, const uint32_t* __restrict__ rows, char* out,
                                                       uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t r = rows[i];
    const char L[] = {'$', 'N', 'A', 'C', 'G', 'T'};
    out[i] = r <= ix.n ? L[bwt_code(ix, r)] : '?';
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {
inline unsigned grid_for(uint64_t items, unsigned block = 256) { return (unsigned)((items + block - 1) / block); }
inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

constexpr uint64_t kK8Waves = 4096;  // what the chip holds at 4 waves per SIMD: no workgroup waits behind the grid
constexpr uint64_t kCarryLanes = kK8Waves * 64;

struct WorkspaceLayout {
    uint64_t off_total, off_cnt, off_first, off_scan32, off_items, off_counts, off_attempt, off_alive, off_workids, off_select, select_bytes, off_itemoff, off_gsum, off_gpre, off_rawkey,
        off_rawmem, off_inline, off_scan, scan_bytes, off_wps, off_wscan, off_itempk, off_pq, pq_bytes, off_pq2, off_itemflags, off_mamstate, off_mamrun, off_itemblock, off_slicestate, off_carry, off_defscal, off_jobq, off_pool, off_rawseg, off_deflist, pool_cap, max_bounds, max_items, bytes;
};

// max_items bounds the work items of ANY batch with this many records and characters
WorkspaceLayout layout_workspace(uint64_t num_queries, uint64_t strands, uint64_t query_bytes, uint64_t capacity) {
    WorkspaceLayout w;
    w.max_items = strands * (num_queries + query_bytes / kSliceLen + 1);
    uint64_t off = 0;
    w.off_total = off;    off = align_up(off + 64 + SC_COUNT * 8, 256);  // scalars, then the diagnostic counters
    w.off_cnt = off;      off = align_up(off + (num_queries + 2) * 4, 256);
    w.off_first = off;    off = align_up(off + (num_queries + 2) * 4, 256);
    w.off_scan32 = off;   off = align_up(off + scan_u32_tmp_words(num_queries + 1) * 4, 256);
    w.off_items = off;    off = align_up(off + w.max_items * sizeof(ItemDesc), 256);
    w.off_counts = off;   off = align_up(off + (w.max_items + 1) * 4, 256);
    w.off_attempt = off;  off = align_up(off + w.max_items, 256);
    w.off_alive = off;    off = align_up(off + w.max_items, 256);
    w.off_workids = off;  off = align_up(off + w.max_items * 4, 256);
    {
        size_t need2 = 0;
        (void)select_indices_u32(nullptr, need2, nullptr, nullptr, nullptr, w.max_items, 0);
        w.select_bytes = need2;
        w.off_select = off;
        off = align_up(off + need2, 256);
    }
    w.off_itemoff = off;  off = align_up(off + (w.max_items + 1) * 8, 256);
    w.off_gsum = off;     off = align_up(off + ((w.max_items >> 6) + 2) * 8, 256);  // K9, two levels: sums of 64 strands' counts ...
    w.off_gpre = off;     off = align_up(off + ((w.max_items >> 6) + 2) * 8, 256);  // ... and their prefixes
    w.off_rawkey = off;   off = align_up(off + capacity * sizeof(RawKey), 256);
    w.off_rawmem = off;   off = align_up(off + capacity * sizeof(slamem_mem), 256);
    w.off_inline = off;   off = align_up(off + w.max_items * kInlineMems * sizeof(RawRow), 256);
    size_t need = 0;
    (void)scan_sum_exclusive_u32_u64(nullptr, need, nullptr, nullptr, w.max_items, 0);
    w.scan_bytes = need;
    w.off_scan = off;     off = align_up(off + need, 256);
    // the packed copy of the strands (k_pack_queries): per strand 2 * ceil(len / 32) words <= len / 16 + 2
    w.off_wps = off;      off = align_up(off + (num_queries + 2) * 4, 256);
    w.off_wscan = off;    off = align_up(off + (num_queries + 2) * 8, 256);
    w.off_itempk = off;   off = align_up(off + w.max_items * 8, 256);
    w.pq_bytes = 8 * (4 + strands * (query_bytes / 16 + 2 * num_queries + 2));
    w.off_pq = off;       off = align_up(off + w.pq_bytes, 256);
    w.off_pq2 = off;      off = align_up(off + w.pq_bytes / 2 + 64, 256);
    w.off_itemflags = off; off = align_up(off + w.max_items, 256);
    // -mam over slices: two states per slice boundary, the list of slices to scan again, the strand block of every item
    w.max_bounds = strands * (query_bytes / kSliceLen + 1);
    w.off_mamstate = off; off = align_up(off + 2 * w.max_bounds * sizeof(MamState), 256);
    w.off_mamrun = off;   off = align_up(off + w.max_bounds * 4, 256);
    w.off_itemblock = off; off = align_up(off + w.max_items * 4, 256);
    w.off_slicestate = off; off = align_up(off + w.max_bounds * sizeof(SliceState), 256);  // start states of slices (k_slice_states)
    w.off_carry = off;    off = align_up(off + kCarryLanes * sizeof(CarryRec), 256);  // lanes handed to the next launch (kCarry)
    // kDefer: the queue of enumeration jobs, the strands' job counters, the records' places in them, the list of such strands.
    // A counter per query position (and a few per strand) covers every job a batch can have; capped at 32 M (a batch of 10 M
    // reads in which every tenth strand is repeat-heavy): a strand that finds the pool full runs its jobs in its wave
    w.pool_cap = strands * (query_bytes + 6 * num_queries) + 64;
    if (w.pool_cap > (32ull << 20)) w.pool_cap = 32ull << 20;
    w.off_defscal = off;  off = align_up(off + 64, 256);
    w.off_jobq = off;     off = align_up(off + (w.pool_cap + 64 * kK8Waves) * sizeof(JobRec), 256);
    w.off_pool = off;     off = align_up(off + (w.pool_cap + 16) * 4, 256);
    w.off_rawseg = off;   off = align_up(off + capacity * 4, 256);
    w.off_deflist = off;  off = align_up(off + w.max_items * 8, 256);
    w.bytes = off;
    return w;
}
}  // namespace

uint64_t find_mems_workspace_bytes(uint64_t num_queries, int both_strands, uint64_t query_bytes, uint64_t mems_capacity) {
    return layout_workspace(num_queries, both_strands ? 2 : 1, query_bytes, mems_capacity).bytes;
}

// SLAMEM_MAM_WHOLE=1: -mam with one lane per whole strand (k_find_mams), no slices
static bool mam_whole_strands() {
    static const bool on = [] { const char* v = getenv("SLAMEM_MAM_WHOLE"); return v && atoi(v) != 0; }();
    return on;
}

struct SearchJob {
    const slamem_index* idx = nullptr;
    const void* queries_dev = nullptr;
    const uint64_t* offsets_dev = nullptr;
    uint32_t num_queries = 0, min_len = 0, strands = 1;
    uint64_t query_bytes = 0, mems_capacity = 0, workspace_bytes = 0, num_blocks = 0, nitems = 0, total = 0;
    int both_strands = 0, match_type = 0;
    slamem_mem* mems_dev = nullptr;
    uint64_t* block_offsets_dev = nullptr;
    void* workspace_dev = nullptr;
    WorkspaceLayout w;
    SearchArgs A;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // before K8a, after K8, after K9, after K8a, before K8, after K7q
    bool want_stats = false, prefiltered = false, timed_k8 = false, launched = false;
    bool rawkey_marked = false;  // the overflow list already carries its "unused" marks (kChunk)
    bool deferred = false;       // this launch ran K8's kDefer instantiation: k_enum_jobs and k_defer_prefix follow
    bool speculate = false;  // the caller starts again when a record turns out longer than a slice (saw_long)
    bool speculated = false, saw_long = false;
    bool seed_path = false;  // tables(): this batch takes the seed path
    bool seeded = false;  // this batch's MEMs come from K8s (k_seed_mems); K8 scans only the strands it left
    uint32_t seed_words = 0, seed_words_avg = 0;  // plane words a strand of this batch's K8s launch, and what the average read length alone asks for
    bool mam_v3 = false;  // -mam on a batch without long records: K8's kMam instantiation (set by tables())
    unsigned long long scal_own[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t k8_wave_cap = 0;  // waves of this batch's K8 (0: as many as the chip holds); a pipeline that keeps two K8 launches in flight gives each a part of the chip
    uint32_t slices_hint = 0xFFFFFFFFu;  // a caller that has the offsets on the host and knows the slice count (no record longer than a slice: one per record) saves tables() its round trip
    unsigned long long* h_scal = scal_own;  // where search() has the scalars copied: [0..7] the scalar block, [8] all MEMs; pinned memory if the caller has some
    ~SearchJob();
    int init(const slamem_index* idx_, const void* queries_dev_, const uint64_t* offsets_dev_, uint32_t num_queries_,
             uint64_t query_bytes_, uint32_t min_len_, int both_strands_, int match_type_, slamem_mem* mems_dev_,
             uint64_t mems_capacity_, uint64_t* block_offsets_dev_, void* workspace_dev_, uint64_t workspace_bytes_);
    int tables(hipStream_t stream);
    int prep(hipStream_t stream);
    int search(hipStream_t stream) { int rc = search_k8(stream, nullptr, false); return rc != SLAMEM_OK ? rc : place(stream); }
    // K8 alone.  carry_from: the job of the previous launch on this stream whose unfinished lanes this launch takes in (it
    // must have been launched with carry_out); carry_out: end when the work list is empty and pass the unfinished lanes on
    // (the NEXT launch, or flush(), finishes them -- only then may place() of this job follow)
    int search_k8(hipStream_t stream, SearchJob* carry_from, bool carry_out);
    bool enum_chunks() const;                        // the text is repeat-rich at this minimum length: K8's kChunk instantiation
    bool can_carry() const;                          // this batch runs on an instantiation that can pass lanes on / take them in
    bool can_carry_into(const SearchJob& next) const;  // ... and `next` can take them
    int flush(hipStream_t stream);                   // a launch without new work that finishes the lanes this job passed on
    int place(hipStream_t stream);                   // K9 + the batch's scalars to the host
    int collect();
    bool carried_out = false;
    bool chunked = false;  // this launch ran the kChunk instantiation (its overflow list has unused places)
};

// One batch through the search, in steps that a caller may issue apart and on different streams (slamem_stream_* does: the
// preparation of batches b+1, b+2 runs on its own stream and fills the GPU while the last waves of K8(b) drain; K8 and K9 of
// all batches follow each other on the search stream with no host round trip in between):
//   tables()  work-item counts and offsets (one small sync: the number of items sizes the grids)
//   prep()    K8a prefilter, work-list compaction, slice states, K7q packing -- asynchronous
//   search()  K8 search, per-item counts -> offsets, K9 placement, the batch's scalars to the host -- asynchronous
//   collect() AFTER the search stream has finished those: totals, capacity check, timings
// find_mems_device() runs them in a row on one stream.
int SearchJob::init(const slamem_index* idx_, const void* queries_dev_, const uint64_t* offsets_dev_, uint32_t num_queries_,
                    uint64_t query_bytes_, uint32_t min_len_, int both_strands_, int match_type_, slamem_mem* mems_dev_,
                    uint64_t mems_capacity_, uint64_t* block_offsets_dev_, void* workspace_dev_, uint64_t workspace_bytes_) {
    idx = idx_; queries_dev = queries_dev_; offsets_dev = offsets_dev_; num_queries = num_queries_; query_bytes = query_bytes_;
    min_len = min_len_; both_strands = both_strands_; match_type = match_type_; mems_dev = mems_dev_; mems_capacity = mems_capacity_;
    block_offsets_dev = block_offsets_dev_; workspace_dev = workspace_dev_; workspace_bytes = workspace_bytes_;
    total = 0; nitems = 0; prefiltered = false; timed_k8 = false; launched = false;
    if (!idx || !offsets_dev || !block_offsets_dev || !workspace_dev || (!mems_dev && mems_capacity) || (!queries_dev && num_queries)) {
        set_error("slamem_find_mems_device: null argument");
        return SLAMEM_ERR_ARG;
    }
    if (min_len < 1) {
        set_error("slamem_find_mems_device: minimum MEM length must be >= 1");
        return SLAMEM_ERR_ARG;
    }
    if (((uintptr_t)queries_dev & 15u) != 0) {
        set_error("slamem_find_mems_device: queries_dev must be 16-byte aligned");
        return SLAMEM_ERR_ARG;
    }
    strands = both_strands ? 2u : 1u;
    num_blocks = (uint64_t)num_queries * strands;
    w = layout_workspace(num_queries, strands, query_bytes, mems_capacity);
    if (w.max_items >= 0xFFF00000ull) {  // (the work cursor runs up to 4096 x 64 past the end of the list)
        set_error("slamem_find_mems_device: at most 2^32 - 2^20 work items per call");
        return SLAMEM_ERR_ARG;
    }
    if (workspace_bytes < w.bytes) {
        set_error("slamem_find_mems_device: workspace too small (%llu < %llu bytes)",
                  (unsigned long long)workspace_bytes, (unsigned long long)w.bytes);
        return SLAMEM_ERR_ARG;
    }
    want_stats = search_stats_wanted();
    for (int i = 0; i < 6; i++)
        if (!ev[i]) SLAMEM_HIP(hipEventCreate(&ev[i]));
    return SLAMEM_OK;
}

SearchJob::~SearchJob() {
    for (int i = 0; i < 6; i++)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
}

#define STEP(call, what) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_fail(e__, what, __FILE__, __LINE__); } while (0)

int SearchJob::tables(hipStream_t stream) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    char* ws = static_cast<char*>(workspace_dev);
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(ws + w.off_total);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(ws + w.off_cnt);
    uint32_t* d_first = reinterpret_cast<uint32_t*>(ws + w.off_first);
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(ws + w.off_counts);
    STEP(hipMemsetAsync(d_total, 0, 64 + SC_COUNT * 8, stream), "memset");
    if (want_stats) STEP(hipMemsetAsync(d_total + 8 + SC_T_FIRST, 0xFF, 16, stream), "memset");  // the two minima
    // ---- work items: one per strand, long records cut into slices ----------------------------------------------
    uint32_t* d_wps = reinterpret_cast<uint32_t*>(ws + w.off_wps);
    uint64_t* d_wscan = reinterpret_cast<uint64_t*>(ws + w.off_wscan);
    // reads (no record was cut into slices, none longer on average than the seed kernel's strands), -mem, a minimum length
    // that leaves at least four letters between two windows: K8s finds the MEMs by seed-and-compare and leaves to K8
    // only the strands it cannot decide (SLAMEM_SEED_SEARCH=0: the prefilter and the index walk for everything)
    // (read per call, not once per process: the tests run both paths in one process)
    const bool use_seed = [] { const char* v = getenv("SLAMEM_SEED_SEARCH"); return !(v && atoi(v) == 0); }();
    // (min_len >= k + 2: at least three letters between two windows -- 59 windows in a strand of 192 letters)
    const bool seed_ok = use_seed && match_type == 0 && idx->view.seed && min_len >= idx->view.seed_k + 2u &&
                         min_len < 0x8000u && query_bytes <= (uint64_t)num_queries * kSeedMaxLenLong;
    uint32_t slices = slices_hint;
    const bool ask = slices_hint == 0xFFFFFFFFu;
    // a batch on the seed path takes its items from the offsets (item_of): no counts, no scans, no tables -- when the caller
    // says that no record is longer than a slice; when it does not, one pass over the lengths says so
    bool counted = false, scanned = false;
    unsigned int* d_most = reinterpret_cast<unsigned int*>(d_total) + 2;  // a word of the zeroed scalar block
    speculated = false;
    if (seed_ok && ask && speculate) {
        // slamem_find_mems_device: take the seed path without asking the device for the longest record first (a kernel and a
        // round trip in front of every call); K8s says at the end whether some record was longer than a slice, and the call
        // then starts again the careful way (collect(): saw_long)
        slices = num_queries;
        speculated = true;
    } else if (!(seed_ok && !ask && slices == num_queries)) {
        hipLaunchKernelGGL(k_item_counts, dim3(grid_for((uint64_t)num_queries + 1)), dim3(256), 0, stream, offsets_dev,
                           num_queries, (match_type == 1 && mam_whole_strands()) ? 0u : kSliceLen, d_cnt, d_wps, d_most);
        STEP(hipGetLastError(), "k_item_counts");
        counted = true;
        if (slices == 0xFFFFFFFFu) {
            unsigned int most = 0;
            STEP(hipMemcpyAsync(&most, d_most, 4, hipMemcpyDeviceToHost, stream), "memcpy");
            STEP(hipStreamSynchronize(stream), "item count (sync)");
            if (most <= 1u) slices = num_queries;
        }
    }
    seed_path = seed_ok && slices == num_queries;
    if (counted && !seed_path) {
        STEP(exclusive_scan_u32(d_cnt, d_first, (uint64_t)num_queries + 1, reinterpret_cast<uint32_t*>(ws + w.off_scan32), stream), "scan");
        scanned = true;
        if (slices == 0xFFFFFFFFu) {
            STEP(hipMemcpyAsync(&slices, d_first + num_queries, 4, hipMemcpyDeviceToHost, stream), "memcpy");
            STEP(hipStreamSynchronize(stream), "item count (sync)");
        }
    }
    (void)scanned;
    nitems = (uint64_t)slices * strands;
    if (!seed_path) {   // packed strands: offsets of the strand blocks (a batch on the seed path: item_pk_of)
        size_t need3 = w.scan_bytes;
        STEP(scan_sum_exclusive_u32_u64(ws + w.off_scan, need3, d_wps, d_wscan, (uint64_t)num_queries, stream), "scan");
    }
    {   // -mam: reads go through K8 (kMam); batches with a record longer than a slice through k_find_mams_sliced
        static const bool env_v3 = [] { const char* v = getenv("SLAMEM_MAM_V3"); return !(v && atoi(v) == 0); }();
        mam_v3 = match_type == 1 && nitems == num_blocks && !mam_whole_strands() && env_v3;
    }
    if (nitems > w.max_items) { set_error("slamem_find_mems_device: query_bytes is smaller than the offsets say"); return SLAMEM_ERR_ARG; }
    STEP(hipMemsetAsync(d_counts + nitems, 0, 4, stream), "memset");
    memset(&A, 0, sizeof(A));
    A.ix = idx->view;
    A.qwords = static_cast<const uint64_t*>(queries_dev);
    A.offsets = offsets_dev;
    A.num_queries = num_queries;
    A.strands = strands;
    A.min_len = min_len;
    A.capacity = mems_capacity;
    A.total = d_total;
    A.raw_key = reinterpret_cast<RawKey*>(ws + w.off_rawkey);
    A.raw_mem = reinterpret_cast<slamem_mem*>(ws + w.off_rawmem);
    A.block_counts = d_counts;
    A.inline_rows = reinterpret_cast<RawRow*>(ws + w.off_inline);
    A.inline_stride = w.max_items;
    A.items = reinterpret_cast<const ItemDesc*>(ws + w.off_items);
    A.num_items = nitems;
    A.item_attempt = reinterpret_cast<uint8_t*>(ws + w.off_attempt);
    A.query_words = (query_bytes + 7) / 8;
    A.stats = d_total + 8;  // behind the 64 bytes of scalars
    if (nitems && (match_type != 1 || mam_v3)) {
        uint64_t* d_itempk = reinterpret_cast<uint64_t*>(ws + w.off_itempk);
        uint64_t* d_pq = reinterpret_cast<uint64_t*>(ws + w.off_pq);
        if (seed_path) {  // no item tables: K8, its packer and the list prefilter take a strand's descriptor from the offsets
            A.implicit_items = 1u;
            A.items = nullptr;
        } else {
            hipLaunchKernelGGL(k_item_fill, dim3(grid_for(num_queries)), dim3(256), 0, stream, offsets_dev, d_first, d_wscan,
                               num_queries, strands, reinterpret_cast<ItemDesc*>(ws + w.off_items), d_itempk,
                               nitems != num_blocks ? reinterpret_cast<uint32_t*>(ws + w.off_itemblock) : (uint32_t*)nullptr);
            STEP(hipGetLastError(), "k_item_fill");
        }
        STEP(hipMemsetAsync(d_pq, 0, 16, stream), "memset");  // the two leading zero words
        A.pq = d_pq;
        A.pq_out = d_pq;
        A.item_pk = seed_path ? nullptr : d_itempk;
        // direct extension of single-row matches: on when the index has the text-ordered sections and the class
        // threshold can discriminate (min_len >= 8); entry depth = where chance matches stop, log4(n) + 3
        static const int env_depth = [] { const char* v = getenv("SLAMEM_DIRECT_DEPTH"); return v ? atoi(v) : 0; }();
        int lg = 0;
        for (uint64_t v = idx->hdr.n; v > 1; v >>= 2) lg++;
        A.direct_min_depth = (idx->view.tgrp && depth_class((int)min_len) >= 1u) ? lg + 3 : -1;
        if (env_depth > 0 && A.direct_min_depth >= 0) A.direct_min_depth = env_depth;
        if (env_depth < 0) A.direct_min_depth = -1;
        static const bool env_jump = [] { const char* v = getenv("SLAMEM_KJUMP_USE"); return !(v && atoi(v) == 0); }();
        A.use_jump = env_jump ? 1u : 0u;
        {   // skipping the chance matches behind a disagreeing letter (K8 states SKV / SKQ / SKP): needs the occurrence
            // bitmap, min_len >= its k, at most 4 probes per letter (their words go into the record slots), and a class that can vouch for a depth of min_len - 1
            static const bool env_skip = [] { const char* v = getenv("SLAMEM_SKIP"); return v && atoi(v) != 0; }();  // off unless asked for
            const uint32_t kd = idx->view.kbits ? idx->view.kbits_k : 0u;
            A.skip_w = 0;
            if (env_skip && kd && min_len >= kd && min_len >= 9u && min_len < 0x10000u && A.direct_min_depth >= 0) {
                const uint32_t s1 = min_len - kd + 1u;
                const uint32_t nprobes = (kd - 1u + s1 - 1u) / s1 + 1u;  // starts p-(k-1), +s1, ..., p
                if (nprobes <= 4u && depth_class((int)min_len - 1) >= 1u) {
                    A.skip_w = min_len - 1u;
                    A.skip_s1 = s1;
                    A.pq2 = reinterpret_cast<const uint64_t*>(ws + w.off_pq2);
                    A.pq2_out = reinterpret_cast<uint64_t*>(ws + w.off_pq2);
                    A.item_flags = reinterpret_cast<uint8_t*>(ws + w.off_itemflags);
                }
            }
        }
    }
    return SLAMEM_OK;
}

int SearchJob::prep(hipStream_t stream) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    char* ws = static_cast<char*>(workspace_dev);
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(ws + w.off_total);
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(ws + w.off_counts);
    prefiltered = false; timed_k8 = false; launched = true; seeded = false; rawkey_marked = false;
    (void)hipEventRecord(ev[0], stream);
    if (nitems && (match_type != 1 || mam_v3)) {
        static const bool use_filter = [] { const char* v = getenv("SLAMEM_KFILTER"); return !(v && atoi(v) == 0); }();
        seeded = seed_path;  // (decided in tables(), which makes no item tables for such a batch)
        if (seeded) {
            uint8_t* d_alive = reinterpret_cast<uint8_t*>(ws + w.off_alive);
            // (K8's kChunk instantiation wants every place of the overflow list marked "unused" before the launch: K8s puts
            //  records there too, so the marks come first)
            if (mems_capacity && !A.skip_w && !want_stats && enum_chunks()) {
                STEP(hipMemsetAsync(ws + w.off_rawkey, 0xFF, mems_capacity * sizeof(RawKey), stream), "memset");
                rawkey_marked = true;
            }
            STEP(hipMemsetAsync(d_counts, 0, nitems * 4, stream), "memset");
            STEP(hipMemsetAsync(A.item_attempt, 0, nitems, stream), "memset");
            {   // (experiments) SLAMEM_SEED_STEP=1..6: the stride of the first round's windows (1: one round, every window)
                const char* v = getenv("SLAMEM_SEED_STEP");
                A.seed_step = v && atoi(v) >= 1 && atoi(v) <= 6 ? (uint32_t)atoi(v) : 0u;
            }
            // K8s writes K8's work list itself (no pass over 20 M flags for the handful it leaves)
            uint32_t* d_ids = reinterpret_cast<uint32_t*>(ws + w.off_workids);
            uint32_t* d_nwork = reinterpret_cast<uint32_t*>(d_total) + 8;  // a word of the zeroed scalar block
            A.seed_left_ids = d_ids;
            A.seed_left_count = d_nwork;
            A.seed_long_flag = reinterpret_cast<unsigned int*>(d_total) + 3;  // a word of the zeroed scalar block
            A.seed_wide = reinterpret_cast<unsigned int*>(d_total) + 4;  // four words of the zeroed scalar block
            // the form: by the batch's average read length -- and not narrower than the last batches against this index asked for
            seed_words_avg = query_bytes > (uint64_t)num_queries * kSeedMaxLenMid ? 6u : query_bytes > (uint64_t)num_queries * kSeedMaxLen ? 4u : 3u;
            const uint32_t hint = __atomic_load_n(&idx->seed_words_hint, __ATOMIC_RELAXED);
            seed_words = seed_words_avg > hint ? seed_words_avg : hint;
            const bool mid_reads = seed_words == 4u, long_reads = seed_words == 6u;
            const dim3 gs(grid_for((uint64_t)num_queries, 4 * (long_reads ? kSeedReadsLong : mid_reads ? kSeedReadsMid : kSeedReads)));
            // (reads of up to 192 letters: three plane words a strand; a batch whose reads average more: four, or six)
            if (long_reads) {
                if (want_stats) hipLaunchKernelGGL((k_seed_mems<true, 6, kSeedReadsLong>), gs, dim3(256), 0, stream, A, d_alive);
                else hipLaunchKernelGGL((k_seed_mems<false, 6, kSeedReadsLong>), gs, dim3(256), 0, stream, A, d_alive);
            } else if (mid_reads) {
                if (want_stats) hipLaunchKernelGGL((k_seed_mems<true, 4, kSeedReadsMid>), gs, dim3(256), 0, stream, A, d_alive);
                else hipLaunchKernelGGL((k_seed_mems<false, 4, kSeedReadsMid>), gs, dim3(256), 0, stream, A, d_alive);
            } else {
                if (want_stats) hipLaunchKernelGGL((k_seed_mems<true, 3, kSeedReads>), gs, dim3(256), 0, stream, A, d_alive);
                else hipLaunchKernelGGL((k_seed_mems<false, 3, kSeedReads>), gs, dim3(256), 0, stream, A, d_alive);
            }
            STEP(hipGetLastError(), "k_seed_mems");
            (void)hipEventRecord(ev[3], stream);
            prefiltered = true;
            A.item_alive = d_alive;
            A.work_ids = d_ids;
            A.work_count = d_nwork;
            if (use_filter && idx->view.kfilter && min_len >= idx->view.kfilter_k) {
                // the strands K8s left, through the presence filter (the wrong strand of a read that was left whole dies here:
                // alive[] = 0; K8 and its packer pass over those entries of the list)
                hipLaunchKernelGGL(k_prefilter_list, dim3(512), dim3(256), 0, stream, A, (const uint32_t*)d_ids, (const uint32_t*)d_nwork, d_alive);
                STEP(hipGetLastError(), "k_prefilter_list");
            }
        } else if (use_filter && idx->view.kfilter && min_len >= idx->view.kfilter_k) {
            uint8_t* d_alive = reinterpret_cast<uint8_t*>(ws + w.off_alive);
            if (want_stats) hipLaunchKernelGGL(k_prefilter<true>, dim3(grid_for(nitems)), dim3(256), 0, stream, A, d_alive);
            else hipLaunchKernelGGL(k_prefilter<false>, dim3(grid_for(nitems)), dim3(256), 0, stream, A, d_alive);
            STEP(hipGetLastError(), "k_prefilter");
            (void)hipEventRecord(ev[3], stream);
            prefiltered = true;
            A.item_alive = d_alive;
            // dead items emit nothing: their counts are zero; the survivors become a dense work list
            STEP(hipMemsetAsync(d_counts, 0, nitems * 4, stream), "memset");
            STEP(hipMemsetAsync(A.item_attempt, 0, nitems, stream), "memset");
            uint32_t* d_ids = reinterpret_cast<uint32_t*>(ws + w.off_workids);
            uint32_t* d_nwork = reinterpret_cast<uint32_t*>(d_total) + 8;  // a word of the zeroed scalar block
            size_t need2 = w.select_bytes;
            STEP(select_indices_u32(ws + w.off_select, need2, d_alive, d_ids, d_nwork, nitems, stream), "select");
            A.work_ids = d_ids;
            A.work_count = d_nwork;
        }
        if (A.item_flags) STEP(hipMemsetAsync(A.item_flags, 0, nitems, stream), "memset");
        if (nitems != num_blocks) {  // some record is longer than a slice: the slices' start states (k_slice_states)
            static const uint32_t env_warm = [] { const char* v = getenv("SLAMEM_SLICE_WARMUP"); return v && atoi(v) > 0 ? (uint32_t)atoi(v) : kWarmUp; }();
            SliceState* d_states = reinterpret_cast<SliceState*>(ws + w.off_slicestate);
            A.item_block = reinterpret_cast<const uint32_t*>(ws + w.off_itemblock);
            hipLaunchKernelGGL(k_slice_states, dim3(grid_for(nitems)), dim3(256), 0, stream, A, d_states, env_warm);
            STEP(hipGetLastError(), "k_slice_states");
            hipLaunchKernelGGL(k_slice_chain, dim3(grid_for(nitems)), dim3(256), 0, stream, A, d_states);
            STEP(hipGetLastError(), "k_slice_chain");
            A.slice_state = d_states;
        }
        {   // K7q: the strands K8 will scan, packed; then (only if some record was cut into slices) all of those slices
#ifndef SLAMEM_K7Q_LANES
#define SLAMEM_K7Q_LANES 8
#endif
#ifndef SLAMEM_K7Q_BLOCKS
#define SLAMEM_K7Q_BLOCKS 32768  // (8192: 1.01 ms on the benchmark batch, 32768-131072: 0.93-0.96, one block per 32 items: 1.10)
#endif
            uint64_t pb = (nitems * SLAMEM_K7Q_LANES + 255) / 256;
            hipLaunchKernelGGL(k_pack_queries<SLAMEM_K7Q_LANES>, dim3((unsigned)(pb < SLAMEM_K7Q_BLOCKS ? pb : SLAMEM_K7Q_BLOCKS)), dim3(256), 0, stream, A, false);
            if (nitems != num_blocks) {
                pb = (nitems * 16 + 255) / 256;
                hipLaunchKernelGGL(k_pack_queries<16>, dim3((unsigned)(pb < 8192 ? pb : 8192)), dim3(256), 0, stream, A, true);
            }
        }
        STEP(hipGetLastError(), "k_pack_queries");
    }
    (void)hipEventRecord(ev[5], stream);
    return SLAMEM_OK;
}

// More than 1/32 of the BWT rows share at least min_len letters with a neighbour in suffix order (ArenaHeader::lcp_ge, counted
// at build time): intervals of several rows >= min_len deep -- enumeration jobs -- are then a large part of the work.  SURVEY's
// repeat model (0.5 % of the text in copies) stays below (1 %), a 10^5-copy family lies above (10 %).  SLAMEM_ENUM_CHUNKS=0/1
// decides instead.
bool SearchJob::enum_chunks() const {
    static const int env = [] { const char* v = getenv("SLAMEM_ENUM_CHUNKS"); return v ? (atoi(v) != 0 ? 1 : 0) : -1; }();
    if (env >= 0) return env == 1;
    int b = 0;
    while (b < 9 && kLcpGe[b + 1] <= min_len) b++;
    return (uint64_t)idx->hdr.lcp_ge[b] * 32ull > (uint64_t)idx->hdr.n;
}
bool SearchJob::can_carry() const {
    return nitems != 0 && (match_type != 1 || mam_v3) && nitems == num_blocks && !A.skip_w && !want_stats;
}
bool SearchJob::can_carry_into(const SearchJob& next) const {
    return can_carry() && next.can_carry() && next.idx == idx && next.min_len == min_len && next.strands == strands &&
           next.mam_v3 == mam_v3 && next.A.direct_min_depth == A.direct_min_depth && next.A.use_jump == A.use_jump;
}

static void fill_prev_ctx(SearchArgs& A, const SearchJob& from) {
    A.prev.inline_rows = from.A.inline_rows; A.prev.inline_stride = from.A.inline_stride; A.prev.raw_key = from.A.raw_key; A.prev.raw_mem = from.A.raw_mem;
    A.prev.total = from.A.total; A.prev.capacity = from.A.capacity; A.prev.block_counts = from.A.block_counts;
    A.prev.item_attempt = from.A.item_attempt;
    char* pws = static_cast<char*>(from.workspace_dev);
    A.carry_in = reinterpret_cast<const CarryRec*>(pws + from.w.off_carry);
    A.carry_in_count = reinterpret_cast<const unsigned int*>(reinterpret_cast<unsigned long long*>(pws + from.w.off_total) + 7);
}

int SearchJob::flush(hipStream_t stream) {
    if (!carried_out) return SLAMEM_OK;
    SLAMEM_HIP(hipSetDevice(idx->device));
    SearchArgs F = A;  // no new work: an empty list, nothing passed on
    F.work_ids = nullptr; F.work_count = nullptr; F.num_items = 0;
    F.carry_out = nullptr; F.carry_out_count = nullptr;
    fill_prev_ctx(F, *this);
    // (the cursor word of this job's scalar block has run past the end of its list: the empty list is drained at once)
    static const uint64_t env_waves = [] { const char* v = getenv("SLAMEM_K8_WAVES"); return v ? (uint64_t)atoll(v) : 0ull; }();
    const dim3 grid8(grid_for((env_waves ? env_waves : kK8Waves) * 64));
    if (mam_v3) hipLaunchKernelGGL((k_find_mems_v3<false, false, false, true, true>), grid8, dim3(256), 0, stream, F);
    else hipLaunchKernelGGL((k_find_mems_v3<false, false, false, false, true>), grid8, dim3(256), 0, stream, F);
    STEP(hipGetLastError(), "k_find_mems_v3 (flush)");
    carried_out = false;
    return SLAMEM_OK;
}

int SearchJob::search_k8(hipStream_t stream, SearchJob* carry_from, bool carry_out) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    char* ws = static_cast<char*>(workspace_dev);
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(ws + w.off_total);
    launched = true;
    carried_out = false;
    chunked = false;
    deferred = false;
    if (carry_from && !(carry_from->carried_out && carry_from->can_carry_into(*this))) {
        set_error("internal: K8 asked to take in the lanes of a launch that cannot pass them on");
        return SLAMEM_ERR_ARG;
    }
    if (carry_out && !can_carry()) carry_out = false;
    if (nitems && (match_type != 1 || mam_v3)) {
        // persistent waves: as many as the chip holds (256 CUs x 16 waves), fewer for small batches
        uint64_t waves = (nitems + kFetch - 1) / kFetch;
        static const uint64_t env_waves = [] { const char* v = getenv("SLAMEM_K8_WAVES"); return v ? (uint64_t)atoll(v) : 0ull; }();
        const uint64_t cap_waves = k8_wave_cap ? k8_wave_cap : env_waves ? env_waves : kK8Waves;
        if (waves > cap_waves) waves = cap_waves;
        const bool carry = carry_from != nullptr || carry_out;
        if (carry) {  // the grid covers every lane that may come in, whatever the size of this batch (every launch of a stream: the same)
            waves = cap_waves;
            A.carry_in = nullptr; A.carry_in_count = nullptr;
            if (carry_from) fill_prev_ctx(A, *carry_from);
            A.carry_out = carry_out ? reinterpret_cast<CarryRec*>(ws + w.off_carry) : nullptr;
            A.carry_out_count = carry_out ? reinterpret_cast<unsigned int*>(d_total + 7) : nullptr;  // a word of the zeroed scalar block
        }
        A.work_cursor = reinterpret_cast<unsigned int*>(d_total + 5);  // a word of the zeroed scalar block
        (void)hipEventRecord(ev[4], stream);
        timed_k8 = true;
        const dim3 grid8(grid_for(waves * 64));
        const bool sliced = nitems != num_blocks;  // some record is longer than a slice
        if (carry) {
            if (mam_v3) hipLaunchKernelGGL((k_find_mems_v3<false, false, false, true, true>), grid8, dim3(256), 0, stream, A);
            else hipLaunchKernelGGL((k_find_mems_v3<false, false, false, false, true>), grid8, dim3(256), 0, stream, A);
            carried_out = carry_out;
            if (carry_from) carry_from->carried_out = false;  // its lanes are taken care of by this launch
        } else if (!A.skip_w && !want_stats && enum_chunks()) {
            chunked = true;
            // a repeat-rich text at this minimum length: the instantiation whose enumeration jobs take their places in the
            // overflow list chunk-wise (wave_emit_step); every place of the list starts as "unused"
            if (mems_capacity && !rawkey_marked) STEP(hipMemsetAsync(ws + w.off_rawkey, 0xFF, mems_capacity * sizeof(RawKey), stream), "memset");
            // reads, -mem: the jobs leave the strands' chains (kDefer; SLAMEM_ENUM_DEFER=0: in the waves, as for slices and -mam)
            const bool env_defer = [] { const char* v = getenv("SLAMEM_ENUM_DEFER"); return !(v && atoi(v) == 0); }();
            if (sliced) hipLaunchKernelGGL((k_find_mems_v3<false, false, true, false, false, true>), grid8, dim3(256), 0, stream, A);
            else if (mam_v3) hipLaunchKernelGGL((k_find_mems_v3<false, false, false, true, false, true>), grid8, dim3(256), 0, stream, A);
            else if (env_defer && mems_capacity) {
                deferred = true;
                A.defer.queue = reinterpret_cast<JobRec*>(ws + w.off_jobq);
                unsigned int* ds = reinterpret_cast<unsigned int*>(ws + w.off_defscal);
                A.defer.njobs = ds; A.defer.pool_next = ds + 1; A.defer.nlist = ds + 2;
                A.defer.pool = reinterpret_cast<uint32_t*>(ws + w.off_pool);
                A.defer.pool_cap = (uint32_t)w.pool_cap;
                A.defer.queue_cap = (uint32_t)(w.pool_cap + 64 * kK8Waves);
                {
                    static std::atomic<uint32_t> launch_stamp{0x5EED0000u};
                    A.defer.epoch = launch_stamp.fetch_add(1u) + 1u;
                }
                A.defer.raw_seg = reinterpret_cast<uint32_t*>(ws + w.off_rawseg);
                A.defer.list = reinterpret_cast<uint2*>(ws + w.off_deflist);
                A.defer.inline_valid = reinterpret_cast<uint8_t*>(ws + w.off_itemflags);
                STEP(hipMemsetAsync(ds, 0, 64, stream), "memset");
                STEP(hipMemsetAsync(A.defer.raw_seg, 0xFF, mems_capacity * 4, stream), "memset");
                STEP(hipMemsetAsync(A.defer.inline_valid, 0, nitems, stream), "memset");
                hipLaunchKernelGGL((k_find_mems_v3<false, false, false, false, false, true, true>), grid8, dim3(256), 0, stream, A);
                STEP(hipGetLastError(), "k_find_mems_v3 (kDefer)");
                hipLaunchKernelGGL(k_enum_jobs, dim3((unsigned)(kK8Waves / 4)), dim3(256), 0, stream, A);
                STEP(hipGetLastError(), "k_enum_jobs");
                hipLaunchKernelGGL(k_defer_prefix, dim3(256), dim3(256), 0, stream, A);
#ifdef SLAMEM_DIAG_TRIPS
                {
                    unsigned int h[16];
                    (void)hipMemcpyAsync(h, ds, 64, hipMemcpyDeviceToHost, stream);
                    (void)hipStreamSynchronize(stream);
                    fprintf(stderr, "[defer] jobs %u pool %u strands-with-jobs %u | lane trips: max %u, sum %u over %u strands; of deferring strands: sum %u over %u | wave trips: max %u sum %u | job steps per wave: max %u, sum %u, busy groups sum %u\n",
                            h[0], h[1], h[2], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13]);
                }
#endif
            }
            else hipLaunchKernelGGL((k_find_mems_v3<false, false, false, false, false, true>), grid8, dim3(256), 0, stream, A);
        } else if (A.skip_w) {  // (the skipping variant: one instantiation, with the slice logic)
            if (want_stats) hipLaunchKernelGGL((k_find_mems_v3<true, true, true, false, false>), grid8, dim3(256), 0, stream, A);
            else hipLaunchKernelGGL((k_find_mems_v3<false, true, true, false, false>), grid8, dim3(256), 0, stream, A);
        } else if (sliced) {
            if (want_stats) hipLaunchKernelGGL((k_find_mems_v3<true, false, true, false, false>), grid8, dim3(256), 0, stream, A);
            else hipLaunchKernelGGL((k_find_mems_v3<false, false, true, false, false>), grid8, dim3(256), 0, stream, A);
        } else if (mam_v3) {
            if (want_stats) hipLaunchKernelGGL((k_find_mems_v3<true, false, false, true, false>), grid8, dim3(256), 0, stream, A);
            else hipLaunchKernelGGL((k_find_mems_v3<false, false, false, true, false>), grid8, dim3(256), 0, stream, A);
        } else {
            if (want_stats) hipLaunchKernelGGL((k_find_mems_v3<true, false, false, false, false>), grid8, dim3(256), 0, stream, A);
            else hipLaunchKernelGGL((k_find_mems_v3<false, false, false, false, false>), grid8, dim3(256), 0, stream, A);
        }
        STEP(hipGetLastError(), "k_find_mems_v3");
    } else if (nitems) {  // -mam: K9 of the v3 path places the MAMs and resolves their rows
        STEP(hipMemsetAsync(A.item_attempt, 0, nitems, stream), "memset");
        static const uint32_t env_warm = [] { const char* v = getenv("SLAMEM_MAM_WARMUP"); return v && atoi(v) > 0 ? (uint32_t)atoi(v) : kWarmUp; }();
        static const bool env_trace = getenv("SLAMEM_MAM_TRACE") != nullptr;
        if (mam_whole_strands()) {  // the whole-strand scan (what the slices are checked against in the tests)
            hipLaunchKernelGGL(k_find_mams, dim3(grid_for(num_blocks)), dim3(256), 0, stream, A);
            STEP(hipGetLastError(), "k_find_mams");
        } else {
            MamPass P;
            uint32_t* d_block = reinterpret_cast<uint32_t*>(ws + w.off_itemblock);
            uint32_t* d_run = reinterpret_cast<uint32_t*>(ws + w.off_mamrun);
            unsigned int* d_nrun = reinterpret_cast<unsigned int*>(d_total + 6);  // a word of the zeroed scalar block
            hipLaunchKernelGGL(k_item_fill, dim3(grid_for(num_queries)), dim3(256), 0, stream, offsets_dev,
                               reinterpret_cast<const uint32_t*>(ws + w.off_first), (const uint64_t*)nullptr, num_queries, strands,
                               reinterpret_cast<ItemDesc*>(ws + w.off_items), (uint64_t*)nullptr, d_block);
            STEP(hipGetLastError(), "k_item_fill");
            P.guess = nullptr;
            P.alive = nullptr;
            {   // the presence filter of -mem serves -mam as well: a strand without any match of min_len letters has no MAM
                static const bool use_filter = [] { const char* v = getenv("SLAMEM_KFILTER"); return !(v && atoi(v) == 0); }();
                if (use_filter && idx->view.kfilter && min_len >= idx->view.kfilter_k) {
                    uint8_t* d_alive = reinterpret_cast<uint8_t*>(ws + w.off_alive);
                    hipLaunchKernelGGL(k_prefilter<false>, dim3(grid_for(nitems)), dim3(256), 0, stream, A, d_alive);
                    STEP(hipGetLastError(), "k_prefilter");
                    (void)hipEventRecord(ev[3], stream);
                    prefiltered = true;
                    P.alive = d_alive;
                }
            }
            if (nitems != num_blocks && idx->view.tgrp) {  // long records: the states where a warm-up meets no failed extension
                SliceState* d_states = reinterpret_cast<SliceState*>(ws + w.off_slicestate);
                A.item_block = d_block;
                hipLaunchKernelGGL(k_slice_states, dim3(grid_for(nitems)), dim3(256), 0, stream, A, d_states, env_warm);
                STEP(hipGetLastError(), "k_slice_states");
                hipLaunchKernelGGL(k_slice_chain, dim3(grid_for(nitems)), dim3(256), 0, stream, A, d_states);
                STEP(hipGetLastError(), "k_slice_chain");
                P.guess = d_states;
            }
            P.run_list = nullptr; P.run_count = nullptr;
            P.in_used = reinterpret_cast<MamState*>(ws + w.off_mamstate);
            P.out = P.in_used + w.max_bounds;
            P.item_block = d_block;
            P.slice_len = kSliceLen;
            P.warm_up = env_warm;
            hipLaunchKernelGGL(k_find_mams_sliced, dim3(grid_for(nitems)), dim3(256), 0, stream, A, P);
            STEP(hipGetLastError(), "k_find_mams_sliced");
            uint64_t reruns = 0, passes = 0;
            while (nitems != num_blocks) {  // some strand has more than one slice: verify, scan again what was guessed wrong
                unsigned int nrun = 0;
                STEP(hipMemsetAsync(d_nrun, 0, 4, stream), "memset");
                hipLaunchKernelGGL(k_mam_check, dim3(grid_for(nitems)), dim3(256), 0, stream, A, P, d_run, d_nrun);
                STEP(hipGetLastError(), "k_mam_check");
                STEP(hipMemcpyAsync(&nrun, d_nrun, 4, hipMemcpyDeviceToHost, stream), "memcpy");
                STEP(hipStreamSynchronize(stream), "k_mam_check (sync)");
                if (nrun == 0) break;
                reruns += nrun; passes++;
                MamPass R = P;
                R.run_list = d_run;
                R.run_count = d_nrun;
                hipLaunchKernelGGL(k_find_mams_sliced, dim3(grid_for(nrun)), dim3(256), 0, stream, A, R);
                STEP(hipGetLastError(), "k_find_mams_sliced (rerun)");
            }
            if (env_trace)
                fprintf(stderr, "[mam] %llu slices of %llu strands, warm-up %u: %llu scanned again in %llu passes\n", (unsigned long long)nitems,
                        (unsigned long long)num_blocks, env_warm, (unsigned long long)reruns, (unsigned long long)passes);
        }
    }
    (void)hipEventRecord(ev[1], stream);
    return SLAMEM_OK;
}

int SearchJob::place(hipStream_t stream) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    if (carried_out) { set_error("internal: K9 before the batch's last lanes are finished"); return SLAMEM_ERR_ARG; }
    char* ws = static_cast<char*>(workspace_dev);
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(ws + w.off_total);
    // ---- K9, right behind K8: per-item counts -> offsets, strand-block offsets, placement.  Nothing here needs a number from
    // the host: the grids cover the items, the list of overflow records is walked by a fixed grid up to the device-side count,
    // and every store is bounds-checked against the capacity (a batch that does not fit is reported by collect()).
    uint32_t* d_first = reinterpret_cast<uint32_t*>(ws + w.off_first);
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(ws + w.off_counts);
    // (no record cut into slices: item g is strand block g and the items' offsets ARE the blocks' -- scanned straight into the
    //  caller's array; otherwise a block starts where its first item does)
    const bool blocks_are_items = nitems == num_blocks;
    uint64_t* d_itemoff = blocks_are_items ? block_offsets_dev : reinterpret_cast<uint64_t*>(ws + w.off_itemoff);
    size_t need = w.scan_bytes;
    // (SLAMEM_K9_GROUPED=0: the plain scan and k_place_inline, what batches with sliced records take anyway)
    const bool grouped = blocks_are_items && nitems && mems_capacity && [] { const char* v = getenv("SLAMEM_K9_GROUPED"); return !(v && atoi(v) == 0); }();
    if (grouped) {
        const uint64_t ngroups = (nitems >> 6) + 1;  // (the group of entry nitems, the total, included)
        uint64_t* d_gsum = reinterpret_cast<uint64_t*>(ws + w.off_gsum);
        uint64_t* d_gpre = reinterpret_cast<uint64_t*>(ws + w.off_gpre);
        STEP(hipMemsetAsync(d_gsum + ngroups, 0, 8, stream), "memset");
        hipLaunchKernelGGL(k_group_totals, dim3(grid_for(ngroups * 64)), dim3(256), 0, stream, (const uint32_t*)d_counts, nitems, d_gsum);
        STEP(hipGetLastError(), "k_group_totals");
        STEP(scan_sum_exclusive_u64(ws + w.off_scan, need, d_gsum, d_gpre, ngroups, stream), "scan");
        hipLaunchKernelGGL(k_place_grouped, dim3(grid_for(nitems + 1)), dim3(256), 0, stream, (const uint32_t*)d_counts, nitems, (const uint64_t*)d_gpre,
                           block_offsets_dev, (const RawRow*)A.inline_rows, A.inline_stride, idx->view.sa, mems_capacity, mems_dev,
                           deferred ? (const uint8_t*)A.defer.inline_valid : (const uint8_t*)nullptr);
        STEP(hipGetLastError(), "k_place_grouped");
    } else {
        STEP(scan_sum_exclusive_u32_u64(ws + w.off_scan, need, d_counts, d_itemoff, nitems, stream), "scan");
    }
    if (!blocks_are_items) {
        hipLaunchKernelGGL(k_block_offsets, dim3(grid_for(num_blocks + 1)), dim3(256), 0, stream, d_first, d_itemoff,
                           num_queries, strands, nitems, block_offsets_dev);
        STEP(hipGetLastError(), "k_block_offsets");
    }
    if (nitems && mems_capacity) {
        if (!grouped) {
            hipLaunchKernelGGL(k_place_inline, dim3(grid_for(nitems)), dim3(256), 0, stream, A.inline_rows, A.inline_stride, d_counts,
                               d_itemoff, nitems, idx->view.sa, mems_capacity, mems_dev, deferred ? (const uint8_t*)A.defer.inline_valid : (const uint8_t*)nullptr);
            STEP(hipGetLastError(), "k_place_inline");
        }
        const uint64_t ob = (mems_capacity + 255) / 256;
        hipLaunchKernelGGL(k_place_overflow, dim3((unsigned)(ob < 2048 ? ob : 2048)), dim3(256), 0, stream, A.raw_key, A.raw_mem,
                           (const unsigned long long*)d_total, d_itemoff, A.item_attempt, idx->view.sa, mems_capacity, mems_dev,
                           deferred ? (const uint32_t*)A.defer.raw_seg : (const uint32_t*)nullptr, deferred ? (const uint32_t*)A.defer.pool : (const uint32_t*)nullptr);
        STEP(hipGetLastError(), "k_place_overflow");
    }
    (void)hipEventRecord(ev[2], stream);
    // [0] listed; u32 word 8: survivors of K8a, word 9: ordinal overflow flag; [8] all MEMs
    STEP(hipMemcpyAsync(h_scal, d_total, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream), "memcpy");
    STEP(hipMemcpyAsync(h_scal + 8, d_itemoff + nitems, 8, hipMemcpyDeviceToHost, stream), "memcpy");
    return SLAMEM_OK;
}

int SearchJob::collect() {
    Timings& tm = thread_timings();
    const unsigned long long* scal = h_scal;
    const unsigned long long listed = scal[0];  // records in the atomic list
    total = scal[8];                            // all MEMs
    saw_long = speculated && (uint32_t)(scal[1] >> 32) != 0u;
    if (saw_long) return SLAMEM_OK;  // (nothing of this run is used: find_mems_device starts again)
    if (seeded) {
        // more than an eighth of the (sampled) reads were left to the index walk only because the form was too narrow: the next batch
        // takes the form that holds them; a wider form than the average asks for that no read needed: one step back
        const uint64_t c4 = (uint32_t)scal[2], c6 = (uint32_t)(scal[2] >> 32), sampled = (uint32_t)(scal[3] >> 32);  // (a sample of the blocks counts)
        const bool needed = (uint32_t)scal[3] != 0u;
        uint32_t* hp = const_cast<uint32_t*>(&idx->seed_words_hint);
        const uint32_t hint = __atomic_load_n(hp, __ATOMIC_RELAXED);
        const uint32_t next = seed_words_next(hint, seed_words, seed_words_avg, c4, c6, sampled, needed);
        if (next != hint) __atomic_store_n(hp, next, __ATOMIC_RELAXED);
    }
    if ((uint32_t)(scal[4] >> 32) != 0u) {
        // (not SLAMEM_ERR_CAPACITY: callers answer that one by asking again with more room)
        set_error("slamem_find_mems_device: one work item emits 2^28 or more MEMs (a 4096-position slice against a highly "
                  "repetitive text with a small minimum length); raise min_len");
        return SLAMEM_ERR_ARG;
    }
    if (want_stats) {
        unsigned long long c[SC_COUNT];
        const uint32_t nwork = (uint32_t)scal[4];
        STEP(hipMemcpy(c, A.stats, sizeof(c), hipMemcpyDeviceToHost), "memcpy(stats)");
        slamem_search_stats& o = last_search_stats();
        memset(&o, 0, sizeof(o));
        o.fm_lines_top = c[SC_FM_TOP]; o.fm_lines_bottom = c[SC_FM_BOT];
        o.rec_lines_fail = c[SC_REC_FAIL_LINES]; o.rec_lines_pend = c[SC_REC_PEND_LINES]; o.rec_lines_flush = c[SC_REC_FLUSH_LINES];
        o.query_loads = c[SC_QUERY_LOADS]; o.lane_trips = c[SC_LANE_TRIPS]; o.wave_trips = c[SC_WAVE_TRIPS];
        o.positions = c[SC_POSITIONS]; o.enum_jobs = c[SC_ENUM_JOBS];
        o.dir_sa_lines = c[SC_DIR_SA]; o.dir_group_loads = c[SC_DIR_GROUPS]; o.dir_rec_lines = c[SC_DIR_RECS];
        o.dir_letters = c[SC_DIR_LETTERS]; o.jump_lines = c[SC_JUMP_LINES];
        o.skip_group_loads = c[SC_SKIP_GROUPS]; o.skip_probe_lines = c[SC_SKIP_PROBES]; o.skips = c[SC_SKIP_OK];
        o.skip_attempts = c[SC_SKIP_QLOADS];
        o.enum_row_steps = c[SC_ENUM_ROW_STEPS]; o.enum_levels = c[SC_ENUM_LEVELS];
        // 100 MHz clock -> microseconds
        last_search_clock()[0] = (c[SC_T_DRAIN] - c[SC_T_FIRST]) / 100.0;
        last_search_clock()[1] = (c[SC_T_LAST] - c[SC_T_DRAIN]) / 100.0;
        last_search_clock()[2] = c[SC_T_WAVE_SUM] / 100.0;
        o.enum_wave_us = c[SC_T_ENUM_SUM] / 100;
        for (int q = 0; q < 11; q++) { o.state_lane_trips[q] = c[SC_STATE_TRIPS + q]; o.state_wave_trips[q] = c[SC_STATE_WAVES + q]; }
        o.prefilter_probes = c[SC_PF_PROBES]; o.prefilter_query_loads = c[SC_PF_QUERY_LOADS]; o.prefilter_items = c[SC_PF_ITEMS];
        o.seed_windows = c[SC_SEED_WINDOWS]; o.seed_compares = c[SC_SEED_COMPARES]; o.seed_letter_masks = c[SC_SEED_NMASKS];
        o.seed_mems = c[SC_SEED_MEMS]; o.seed_strands_left = c[SC_SEED_LEFT]; o.seed_reads = c[SC_SEED_READS];
        o.seed_query_bytes = c[SC_SEED_QBYTES];
        for (int q = 0; q < 7; q++) o.seed_left_why[q] = c[SC_SEED_WHY + q];
        o.seed_once_reads = c[SC_SEED_ONCE];
        o.items = nitems;
        o.survivors = prefiltered ? nwork : nitems;
        o.mems = total;
        o.overflow_records = listed;
        o.valid = match_type != 1 ? 1 : 0;
    }
    // device times of the batch.  prep and search may have run on different streams (slamem_stream_*): the kernel time is the
    // sum of the two parts, not the span from the first event to the last (which would include the wait in between)
    float ms = 0, ms_prep = 0, ms_k8 = 0;
    if (hipEventElapsedTime(&ms, ev[0], ev[5]) == hipSuccess) ms_prep = ms;
    if (timed_k8 && hipEventElapsedTime(&ms, ev[4], ev[1]) == hipSuccess) {  // K8 alone
        ms_k8 = ms;
        tm.t.k8_ms = ms;
        tm.t.k8_ms_sum += ms;
    } else if (hipEventElapsedTime(&ms, ev[5], ev[1]) == hipSuccess) ms_k8 = ms;  // (-mam over slices: everything behind the preparation)
    tm.t.search_kernel_ms = ms_prep + ms_k8;
    tm.t.search_kernel_ms_sum += ms_prep + ms_k8;
    tm.t.search_launches++;
    if (prefiltered && hipEventElapsedTime(&ms, ev[0], ev[3]) == hipSuccess) {  // K8a, or K8s in its place
        if (seeded) { tm.t.seed_ms = ms; tm.t.seed_ms_sum += ms; tm.t.prefilter_ms = 0; }
        else { tm.t.prefilter_ms = ms; tm.t.prefilter_ms_sum += ms; tm.t.seed_ms = 0; }
    }
    if (hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess) tm.t.search_total_ms = ms_prep + ms_k8 + ms;
    if (total > mems_capacity || listed > mems_capacity) {
        // the atomic list also holds the records of abandoned slice attempts: ask for room for those too
        if (listed > total) total = listed;
        // (kChunk: the waves reserve places in chunks and may leave some unused; WHICH waves do depends on the run, the bound
        //  covers every run, so that a caller who asks again with this much room succeeds)
        if (chunked) total += kK8Waves * kOvfChunk;
        set_error("slamem_find_mems_device: %llu MEMs found, output capacity is %llu", (unsigned long long)total,
                  (unsigned long long)mems_capacity);
        return SLAMEM_ERR_CAPACITY;
    }
    return SLAMEM_OK;
}
#undef STEP

static int run_job(SearchJob& job, const slamem_index* idx, const void* queries_dev, const uint64_t* offsets_dev,
                   uint32_t num_queries, uint64_t query_bytes, uint32_t min_len, int both_strands, int match_type,
                   slamem_mem* mems_dev, uint64_t mems_capacity, uint64_t* block_offsets_dev, void* workspace_dev,
                   uint64_t workspace_bytes, hipStream_t stream) {
    int rc = job.init(idx, queries_dev, offsets_dev, num_queries, query_bytes, min_len, both_strands, match_type, mems_dev,
                      mems_capacity, block_offsets_dev, workspace_dev, workspace_bytes);
    if (rc == SLAMEM_OK) rc = job.tables(stream);
    if (rc == SLAMEM_OK) rc = job.prep(stream);
    if (rc == SLAMEM_OK) rc = job.search(stream);
    if (job.launched) {  // never return with kernels of this call in flight
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess && rc == SLAMEM_OK) rc = hip_fail(e, "MEM search (sync)", __FILE__, __LINE__);
    }
    if (rc == SLAMEM_OK) rc = job.collect();
    return rc;
}
int find_mems_device(const slamem_index* idx, const void* queries_dev, const uint64_t* offsets_dev,
                     uint32_t num_queries, uint64_t query_bytes, uint32_t min_len, int both_strands, int match_type,
                     slamem_mem* mems_dev, uint64_t mems_capacity, uint64_t* block_offsets_dev, void* workspace_dev,
                     uint64_t workspace_bytes, hipStream_t stream, uint64_t* total_out) {
    if (!total_out) { set_error("slamem_find_mems_device: null argument"); return SLAMEM_ERR_ARG; }
    SearchJob job;
    job.speculate = true;
    int rc = run_job(job, idx, queries_dev, offsets_dev, num_queries, query_bytes, min_len, both_strands, match_type, mems_dev,
                     mems_capacity, block_offsets_dev, workspace_dev, workspace_bytes, stream);
    if (rc == SLAMEM_OK && job.saw_long) {  // a record longer than a slice among the reads: once more, with the item tables
        job.speculate = false;
        rc = run_job(job, idx, queries_dev, offsets_dev, num_queries, query_bytes, min_len, both_strands, match_type, mems_dev,
                     mems_capacity, block_offsets_dev, workspace_dev, workspace_bytes, stream);
    }
    *total_out = job.total;
    return rc;
}

// The same steps for a caller that issues them apart (stream.hip): an opaque job per batch slot.
SearchJob* search_job_new() { return new (std::nothrow) SearchJob(); }
void search_job_delete(SearchJob* j) { delete j; }
int search_job_init(SearchJob* j, const slamem_index* idx, const void* queries_dev, const uint64_t* offsets_dev, uint32_t num_queries,
                    uint64_t query_bytes, uint32_t min_len, int both_strands, int match_type, slamem_mem* mems_dev,
                    uint64_t mems_capacity, uint64_t* block_offsets_dev, void* workspace_dev, uint64_t workspace_bytes,
                    unsigned long long* host_scalars) {
    j->h_scal = host_scalars ? host_scalars : j->scal_own;
    j->slices_hint = 0xFFFFFFFFu;
    j->k8_wave_cap = 0;
    return j->init(idx, queries_dev, offsets_dev, num_queries, query_bytes, min_len, both_strands, match_type, mems_dev,
                   mems_capacity, block_offsets_dev, workspace_dev, workspace_bytes);
}
void search_job_slices_hint(SearchJob* j, uint32_t slices) { j->slices_hint = slices; }
void search_job_k8_wave_cap(SearchJob* j, uint32_t waves) { j->k8_wave_cap = waves; }
int search_job_tables(SearchJob* j, hipStream_t stream) { return j->tables(stream); }
int search_job_prep(SearchJob* j, hipStream_t stream) { return j->prep(stream); }
int search_job_search(SearchJob* j, hipStream_t stream) { return j->search(stream); }
int search_job_can_carry_into(const SearchJob* j, const SearchJob* next) { return j->can_carry_into(*next) ? 1 : 0; }
int search_job_k8(SearchJob* j, hipStream_t stream, SearchJob* carry_from, int carry_out) { return j->search_k8(stream, carry_from, carry_out != 0); }
int search_job_carried_out(const SearchJob* j) { return j->carried_out ? 1 : 0; }
int search_job_flush(SearchJob* j, hipStream_t stream) { return j->flush(stream); }
int search_job_place(SearchJob* j, hipStream_t stream) { return j->place(stream); }
int search_job_collect(SearchJob* j, uint64_t* total_out) {
    int rc = j->collect();
    if (total_out) *total_out = j->total;
    return rc;
}

int follow_letter_batch(const slamem_index* idx, const char* letters, uint32_t* top, uint32_t* bot, uint32_t* size_out,
                        uint64_t count, hipStream_t stream) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    if (!count) return SLAMEM_OK;
    hipLaunchKernelGGL(k_follow_batch, dim3(grid_for(count)), dim3(256), 0, stream, idx->view, letters, top, bot, size_out, count);
    SLAMEM_HIP(hipGetLastError());
    return SLAMEM_OK;
}
int enclosing_interval_batch(const slamem_index* idx, uint32_t* top, uint32_t* bot, int32_t* depth_out, uint64_t count,
                             hipStream_t stream) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    if (!count) return SLAMEM_OK;
    hipLaunchKernelGGL(k_parent_batch, dim3(grid_for(count)), dim3(256), 0, stream, idx->view, top, bot, depth_out, count);
    SLAMEM_HIP(hipGetLastError());
    return SLAMEM_OK;
}
int position_in_text_batch(const slamem_index* idx, const uint32_t* rows, uint32_t* out, uint64_t count, hipStream_t stream) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    if (!count) return SLAMEM_OK;
    hipLaunchKernelGGL(k_locate_batch, dim3(grid_for(count)), dim3(256), 0, stream, idx->view, rows, out, count);
    SLAMEM_HIP(hipGetLastError());
    return SLAMEM_OK;
}
int char_at_bwt_pos_batch(const slamem_index* idx, const uint32_t* rows, char* out, uint64_t count, hipStream_t stream) {
    SLAMEM_HIP(hipSetDevice(idx->device));
    if (!count) return SLAMEM_OK;
    hipLaunchKernelGGL(k_bwtchar_batch, dim3(grid_for(count)), dim3(256), 0, stream, idx->view, rows, out, count);
    SLAMEM_HIP(hipGetLastError());
    return SLAMEM_OK;
}

}  // namespace slamem
