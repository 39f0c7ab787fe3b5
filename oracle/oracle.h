/* oracle.h -- CPU restatement of the slaMEM v0.8.2 MEM path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by
 * or executed from the product (slamem_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against outputs of the real reference (oracle/_ref/slaMEM, compiled from
 * /root/reference by oracle/Makefile) committed under tests/golden/ (22 cases,
 * five of them in -mam mode), and against the brute-force MEM definition
 * (oracle_brute_force_mems).  The full-size known answers of the real reference
 * (tests/golden/config3_known_answer.json) pin the GPU path directly.
 *
 * Every function cites the reference file:line whose behaviour it restates.
 */
#ifndef SLAMEM_ORACLE_H
#define SLAMEM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_index oracle_index;

typedef struct {
    uint32_t ref_pos;   /* 0-based position in the (merged) reference text */
    uint32_t query_pos; /* 0-based position in the query strand            */
    uint32_t length;
} oracle_mem;

/* Operation counters of the reference's hot loop, in the terms of SURVEY.md
 * 8(d): B = 72*follow + 36*bwtchar + 36*lfstep + 36*locate + 40*parent +
 * 1*querybase + 12*mem. */
typedef struct {
    uint64_t n_follow;    /* FMI_FollowLetter calls            bwtindex.c:359 */
    uint64_t n_bwtchar;   /* FMI_GetCharAtBWTPos calls         bwtindex.c:304 */
    uint64_t n_lfstep;    /* LF steps inside FMI_PositionInText bwtindex.c:405 */
    uint64_t n_locate;    /* FMI_PositionInText calls          bwtindex.c:402 */
    uint64_t n_parent;    /* GetEnclosingLCPInterval calls     lcparray.c:330 */
    uint64_t n_querybase; /* iterations of the scan loop       slamem.c:114   */
    uint64_t n_mem;       /* emitted MEMs                      slamem.c:148   */
} oracle_counts;

/* FMI_BuildIndex + BuildSampledLCPArray (bwtindex.c:1318, lcparray.c:545).
 * text: n bytes; A,C,G,T (any case) are themselves, every other byte is N.
 * Returns NULL on allocation failure or n == 0 or n >= 2^31-2. */
oracle_index *oracle_build(const char *text, uint32_t n);
/* FMI_FreeIndex + FreeSampledSuffixArray (bwtindex.c:220, lcparray.c:101). */
void oracle_free(oracle_index *idx);

/* FMI_GetBWTSize (bwtindex.c:263): n + 1. */
uint32_t oracle_bwt_size(const oracle_index *idx);

/* Uniquely defined arrays, for structure-level parity. */
const int32_t *oracle_sa(const oracle_index *idx);  /* n+1 rows; row 0 is '$'      */
const int32_t *oracle_lcp(const oracle_index *idx); /* n+2 values; [0]=[n+1]=-1    */
const uint8_t *oracle_bwt(const oracle_index *idx); /* n+1 ids: $=0 N=1 A=2 C=3 G=4 T=5 */
const int32_t *oracle_psv(const oracle_index *idx); /* n+2: nearest j<i with lcp[j]<lcp[i] */
const int32_t *oracle_nsv(const oracle_index *idx); /* n+2: nearest j>i with lcp[j]<lcp[i] */

/* FMI_FollowLetter (bwtindex.c:359-400): one backward-search step on an
 * inclusive row interval.  Returns the new interval size, or 0 (interval
 * unchanged here; the reference leaves garbage and the caller restores). */
uint32_t oracle_follow_letter(const oracle_index *idx, char c, uint32_t *top, uint32_t *bottom);
/* GetEnclosingLCPInterval (lcparray.c:330-423) with the semantics of
 * GetTrueEnclosingLCPInterval (lcparray.c:514-523).  Returns the parent's
 * string depth, -1 for the root (interval unchanged). */
int oracle_enclosing_interval(const oracle_index *idx, uint32_t *top, uint32_t *bottom);
/* FMI_PositionInText (bwtindex.c:402-420): SA[row] by LF-walking to a row that
 * is a multiple of 32.  *lf_steps (optional) receives the walk length. */
uint32_t oracle_position_in_text(const oracle_index *idx, uint32_t row, uint32_t *lf_steps);
/* FMI_GetCharAtBWTPos (bwtindex.c:304-313): one of "$NACGT". */
char oracle_char_at_bwt_pos(const oracle_index *idx, uint32_t row);

/* Hot body of GetMatches for ONE query strand (slamem.c:105-199), MEM mode.
 * Appends MEMs in the reference's emission order to *out (realloc'ed, *cap is
 * its capacity in elements, `have` the number already stored).  Returns the new
 * number of stored elements, or (size_t)-1 on allocation failure.  counts may be NULL. */
size_t oracle_get_matches(const oracle_index *idx, const char *query, uint32_t len, int min_len,
                          oracle_mem **out, size_t *cap, size_t have, oracle_counts *counts);
/* same with the match type of slamem.c:131,657: 0 = MEM, 1 = MAM (-mam), stale-interval quirk included */
size_t oracle_get_matches_mode(const oracle_index *idx, const char *query, uint32_t len, int min_len, int match_type,
                               oracle_mem **out, size_t *cap, size_t have, oracle_counts *counts);

/* ReverseComplementSequence (sequence.c:413-430), in place; N unchanged. */
void oracle_reverse_complement(char *text, int len);

/* GetSeqIdFromMergedSeqsPos (sequence.c:309-320). */
int oracle_seq_id_from_merged_pos(const uint32_t *starts, int num, uint32_t *pos);

/* The MEM definition itself: diagonal scan, O(n*m).
 * Independent of every index structure above.  Same output conventions as
 * oracle_get_matches; order is by diagonal. */
size_t oracle_brute_force_mems(const char *text, uint32_t n, const char *query, uint32_t m, int min_len,
                               oracle_mem **out, size_t *cap, size_t have);

/* Whole-batch convenience used by bench.py's cpu_baseline leg and the gloo
 * tests: queries are concatenated normalised bytes, offsets[num+1].  For each
 * query the forward strand and (if both_strands) the reverse complement are
 * scanned; block_counts[num*(1+both)] receives MEMs per strand block. */
size_t oracle_match_batch(const oracle_index *idx, const char *queries, const uint64_t *offsets,
                          uint32_t num, int min_len, int both_strands,
                          oracle_mem **out, size_t *cap, uint64_t *block_counts, oracle_counts *counts);
size_t oracle_match_batch_mode(const oracle_index *idx, const char *queries, const uint64_t *offsets,
                               uint32_t num, int min_len, int both_strands, int match_type,
                               oracle_mem **out, size_t *cap, uint64_t *block_counts, oracle_counts *counts);

void oracle_free_mems(oracle_mem *p);

#ifdef __cplusplus
}
#endif
#endif
