/* oracle.c -- CPU restatement of the slaMEM v0.8.2 MEM path.
 *
 * TEST INFRASTRUCTURE ONLY -- see oracle.h.  Parity status: PINNED against the
 * real reference's outputs (tests/golden/, made by oracle/_ref/slaMEM) and
 * against the brute-force MEM definition.
 *
 * This is a restatement, not a copy: it follows the reference's algorithm
 * (FM-index with rank samples every 32 rows and SA samples every 32 rows,
 * backward search, parent LCP-interval widening, enumeration of the interval
 * and of its ancestors with depth >= l, left-maximality filter on the BWT
 * character, locate by LF walk) with its own data structures:
 *   - suffix order comes from a from-scratch SA-IS (the arrays are uniquely
 *     defined by the text -- SURVEY.md Appendix A.2 -- so any correct
 *     construction is a valid restatement of bwtindex.c:706-1310);
 *   - LCP is exact (Kasai), as the reference's is after lcparray.c:650-662;
 *   - the parent interval uses plain PSV/NSV arrays, i.e. the four-line
 *     semantics of GetTrueEnclosingLCPInterval (lcparray.c:514-523) that the
 *     reference's sampled structure (lcparray.c:330-423) is checked against.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* alphabet: $=0 N=1 A=2 C=3 G=4 T=5   (bwtindex.c:39-41,183-196)       */
/* ------------------------------------------------------------------ */
static const char LETTERS[] = "$NACGT";

static inline unsigned letter_id(unsigned char c) {
    switch (c) {
    case 0: case '$': return 0;
    case 'A': case 'a': return 2;
    case 'C': case 'c': return 3;
    case 'G': case 'g': return 4;
    case 'T': case 't': return 5;
    default: return 1; /* every other byte is N, bwtindex.c:184 */
    }
}

/* ------------------------------------------------------------------ */
/* SA-IS (Nong, Zhang, Chan 2009), written from the published algorithm */
/* restates what GetLMSs/SortLMSs/InducedSort compute                   */
/* (bwtindex.c:706-1310): the suffix order of text+'$'.                 */
/* ------------------------------------------------------------------ */
#define CHR(i) (cs == 1 ? (int)((const unsigned char *)T)[i] : ((const int *)T)[i])
#define TGET(i) ((t[(i) >> 3] >> ((i) & 7)) & 1)
#define TSET(i, b) (t[(i) >> 3] = (unsigned char)((b) ? (t[(i) >> 3] | (1u << ((i) & 7))) : (t[(i) >> 3] & ~(1u << ((i) & 7)))))
#define IS_LMS(i) ((i) > 0 && TGET(i) && !TGET((i) - 1))

static void bucket_bounds(const void *T, int *bkt, int n, int K, int cs, int want_end) {
    int i, sum = 0;
    for (i = 0; i < K; i++) bkt[i] = 0;
    for (i = 0; i < n; i++) bkt[CHR(i)]++;
    for (i = 0; i < K; i++) {
        sum += bkt[i];
        bkt[i] = want_end ? sum : sum - bkt[i];
    }
}

static void induce_l(const unsigned char *t, int *SA, const void *T, int *bkt, int n, int K, int cs) {
    int i, j;
    bucket_bounds(T, bkt, n, K, cs, 0);
    for (i = 0; i < n; i++) {
        j = SA[i] - 1;
        if (j >= 0 && !TGET(j)) SA[bkt[CHR(j)]++] = j;
    }
}

static void induce_s(const unsigned char *t, int *SA, const void *T, int *bkt, int n, int K, int cs) {
    int i, j;
    bucket_bounds(T, bkt, n, K, cs, 1);
    for (i = n - 1; i >= 0; i--) {
        j = SA[i] - 1;
        if (j >= 0 && TGET(j)) SA[--bkt[CHR(j)]] = j;
    }
}

/* T[0..n-1], T[n-1] is the unique smallest sentinel; characters in [0,K). */
static int sais(const void *T, int *SA, int n, int K, int cs) {
    unsigned char *t;
    int *bkt, *SA1, *s1;
    int i, j, n1, name, prev;

    if (n == 1) { SA[0] = 0; return 0; }
    t = (unsigned char *)calloc((size_t)n / 8 + 1, 1);
    bkt = (int *)malloc(sizeof(int) * (size_t)K);
    if (!t || !bkt) { free(t); free(bkt); return -1; }

    TSET(n - 1, 1);
    TSET(n - 2, 0);
    for (i = n - 3; i >= 0; i--) {
        int a = CHR(i), b = CHR(i + 1);
        TSET(i, (a < b || (a == b && TGET(i + 1))) ? 1 : 0);
    }

    /* stage 1: sort all LMS substrings */
    bucket_bounds(T, bkt, n, K, cs, 1);
    for (i = 0; i < n; i++) SA[i] = -1;
    for (i = 1; i < n; i++)
        if (IS_LMS(i)) SA[--bkt[CHR(i)]] = i;
    induce_l(t, SA, T, bkt, n, K, cs);
    induce_s(t, SA, T, bkt, n, K, cs);

    /* compact the sorted LMS substrings into SA[0..n1) */
    n1 = 0;
    for (i = 0; i < n; i++)
        if (IS_LMS(SA[i])) SA[n1++] = SA[i];
    for (i = n1; i < n; i++) SA[i] = -1;

    /* name them */
    name = 0;
    prev = -1;
    for (i = 0; i < n1; i++) {
        int pos = SA[i], diff = 0, d;
        if (prev < 0) diff = 1;
        else {
            for (d = 0; d < n; d++) {
                if (CHR(pos + d) != CHR(prev + d) || TGET(pos + d) != TGET(prev + d)) { diff = 1; break; }
                if (d > 0 && (IS_LMS(pos + d) || IS_LMS(prev + d))) break;
            }
        }
        if (diff) { name++; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    for (i = n - 1, j = n - 1; i >= n1; i--)
        if (SA[i] >= 0) SA[j--] = SA[i];

    /* stage 2: order of the LMS suffixes */
    SA1 = SA;
    s1 = SA + n - n1;
    if (name < n1) {
        if (sais(s1, SA1, n1, name, (int)sizeof(int)) != 0) { free(t); free(bkt); return -1; }
    } else {
        for (i = 0; i < n1; i++) SA1[s1[i]] = i;
    }

    /* stage 3: induce the full order */
    bucket_bounds(T, bkt, n, K, cs, 1);
    for (i = 1, j = 0; i < n; i++)
        if (IS_LMS(i)) s1[j++] = i;
    for (i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
    for (i = n1; i < n; i++) SA[i] = -1;
    for (i = n1 - 1; i >= 0; i--) {
        j = SA[i];
        SA[i] = -1;
        SA[--bkt[CHR(j)]] = j;
    }
    induce_l(t, SA, T, bkt, n, K, cs);
    induce_s(t, SA, T, bkt, n, K, cs);
    free(t);
    free(bkt);
    return 0;
}

/* ------------------------------------------------------------------ */
/* index                                                                */
/* ------------------------------------------------------------------ */
/* IndexBlock, bwtindex.c:33-37: 32 BWT rows per block. */
typedef struct {
    uint32_t bits[3];  /* bit b of plane p = bit p of the letter id of row 32k+b        */
    uint32_t jumps[5]; /* for N,A,C,G,T: C[c]-1+occ(c, rows<32k)  bwtindex.c:1454,1481  */
    uint32_t sa;       /* SA[32k]                                  bwtindex.c:1516       */
} fm_block;

struct oracle_index {
    uint32_t n;         /* text length; rows = n+1 */
    uint32_t nblocks;
    fm_block *fm;
    int32_t *sa;        /* kept for structure-level parity only; locate does the LF walk */
    int32_t *lcp;       /* n+2 */
    int32_t *psv, *nsv; /* n+2 */
    uint8_t *bwt;       /* n+1 ids */
};

uint32_t oracle_bwt_size(const oracle_index *x) { return x->n + 1; }
const int32_t *oracle_sa(const oracle_index *x) { return x->sa; }
const int32_t *oracle_lcp(const oracle_index *x) { return x->lcp; }
const uint8_t *oracle_bwt(const oracle_index *x) { return x->bwt; }
const int32_t *oracle_psv(const oracle_index *x) { return x->psv; }
const int32_t *oracle_nsv(const oracle_index *x) { return x->nsv; }

void oracle_free(oracle_index *x) {
    if (!x) return;
    free(x->fm); free(x->sa); free(x->lcp); free(x->psv); free(x->nsv); free(x->bwt);
    free(x);
}

void oracle_free_mems(oracle_mem *p) { free(p); }

/* rows of the block whose letter id equals c, as a 32-bit mask (bwtindex.c:346-349) */
static inline uint32_t match_mask(const fm_block *b, unsigned c) {
    uint32_t m0 = (c & 1) ? b->bits[0] : ~b->bits[0];
    uint32_t m1 = (c & 2) ? b->bits[1] : ~b->bits[1];
    uint32_t m2 = (c & 4) ? b->bits[2] : ~b->bits[2];
    return m0 & m1 & m2;
}

static inline unsigned char_id_at(const oracle_index *x, uint32_t row) {
    const fm_block *b = &x->fm[row >> 5];
    unsigned o = row & 31;
    return ((b->bits[0] >> o) & 1) | (((b->bits[1] >> o) & 1) << 1) | (((b->bits[2] >> o) & 1) << 2);
}

/* FMI_LetterJump, bwtindex.c:340-356: C[c]-1+occ(c, rows<=row). */
static inline uint32_t letter_jump(const oracle_index *x, unsigned c, uint32_t row) {
    const fm_block *b = &x->fm[row >> 5];
    unsigned o = row & 31;
    uint32_t le = (o == 31) ? 0xFFFFFFFFu : ((1u << (o + 1)) - 1u);
    return b->jumps[c - 1] + (uint32_t)__builtin_popcount(match_mask(b, c) & le);
}

oracle_index *oracle_build(const char *text, uint32_t n) {
    oracle_index *x;
    unsigned char *codes;
    int32_t *rank;
    uint32_t i, rows = n + 1;
    uint32_t counts[6] = {0, 0, 0, 0, 0, 0}, C[6], run[6];

    if (n == 0 || n >= 0x7FFFFFF0u) return NULL;
    x = (oracle_index *)calloc(1, sizeof(*x));
    codes = (unsigned char *)malloc((size_t)rows);
    if (!x || !codes) { free(x); free(codes); return NULL; }
    x->n = n;
    for (i = 0; i < n; i++) {
        codes[i] = (unsigned char)letter_id((unsigned char)text[i]);
        if (codes[i] == 0) codes[i] = 1; /* NUL / '$' inside the text cannot happen after normalisation */
    }
    codes[n] = 0;

    x->sa = (int32_t *)malloc(sizeof(int32_t) * (size_t)rows);
    if (!x->sa || sais(codes, x->sa, (int)rows, 6, 1) != 0) { free(codes); oracle_free(x); return NULL; }

    /* BWT (bwtindex.c:1092,1204): BWT[i] = T[SA[i]-1], '$' for SA[i]==0 */
    x->bwt = (uint8_t *)malloc((size_t)rows);
    x->lcp = (int32_t *)malloc(sizeof(int32_t) * ((size_t)rows + 1));
    rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)rows);
    if (!x->bwt || !x->lcp || !rank) { free(codes); free(rank); oracle_free(x); return NULL; }
    for (i = 0; i < rows; i++) {
        x->bwt[i] = x->sa[i] ? codes[x->sa[i] - 1] : 0;
        rank[x->sa[i]] = (int32_t)i;
    }

    /* exact LCP (Kasai); the reference: induced min(LCP,255) bwtindex.c:1099,1221
       then exact values by text comparison lcparray.c:650-662; LCP[0]=LCP[n+1]=-1 lcparray.c:624,667 */
    {
        uint32_t h = 0;
        x->lcp[0] = -1;
        x->lcp[rows] = -1;
        for (i = 0; i < rows; i++) {
            int32_t r = rank[i];
            if (r == 0) { h = 0; continue; }
            {
                uint32_t j = (uint32_t)x->sa[r - 1];
                while (i + h < rows && j + h < rows && codes[i + h] == codes[j + h]) h++;
                x->lcp[r] = (int32_t)h;
                if (h) h--;
            }
        }
    }
    free(rank);

    /* PSV / NSV over lcp[0..n+1] (what the prefix links encode, lcparray.c:782-956) */
    x->psv = (int32_t *)malloc(sizeof(int32_t) * ((size_t)rows + 1));
    x->nsv = (int32_t *)malloc(sizeof(int32_t) * ((size_t)rows + 1));
    if (!x->psv || !x->nsv) { free(codes); oracle_free(x); return NULL; }
    {
        int64_t k, j;
        for (k = 0; k <= (int64_t)rows; k++) {
            j = k - 1;
            while (j >= 0 && x->lcp[j] >= x->lcp[k]) j = x->psv[j];
            x->psv[k] = (int32_t)j;
        }
        for (k = (int64_t)rows; k >= 0; k--) {
            j = k + 1;
            while (j <= (int64_t)rows && x->lcp[j] >= x->lcp[k]) j = x->nsv[j];
            x->nsv[k] = (int32_t)j; /* rows+1 = none */
        }
    }

    /* FM blocks (bwtindex.c:1439-1522) */
    for (i = 0; i < n; i++) counts[codes[i]]++;
    counts[0] = 1;
    C[0] = 0;
    for (i = 1; i < 6; i++) C[i] = C[i - 1] + counts[i - 1];
    x->nblocks = (n >> 5) + 1;
    x->fm = (fm_block *)calloc(x->nblocks, sizeof(fm_block));
    if (!x->fm) { free(codes); oracle_free(x); return NULL; }
    for (i = 0; i < 6; i++) run[i] = 0;
    for (i = 0; i < rows; i++) {
        fm_block *b = &x->fm[i >> 5];
        unsigned c = x->bwt[i], o = i & 31, k;
        if (o == 0) {
            for (k = 1; k < 6; k++) b->jumps[k - 1] = C[k] - 1 + run[k];
            b->sa = (uint32_t)x->sa[i];
        }
        b->bits[0] |= (uint32_t)(c & 1) << o;
        b->bits[1] |= (uint32_t)((c >> 1) & 1) << o;
        b->bits[2] |= (uint32_t)((c >> 2) & 1) << o;
        run[c]++;
    }
    free(codes);
    return x;
}

/* FMI_FollowLetter, bwtindex.c:359-400 */
uint32_t oracle_follow_letter(const oracle_index *x, char ch, uint32_t *top, uint32_t *bottom) {
    unsigned c = letter_id((unsigned char)ch);
    const fm_block *b;
    unsigned o;
    uint32_t lt, nt, nb;
    if (c == 0) return 0; /* the terminator is never searched, bwtindex.c:339 */
    b = &x->fm[*top >> 5];
    o = *top & 31;
    lt = (1u << o) - 1u; /* exclusive mask, bwtindex.c:375 */
    nt = b->jumps[c - 1] + (uint32_t)__builtin_popcount(match_mask(b, c) & lt) + 1; /* :379-385 */
    nb = letter_jump(x, c, *bottom);                                                /* :386-397 */
    if (nt > nb) return 0;                                                          /* :398     */
    *top = nt;
    *bottom = nb;
    return nb - nt + 1;
}

/* GetEnclosingLCPInterval, lcparray.c:330-423 == GetTrueEnclosingLCPInterval, lcparray.c:514-523 */
int oracle_enclosing_interval(const oracle_index *x, uint32_t *top, uint32_t *bottom) {
    uint32_t t = *top, b1 = *bottom + 1;
    int32_t lt = x->lcp[t], lb = x->lcp[b1];
    int32_t d = lt > lb ? lt : lb; /* destination depth = max(LCP[top], LCP[bottom+1]) :515-518 */
    if (d < 0) return -1;          /* root */
    if (lt == d) *top = (uint32_t)x->psv[t];         /* closest row above with a smaller LCP :519 */
    if (lb == d) *bottom = (uint32_t)x->nsv[b1] - 1; /* closest row below with a smaller LCP :520-521 */
    return d;
}

/* FMI_PositionInText, bwtindex.c:402-420 */
uint32_t oracle_position_in_text(const oracle_index *x, uint32_t row, uint32_t *lf_steps) {
    uint32_t add = 0;
    while (row & 31) {
        unsigned c = char_id_at(x, row);
        if (c == 0) { if (lf_steps) *lf_steps = add; return add; } /* :407-412 */
        row = letter_jump(x, c, row);
        add++;
    }
    if (lf_steps) *lf_steps = add;
    return x->fm[row >> 5].sa + add;
}

/* FMI_GetCharAtBWTPos, bwtindex.c:304-313 */
char oracle_char_at_bwt_pos(const oracle_index *x, uint32_t row) { return LETTERS[char_id_at(x, row)]; }

static int push_mem(oracle_mem **out, size_t *cap, size_t *have, uint32_t r, uint32_t q, uint32_t len) {
    if (*have == *cap) {
        size_t nc = *cap ? *cap * 2 : 1024;
        oracle_mem *p = (oracle_mem *)realloc(*out, nc * sizeof(oracle_mem));
        if (!p) return -1;
        *out = p;
        *cap = nc;
    }
    (*out)[*have].ref_pos = r;
    (*out)[*have].query_pos = q;
    (*out)[*have].length = len;
    (*have)++;
    return 0;
}

/* hot body of GetMatches, slamem.c:105-199 (MEM mode) */
size_t oracle_get_matches(const oracle_index *x, const char *q, uint32_t len, int min_len,
                          oracle_mem **out, size_t *cap, size_t have, oracle_counts *cnt) {
    return oracle_get_matches_mode(x, q, len, min_len, 0, out, cap, have, cnt);
}

/* match_type 0 = MEM (default), 1 = MAM (-mam, slamem.c:657): positions whose interval is not a single row are
 * skipped by the `continue` at slamem.c:131 -- which also skips the bookkeeping at :197-198, so the interval that a
 * later failed extension falls back to (:122-123) is a stale one.  Restated as is. */
size_t oracle_get_matches_mode(const oracle_index *x, const char *q, uint32_t len, int min_len, int match_type,
                               oracle_mem **out, size_t *cap, size_t have, oracle_counts *cnt) {
    oracle_counts local;
    uint32_t top = 0, bottom = x->n; /* slamem.c:110-111; n+1 there, n here */
    uint32_t prev_top = top, prev_bottom = bottom, saved_top, saved_bottom, row, j, n;
    int depth = 0, match;
    char c;
    memset(&local, 0, sizeof(local));
    for (j = len; j != 0;) { /* :114 */
        j--;
        local.n_querybase++;
        for (;;) { /* :121-128 */
            local.n_follow++;
            n = oracle_follow_letter(x, q[j], &top, &bottom);
            if (n) break;
            top = prev_top;
            bottom = prev_bottom;
            local.n_parent++;
            depth = oracle_enclosing_interval(x, &top, &bottom);
            if (depth == -1) break;
            prev_top = top;
            prev_bottom = bottom;
        }
        depth++; /* :129 */
        if (depth >= min_len) { /* :130 */
            if (match_type == 1 && n != 1) continue; /* :131 */
            saved_top = top;
            saved_bottom = bottom;
            prev_top = bottom + 1; /* :134 */
            prev_bottom = bottom;
            match = depth;
            c = j ? q[j - 1] : '\0'; /* :137-138 */
            if (c != '\0') c = LETTERS[letter_id((unsigned char)c)];
            while (match >= min_len) { /* :139 */
                for (row = top; row != prev_top; row++) { /* :140 */
                    local.n_bwtchar++;
                    if (oracle_char_at_bwt_pos(x, row) != c) {
                        uint32_t steps, r = oracle_position_in_text(x, row, &steps);
                        local.n_locate++;
                        local.n_lfstep += steps;
                        local.n_mem++;
                        if (push_mem(out, cap, &have, r, j, (uint32_t)match)) return (size_t)-1;
                    }
                }
                for (row = bottom; row != prev_bottom; row--) { /* :165 */
                    local.n_bwtchar++;
                    if (oracle_char_at_bwt_pos(x, row) != c) {
                        uint32_t steps, r = oracle_position_in_text(x, row, &steps);
                        local.n_locate++;
                        local.n_lfstep += steps;
                        local.n_mem++;
                        if (push_mem(out, cap, &have, r, j, (uint32_t)match)) return (size_t)-1;
                    }
                }
                prev_top = top; /* :190-192 */
                prev_bottom = bottom;
                local.n_parent++;
                match = oracle_enclosing_interval(x, &top, &bottom);
            }
            top = saved_top; /* :194-195 */
            bottom = saved_bottom;
        }
        prev_top = top; /* :197-198 */
        prev_bottom = bottom;
    }
    if (cnt) {
        cnt->n_follow += local.n_follow;
        cnt->n_bwtchar += local.n_bwtchar;
        cnt->n_lfstep += local.n_lfstep;
        cnt->n_locate += local.n_locate;
        cnt->n_parent += local.n_parent;
        cnt->n_querybase += local.n_querybase;
        cnt->n_mem += local.n_mem;
    }
    return have;
}

/* ReverseComplementSequence, sequence.c:413-430 */
void oracle_reverse_complement(char *s, int len) {
    int a = 0, b = len - 1;
    while (a <= b) {
        char ca = s[a], cb = s[b];
        ca = ca == 'A' ? 'T' : ca == 'C' ? 'G' : ca == 'G' ? 'C' : ca == 'T' ? 'A' : ca;
        cb = cb == 'A' ? 'T' : cb == 'C' ? 'G' : cb == 'G' ? 'C' : cb == 'T' ? 'A' : cb;
        s[a] = cb;
        s[b] = ca;
        a++;
        b--;
    }
}

/* GetSeqIdFromMergedSeqsPos, sequence.c:309-320 */
int oracle_seq_id_from_merged_pos(const uint32_t *starts, int num, uint32_t *pos) {
    int lo = 0, hi = num - 1;
    while (lo != hi) {
        int mid = (lo + hi + 1) / 2;
        if (*pos >= starts[mid]) lo = mid;
        else hi = mid - 1;
    }
    *pos -= starts[lo];
    return lo;
}

/* the definition, by diagonals. */
size_t oracle_brute_force_mems(const char *text, uint32_t n, const char *query, uint32_t m, int min_len,
                               oracle_mem **out, size_t *cap, size_t have) {
    int64_t d;
    for (d = -(int64_t)m + 1; d < (int64_t)n; d++) { /* d = r - q */
        int64_t q = d < 0 ? -d : 0, r = d < 0 ? 0 : d, run = 0;
        for (; q < (int64_t)m && r < (int64_t)n; q++, r++) {
            if (letter_id((unsigned char)text[r]) == letter_id((unsigned char)query[q])) run++;
            else {
                if (run >= min_len && run > 0)
                    if (push_mem(out, cap, &have, (uint32_t)(r - run), (uint32_t)(q - run), (uint32_t)run)) return (size_t)-1;
                run = 0;
            }
        }
        if (run >= min_len && run > 0)
            if (push_mem(out, cap, &have, (uint32_t)(r - run), (uint32_t)(q - run), (uint32_t)run)) return (size_t)-1;
    }
    return have;
}

/* the query loop of GetMatches, slamem.c:90-207 */
size_t oracle_match_batch(const oracle_index *x, const char *queries, const uint64_t *offsets,
                          uint32_t num, int min_len, int both, oracle_mem **out, size_t *cap,
                          uint64_t *block_counts, oracle_counts *cnt) {
    return oracle_match_batch_mode(x, queries, offsets, num, min_len, both, 0, out, cap, block_counts, cnt);
}

size_t oracle_match_batch_mode(const oracle_index *x, const char *queries, const uint64_t *offsets,
                               uint32_t num, int min_len, int both, int match_type, oracle_mem **out, size_t *cap,
                               uint64_t *block_counts, oracle_counts *cnt) {
    size_t have = 0, before;
    uint32_t i;
    char *buf = NULL;
    size_t bufcap = 0;
    for (i = 0; i < num; i++) {
        uint64_t len = offsets[i + 1] - offsets[i];
        before = have;
        have = oracle_get_matches_mode(x, queries + offsets[i], (uint32_t)len, min_len, match_type, out, cap, have, cnt);
        if (have == (size_t)-1) { free(buf); return have; }
        if (block_counts) block_counts[(size_t)i * (both ? 2 : 1)] = have - before;
        if (both) {
            if (len + 1 > bufcap) {
                char *nb = (char *)realloc(buf, len + 1);
                if (!nb) { free(buf); return (size_t)-1; }
                buf = nb;
                bufcap = len + 1;
            }
            memcpy(buf, queries + offsets[i], len);
            buf[len] = 0;
            oracle_reverse_complement(buf, (int)len);
            before = have;
            have = oracle_get_matches_mode(x, buf, (uint32_t)len, min_len, match_type, out, cap, have, cnt);
            if (have == (size_t)-1) { free(buf); return have; }
            if (block_counts) block_counts[(size_t)i * 2 + 1] = have - before;
        }
    }
    free(buf);
    return have;
}
