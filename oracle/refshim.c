/* refshim.c -- see refshim.h.  TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"
#include "refshim.h"

static oracle_index *g_idx = NULL;         /* one index per process, like bwtindex.c:150-179 */
static const unsigned int *g_starts = NULL;
static int g_nstarts = 0;

void FMI_BuildIndex(char **texts, unsigned int *sizes, unsigned int numTexts, unsigned char **lcpOut, char verbose) {
    unsigned int n, i;
    const int32_t *lcp;
    (void)verbose;
    if (numTexts != 1 || !texts || !sizes) { printf("> ERROR: one text expected\n"); exit(-1); }
    if (g_idx) oracle_free(g_idx);
    n = sizes[0];
    g_idx = oracle_build(texts[0], n);
    if (!g_idx) { printf("> ERROR: Not enough memory to create index\n"); exit(-1); }
    if (lcpOut) { /* the byte array of bwtindex.c:1094-1304: min(LCP, 255) for rows 0..n (row 0: 0) */
        unsigned char *out = (unsigned char *)malloc((size_t)n + 1);
        if (!out) { printf("> ERROR: Not enough memory\n"); exit(-1); }
        lcp = oracle_lcp(g_idx);
        for (i = 0; i <= n; i++) out[i] = (unsigned char)(lcp[i] < 0 ? 0 : lcp[i] > 255 ? 255 : lcp[i]);
        *lcpOut = out;
    }
}

int BuildSampledLCPArray(char *text, unsigned int n, unsigned char *lcp, int minlcp, int verbose) {
    /* the restatement keeps exact LCP / PSV / NSV for every row (built by oracle_build); what is left of this call is its
       return value: the number of samples */
    const int32_t *l;
    unsigned int i;
    int samples = 0;
    (void)text; (void)lcp; (void)minlcp; (void)verbose;
    if (!g_idx || oracle_bwt_size(g_idx) != n + 1) { printf("> ERROR: index not built for this text\n"); exit(-1); }
    l = oracle_lcp(g_idx);
    for (i = 0; i <= n; i++) samples += l[i] != l[i + 1];
    return samples;
}

unsigned int FMI_GetBWTSize(void) { return oracle_bwt_size(g_idx); }
unsigned int FMI_FollowLetter(char c, unsigned int *top, unsigned int *bottom) {
    uint32_t t = *top, b = *bottom, r;
    if (b >= oracle_bwt_size(g_idx)) b = oracle_bwt_size(g_idx) - 1; /* the reference's initial bottom = n+1 (slamem.c:111) */
    r = oracle_follow_letter(g_idx, c, &t, &b);
    if (r) { *top = t; *bottom = b; }
    return r;
}
int GetEnclosingLCPInterval(unsigned int *top, unsigned int *bottom) {
    uint32_t t = *top, b = *bottom;
    int d;
    if (b >= oracle_bwt_size(g_idx)) b = oracle_bwt_size(g_idx) - 1;
    d = oracle_enclosing_interval(g_idx, &t, &b);
    *top = t; *bottom = b;
    return d;
}
char FMI_GetCharAtBWTPos(unsigned int bwtpos) { return oracle_char_at_bwt_pos(g_idx, bwtpos); }
unsigned int FMI_PositionInText(unsigned int bwtpos) { return oracle_position_in_text(g_idx, bwtpos, NULL); }
void FMI_FreeIndex(void) { if (g_idx) { oracle_free(g_idx); g_idx = NULL; } }
void FreeSampledSuffixArray(void) {}
void ReverseComplementSequence(char *text, int textsize) { oracle_reverse_complement(text, textsize); }
void RefShim_SetMergedStarts(const unsigned int *starts, int num) { g_starts = starts; g_nstarts = num; }
int GetSeqIdFromMergedSeqsPos(unsigned int *pos) { return oracle_seq_id_from_merged_pos(g_starts, g_nstarts, pos); }
