/* refshim.h -- SURVEY.md 8(b)(2): the eleven functions of the reference's index boundary, with the reference's own names,
 * argument meaning and global-state conventions, on top of the CPU restatement (oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: a driver written against the reference's bwtindex.h /
 * lcparray.h / sequence.h (slamem.c:73-77, 111-192, 208-209) links against liboracle_refshim.so unchanged, which is
 * useful for parity experiments only -- per-base calls are not a GPU boundary (include/slamem_hip.h is the product's).
 * Signatures restated from the reference's call sites (file:line per function below); no reference source is copied.
 * Conventions kept: one index per process in file-static state, no error codes (allocation failure prints to stdout
 * and exits -1 like bwtindex.c:1441-1444), the text is borrowed for the two build calls, the LCP byte array comes back
 * through an out-parameter and belongs to the caller (slamem.c:75), FMI_FollowLetter leaves the pair undefined on 0.
 */
#ifndef SLAMEM_ORACLE_REFSHIM_H
#define SLAMEM_ORACLE_REFSHIM_H

#ifdef __cplusplus
extern "C" {
#endif

/* bwtindex.h:7, call slamem.c:73 -- numTexts is always 1 there; lcpOut receives a malloc'ed array of n+1 bytes min(LCP,255) */
void FMI_BuildIndex(char **texts, unsigned int *sizes, unsigned int numTexts, unsigned char **lcpOut, char verbose);
/* lcparray.h:1, call slamem.c:74 -- returns the number of LCP samples (rows i with LCP[i] != LCP[i+1], lcparray.c:677-678) */
int BuildSampledLCPArray(char *text, unsigned int n, unsigned char *lcp, int minlcp, int verbose);
unsigned int FMI_GetBWTSize(void);                                                   /* bwtindex.c:263, slamem.c:111 */
unsigned int FMI_FollowLetter(char c, unsigned int *top, unsigned int *bottom);      /* bwtindex.c:359, slamem.c:121 */
int GetEnclosingLCPInterval(unsigned int *top, unsigned int *bottom);                /* lcparray.c:330, slamem.c:124,192 */
char FMI_GetCharAtBWTPos(unsigned int bwtpos);                                       /* bwtindex.c:304, slamem.c:141,166 */
unsigned int FMI_PositionInText(unsigned int bwtpos);                                /* bwtindex.c:402, slamem.c:142,167 */
void FMI_FreeIndex(void);                                                            /* slamem.c:208 */
void FreeSampledSuffixArray(void);                                                   /* slamem.c:209 */
void ReverseComplementSequence(char *text, int textsize);                            /* sequence.c:413, slamem.c:100 */
/* sequence.c:309, slamem.c:145: merged-text position -> record id, *pos becomes the offset inside the record.  The
 * reference reads its global sequence table; here the table of merged start positions is set first. */
void RefShim_SetMergedStarts(const unsigned int *starts, int num);
int GetSeqIdFromMergedSeqsPos(unsigned int *pos);

#ifdef __cplusplus
}
#endif
#endif
