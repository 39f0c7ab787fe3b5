/* slamem_refapi.h -- SURVEY.md 8(b)(2), over the GPU engine: the functions of the reference's index boundary with the
 * reference's own names, argument meaning and global-state conventions (bwtindex.h:1-10, lcparray.h:1-4, sequence.h), each
 * one a call into libslamem_hip.so (include/slamem_hip.h).  libslamem_refapi.so lets a driver written against the reference's
 * headers -- its own GetMatches loop, slamem.c:37-218 -- link unchanged against an index that lives in HBM.
 *
 * This is a COMPATIBILITY layer for parity experiments and for porting a caller step by step, not the way to use a GPU: every
 * per-letter call is a kernel launch and a round trip across PCIe (tens of microseconds; the batched calls of slamem_hip.h and
 * slamem_find_mems_* are the product's boundary).  Conventions kept from the reference: one index per process in file-static
 * state, no error codes (a failure prints "> ERROR: ..." to stdout and exits -1, bwtindex.c:1441-1444), the text is borrowed for
 * the two build calls, the LCP byte array comes back through an out-parameter and belongs to the caller (slamem.c:75),
 * FMI_FollowLetter leaves the pair unchanged on 0.  The GPU is SLAMEM_DEVICE (default 0).
 */
#ifndef SLAMEM_REFAPI_H
#define SLAMEM_REFAPI_H

#ifdef __cplusplus
extern "C" {
#endif

/* bwtindex.h:7, call slamem.c:73 -- numTexts is 1 there; *lcpOut receives a malloc'ed array of n+1 bytes min(LCP,255) */
void FMI_BuildIndex(char **texts, unsigned int *sizes, unsigned int num_texts, unsigned char **lcp_out, char verbose);
/* lcparray.h:1, call slamem.c:74 -- the parent structure was built with the index; returns the number of LCP samples */
int BuildSampledLCPArray(char *text, unsigned int n, unsigned char *lcp, int min_lcp, int verbose);
unsigned int FMI_GetBWTSize(void);                                                          /* bwtindex.h:9, slamem.c:111 */
unsigned int FMI_GetTextSize(void);                                                         /* bwtindex.h:8 */
unsigned int FMI_FollowLetter(char letter, unsigned int *top, unsigned int *bottom); /* bwtindex.h:2, slamem.c:121 */
int GetEnclosingLCPInterval(unsigned int *top, unsigned int *bottom);                       /* lcparray.h:4, slamem.c:124,192 */
char FMI_GetCharAtBWTPos(unsigned int row);                                                 /* bwtindex.h:4, slamem.c:141,166 */
unsigned int FMI_PositionInText(unsigned int row);                                          /* bwtindex.h:1, slamem.c:142,167 */
void FMI_FreeIndex(void);                                                                   /* bwtindex.h:6, slamem.c:208 */
void FreeSampledSuffixArray(void);                                                          /* lcparray.h:2, slamem.c:209 */

#ifdef __cplusplus
}
#endif
#endif
