/* slamem_hip.h -- C ABI of the MI355X-native MEM engine (libslamem_hip.so).
 *
 * Drop-in boundary for slaMEM's MEM path.  The reference has no FFI layer: its
 * boundary is the pair of C headers bwtindex.h + lcparray.h as used by
 * GetMatches (slamem.c:37-218).  Those are per-base calls on file-static
 * globals, unusable across a PCIe/GPU boundary, so the ABI below is the coarse
 * (batched, handle-based, error-code) form of the same operations; every entry
 * point cites the reference interface it replaces.  Plain C types only: no
 * torch / HIP types in any signature (streams are passed as void*).
 *
 * Conventions
 *   - every function returns SLAMEM_OK (0) or a SLAMEM_ERR_* code; nothing
 *     calls exit() (the reference prints to stdout and exit(-1)s:
 *     slamem.c:58-61, bwtindex.c:1441-1444).  slamem_last_error_message()
 *     gives the text for the calling thread.
 *   - "_dev" pointers are device (HBM) pointers on the index's device.
 *   - rows are BWT rows 0..n (n = text length; row 0 is the '$' suffix);
 *     intervals are inclusive [top, bottom], as in the reference.
 *   - coordinates in slamem_mem are 0-based; the CLI prints them 1-based
 *     like slamem.c:148.
 */
#ifndef SLAMEM_HIP_H
#define SLAMEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMEM_ABI_VERSION 4

enum {
    SLAMEM_OK = 0,
    SLAMEM_ERR_ARG = 1,      /* bad argument                                     */
    SLAMEM_ERR_HIP = 2,      /* a HIP runtime call failed (message has the text) */
    SLAMEM_ERR_NOMEM = 3,    /* host or device allocation failed                 */
    SLAMEM_ERR_CAPACITY = 4, /* output buffer too small; *total_out has the need */
    SLAMEM_ERR_FORMAT = 5,   /* arena / file is not a slamem index               */
    SLAMEM_ERR_IO = 6,
    SLAMEM_ERR_NO_DEVICE = 7 /* no usable MI355X: there is NO CPU fallback       */
};

typedef struct slamem_index slamem_index; /* opaque; one per device, immutable after build */

/* One MEM: T[ref_pos .. ref_pos+length) == Q[query_pos .. query_pos+length),
 * not extendable on either side.  Replaces the fprintf at slamem.c:148,173. */
typedef struct {
    uint32_t ref_pos;
    uint32_t query_pos; /* in the strand that was scanned (reverse blocks: in the reverse complement, slamem.c:100-102) */
    uint32_t length;
} slamem_mem;

typedef struct {
    uint32_t text_length;  /* n                                           */
    uint32_t bwt_size;     /* n + 1            == FMI_GetBWTSize(), bwtindex.c:263 */
    uint32_t num_n_rows;   /* BWT rows holding 'N'                        */
    uint32_t dollar_row;   /* BWT row holding '$'                         */
    uint32_t max_lcp;      /* largest LCP value                           */
    uint32_t sort_rounds;  /* prefix-doubling rounds the build needed     */
    uint64_t arena_bytes;  /* bytes of HBM the index occupies             */
    int32_t device;
    int32_t owns_arena;
    uint32_t filter_k;     /* k of the k-mer presence filter (0 = the index has none); no reference counterpart */
    uint32_t layout;       /* SLAMEM_LAYOUT_FULL or SLAMEM_LAYOUT_COMPACT: what the build chose (ABI 2: reserved, 0)   */
    uint32_t seed_k;       /* letters of a seed of the seed-and-compare sections (0 = the index has none); ABI 4; no reference counterpart */
    uint32_t reserved1;
} slamem_index_info;

/* Index layouts (no reference counterpart: the reference has one layout of 3.3 B per letter made for CPU caches,
 * bwtindex.c:33-37 + lcparray.c:46-57; here HBM is spent to cut dependent random reads).
 *   FULL     every section: ~37.5 B per text letter + 16-32 B of presence filter + the seed-and-compare sections of the search of
 *            reads (ABI 4: seed table 16-32 B per letter, 8-16 B for texts of 2^28 letters and more -- there only when the build
 *            still fits the free HBM with it --, spill list, the text in 32-byte units): 8.3 GB at 100 Mbp, 188 GB at 3.1 Gbp
 *   COMPACT  no text-ordered sections (the search walks the index where it would have compared with the text) and a
 *            presence filter of half the size: ~21.5 B per letter + 8-16 B (3.3 GB at 100 Mbp, 81 GB at 3.1 Gbp); same
 *            results, slower search (DESIGN.md 2 has the measured cost)
 *   AUTO     FULL when its build peak fits the HBM that is free on the device, else COMPACT, else SLAMEM_ERR_NOMEM with
 *            the numbers in the message.  SLAMEM_INDEX_LAYOUT=full|compact in the environment decides instead;
 *            SLAMEM_HBM_BUDGET_GB caps what counts as free. */
enum { SLAMEM_LAYOUT_AUTO = 0, SLAMEM_LAYOUT_FULL = 1, SLAMEM_LAYOUT_COMPACT = 2 };

/* Per-phase device times of the last build / search on this thread's most recent
 * call, in milliseconds, measured with HIP events on the stream the kernels ran on. */
typedef struct {
    float build_total_ms;
    float build_pack_ms;      /* K1 text pack + histogram                         */
    float build_sort_ms;      /* K2 suffix sort (all radix passes, all rounds)    */
    float build_bwt_ms;       /* K3 BWT planes + rank samples                     */
    float build_lcp_ms;       /* K5 exact LCP                                     */
    float build_links_ms;     /* K7 PSV / NSV                                     */
    float search_kernel_ms;   /* K8a + K8: prefilter, work-list compaction, search */
    float search_total_ms;    /* K8 + scan + K9 scatter                           */
    uint64_t search_launches; /* number of K8 launches accumulated since reset    */
    double search_kernel_ms_sum;
    float prefilter_ms;       /* K8a (work-item fill + presence prefilter), part of search_kernel_ms */
    float k8_ms;              /* K8 k_find_mems_v3 alone, part of search_kernel_ms */
    double prefilter_ms_sum;
    double k8_ms_sum;
    float seed_ms;            /* K8s k_seed_mems (seed-and-compare for reads; runs in K8a's place), part of search_kernel_ms (ABI 4) */
    float reserved0;
    double seed_ms_sum;
} slamem_timings;

/* Load counters of ONE diagnostic search launch (slamem_search_stats_enable): how many loads of each kind the lanes of
 * K8a / K8 issued.  The diagnostic launch runs separate kernel instantiations that carry the counters; the normal
 * (timed) kernels carry none.  "lines" are 64-byte lines: FM blocks are one line each, a row-record pair is one or two.
 * bench.py prices roofline.traffic from these (cross-checked against the rocprofv3 PMC pass kept under profiles/). */
typedef struct {
    uint64_t fm_lines_top;          /* K8: FM block of `top` fetched (one per backward step unless still in registers) */
    uint64_t fm_lines_bottom;       /* K8: FM block of `bottom+1` when it is a different block                         */
    uint64_t rec_lines_fail;        /* K8: row-record lines after a failed extension (parent step)                     */
    uint64_t rec_lines_pend;        /* K8: row-record lines for the exact parent depth of a pending position           */
    uint64_t rec_lines_flush;       /* K8: row-record lines at strand ends                                             */
    uint64_t query_loads;           /* K8: 32-byte query windows loaded                                                */
    uint64_t lane_trips;            /* K8: loop trips summed over active lanes                                         */
    uint64_t wave_trips;            /* K8: loop trips summed over waves (lane_trips / (64 * wave_trips) = lane use)    */
    uint64_t positions;             /* K8: query positions consumed                                                    */
    uint64_t enum_jobs;             /* K8: wave-cooperative enumeration jobs                                           */
    uint64_t prefilter_probes;      /* K8a: presence-filter lines fetched (the tests behind a hit read the same line)  */
    uint64_t prefilter_query_loads; /* K8a: 16-byte query loads                                                        */
    uint64_t prefilter_items;       /* K8a: work items screened                                                        */
    uint64_t items;                 /* work items of the batch                                                         */
    uint64_t survivors;             /* work items K8 scanned                                                           */
    uint64_t mems;                  /* MEMs found                                                                      */
    uint64_t overflow_records;      /* MEMs that went through the atomic overflow list                                 */
    uint64_t valid;                 /* 1 when the counters describe a launch                                           */
    uint64_t dir_sa_lines;          /* K8 direct extension: suffix-array lines (one per run)                           */
    uint64_t dir_group_loads;       /* K8 direct extension: text groups (16 letters + classes, 16 B) with 16 B of query */
    uint64_t dir_rec_lines;         /* K8 direct extension: text-ordered records (one per run)                         */
    uint64_t dir_letters;           /* K8 direct extension: query positions consumed by comparing with the text        */
    uint64_t jump_lines;            /* K8: K-mer jump table entries read (one per scan start)                          */
    uint64_t skip_group_loads;      /* K8 skipping: text groups read to verify the diagonal behind a disagreeing letter */
    uint64_t skip_probe_lines;      /* K8 skipping: words of the k-mer occurrence bitmap read                          */
    uint64_t skip_attempts;         /* K8 skipping: diagonals verified (probes follow)                                 */
    uint64_t skips;                 /* K8 skipping: stretches skipped (min_len positions each)                         */
    uint64_t enum_row_steps;        /* K8: wave steps of the enumeration jobs (64 rows tested for left-maximality each)  */
    uint64_t enum_levels;           /* K8: ancestor intervals the enumeration jobs walked up to (one record round trip each) */
    uint64_t enum_wave_us;          /* K8: microseconds the waves spent inside enumeration jobs, summed over waves (compare k8 wave sum) */
    /* K8: loop trips per state of the lane's state machine (EXT, REC, FLUSH, DSA, DIR, DEND, JQ, JT, SKV, SKQ, SKP): summed
     * over lanes, and the number of WAVE trips in which at least one lane was in the state (what the wave pays for) */
    uint64_t state_lane_trips[11];
    uint64_t state_wave_trips[11];
    /* K8s (ABI 4): seed-and-compare for reads */
    uint64_t seed_windows;          /* K8s: seed-table lines fetched (one 64-byte line per window looked up; both strands share it)    */
    uint64_t seed_compares;         /* K8s: diagonals compared with the text (four 32-byte units of the text each: 128 bytes)          */
    uint64_t seed_letter_masks;     /* K8s: compares whose text units hold a letter that is not A,C,G,T                               */
    uint64_t seed_mems;             /* K8s: MEMs it reported                                                                          */
    uint64_t seed_strands_left;     /* K8s: strands left to the index walk (K8)                                                       */
    uint64_t seed_reads;            /* K8s: reads screened                                                                            */
    uint64_t seed_query_bytes;      /* K8s: bytes of the reads it packed                                                              */
    /* K8s: events counted on the way: [0] reads left to K8 before any lookup (longer than the kernel's strands, a letter that is
     * not A,C,G,T; also the reads of a wave whose compares did not fit), [1] windows whose bucket holds more k-mers than the
     * table keeps (28), [2] palindromic windows that hit (compared on both strands: not left), [3] trips whose compares did not
     * fit, [4] inconsistent hits (never), [5] MEMs beyond the wave's list, [6] MEMs whose tie with another of their strand (same
     * start, same length) the text behind them does not decide                                                                  */
    uint64_t seed_left_why[7];
    uint64_t seed_once_reads;       /* K8s: compares of the first round (they also look at the occurs-once plane of their units)    */
} slamem_search_stats;

/* ---- library ---------------------------------------------------------- */
int slamem_abi_version(void);
const char *slamem_strerror(int code);
const char *slamem_last_error_message(void);
int slamem_device_count(int *count_out);
/* Creates the HIP context of `device` (runtime start-up takes ~0.2 s): a front end calls this from a helper thread
 * while it parses its input, so that the index build does not pay for it.  No reference counterpart. */
int slamem_device_warmup(int device);
/* PCI address of `device` ("0000:c1:00.0"): a front end reads /sys/bus/pci/devices/<address>/local_cpulist to keep its host
 * threads on the GPU's NUMA node.  No reference counterpart. */
int slamem_device_pci_bus_id(int device, char *out, int out_bytes);
/* Free and total HBM of `device` in bytes (hipMemGetInfo): a front end that starts right behind another GPU job waits
 * for the memory its index needs (slamem_index_build_bytes) instead of failing.  No reference counterpart. */
int slamem_device_mem_info(int device, uint64_t *free_out, uint64_t *total_out);
int slamem_get_timings(slamem_timings *out);
int slamem_reset_timings(void);
/* on != 0: the NEXT slamem_find_mems_device calls of this thread run the diagnostic kernel instantiations (same results,
 * slower) and slamem_get_search_stats returns the counters of the last one.  No reference counterpart. */
int slamem_search_stats_enable(int on);
int slamem_get_search_stats(slamem_search_stats *out);
/* Timeline of the same diagnostic launch of K8 (device wall clock): microseconds from the first wave's start until the
 * work list was empty, microseconds from then until the last wave left (the tail), and the sum of all waves' run times. */
int slamem_get_search_clock(double *us_to_empty_list, double *us_tail, double *us_wave_sum);

/* ---- (a) index construction ------------------------------------------- */
/* Replaces FMI_BuildIndex(texts,sizes,1,&lcp,verbose) (bwtindex.h:7, call at
 * slamem.c:73) followed by BuildSampledLCPArray(text,n,lcp,minlcp,verbose)
 * (lcparray.h:1, call at slamem.c:74).  text: n bytes of A,C,G,T,N (any case;
 * every other byte counts as N, as letterIds does at bwtindex.c:183-196).
 * The text is borrowed for the call only (the reference frees it right after
 * the build too, slamem.c:75-77).  Everything runs on the device: suffix sort,
 * BWT bit-planes + rank samples, exact LCP, PSV/NSV links. */
int slamem_index_build(const char *text_host, uint32_t n, int device, slamem_index **out);
int slamem_index_build_device(const void *text_dev, uint32_t n, int device, void *stream, slamem_index **out);
/* The same with the layout stated (SLAMEM_LAYOUT_*; the two entry points above pass SLAMEM_LAYOUT_AUTO). */
int slamem_index_build_layout(const char *text_host, uint32_t n, int device, int layout, slamem_index **out);
int slamem_index_build_device_layout(const void *text_dev, uint32_t n, int device, void *stream, int layout,
                                     slamem_index **out);
/* HBM a text of n letters takes in `layout` (SLAMEM_LAYOUT_FULL / _COMPACT): the arena that stays, and the peak while
 * it is built (arena + suffix-sort scratch).  Host arithmetic only. */
int slamem_index_build_bytes(uint32_t n, int layout, uint64_t *arena_bytes_out, uint64_t *peak_bytes_out);
/* Replaces FMI_FreeIndex() + FreeSampledSuffixArray() (slamem.c:208-209). */
int slamem_index_free(slamem_index *idx);
int slamem_index_get_info(const slamem_index *idx, slamem_index_info *out);

/* The index is ONE contiguous HBM arena (4 KiB header + arrays), so that it can
 * be broadcast to peer GPUs with a single RCCL call and saved / loaded as one
 * blob (the reference only has a commented-out IDX0 sketch, bwtindex.c:480-579). */
int slamem_index_arena(const slamem_index *idx, void **arena_dev_out, uint64_t *bytes_out);
int slamem_index_export(const slamem_index *idx, void *dst_dev, uint64_t dst_bytes, void *stream);
/* Borrow an arena that a peer built (after ncclBroadcast / torch.distributed.broadcast).
 * The caller keeps the memory alive until slamem_index_free(). */
int slamem_index_attach(void *arena_dev, uint64_t bytes, int device, slamem_index **out);
/* Hand the attached arena over to the handle: slamem_index_free() will hipFree it (it must come from hipMalloc). */
int slamem_index_adopt_arena(slamem_index *idx);
int slamem_index_save(const slamem_index *idx, const char *path);
int slamem_index_load(const char *path, int device, slamem_index **out);
/* Host-only check of the first bytes (>= 256) of an arena or index file against the bytes available: magic, version, and
 * every section offset / size the kernels will index (aligned, ordered, inside the arena).  load and attach run the
 * same check; a corrupt or truncated index is SLAMEM_ERR_FORMAT, never a GPU fault. */
int slamem_index_validate_header(const void *header, uint64_t header_bytes, uint64_t available_bytes);

/* Structure-level parity (all uniquely defined by the text).
 * which: one of SLAMEM_ARRAY_*; host_dst must hold count elements of the stated type. */
enum {
    SLAMEM_ARRAY_SA = 0,  /* uint32[n+1]  suffix array                                  */
    SLAMEM_ARRAY_BWT = 1, /* uint8[n+1]   letter ids $=0 N=1 A=2 C=3 G=4 T=5            */
    SLAMEM_ARRAY_LCP = 2, /* int32[n+2]   exact LCP, [0] = [n+1] = -1                   */
    SLAMEM_ARRAY_PSV = 3, /* uint32[n+2]  nearest smaller value above (valid for 1..n)  */
    SLAMEM_ARRAY_NSV = 4  /* uint32[n+2]  nearest smaller value below (valid for 1..n)  */
};
int slamem_index_download(const slamem_index *idx, int which, void *host_dst, uint64_t count);

/* What the reference's sampled structure (SSILCP, lcparray.c:46-57) WOULD hold for this text, computed on the
 * device from the per-row records: the quantities BuildSampledLCPArray prints (lcparray.c:709-711, 999-1000).
 *   sample        = BWT row i with LCP[i] != LCP[i+1]                        (lcparray.c:677-678)
 *   oversized lcp = sample whose value is -1 or >= 255                       (lcparray.c:688-697)
 *   link          = top corner (LCP[i+1] > LCP[i]): PSV[i];  bottom corner: NSV[i+1]-1   (lcparray.c:827-912)
 *   oversized link= |distance| >= 128, plus the first and the last sample    (lcparray.c:755, 840, 890, 966-970)
 * Used by the front end to print the reference's statistics lines and by the tests as a check of rows a9-a11. */
typedef struct {
    uint64_t num_samples;
    uint64_t num_oversized_lcp;
    int64_t sum_lcp;            /* sum over rows 1..n+1 (the last one counts -1), lcparray.c:668 */
    uint32_t max_lcp;
    uint32_t pad;
    uint64_t num_oversized_links;
    uint64_t sum_link_distance;
    uint64_t max_link_distance;
} slamem_sslcp_stats;
int slamem_index_sampled_lcp_stats(const slamem_index *idx, slamem_sslcp_stats *out);

/* ---- fine-grained operations, batched (one lane per element) ------------ */
/* FMI_FollowLetter (bwtindex.h:8 / bwtindex.c:359): in-place on top/bottom; size_out[i] = new
 * interval size or 0 (then top/bottom are left unchanged). */
int slamem_follow_letter_batch(const slamem_index *idx, const char *letters_dev, uint32_t *top_dev,
                               uint32_t *bottom_dev, uint32_t *size_out_dev, uint64_t count, void *stream);
/* GetEnclosingLCPInterval (lcparray.h:2 / lcparray.c:330): in-place; depth_out[i] = parent depth, -1 at the root. */
int slamem_enclosing_interval_batch(const slamem_index *idx, uint32_t *top_dev, uint32_t *bottom_dev,
                                    int32_t *depth_out_dev, uint64_t count, void *stream);
/* FMI_PositionInText (bwtindex.h:9 / bwtindex.c:402). */
int slamem_position_in_text_batch(const slamem_index *idx, const uint32_t *rows_dev, uint32_t *pos_out_dev,
                                  uint64_t count, void *stream);
/* FMI_GetCharAtBWTPos (bwtindex.h:10 / bwtindex.c:304): one of "$NACGT". */
int slamem_char_at_bwt_pos_batch(const slamem_index *idx, const uint32_t *rows_dev, char *chars_out_dev,
                                 uint64_t count, void *stream);

/* ---- (b) MEM retrieval ---------------------------------------------------- */
/* Replaces the query loop of GetMatches (slamem.c:90-207) for a BATCH of query
 * records: per record the forward strand and, if both_strands, its reverse
 * complement (ReverseComplementSequence, sequence.c:413) are scanned right to
 * left with backward search + parent-interval widening, and every MEM of
 * length >= min_len is reported.
 *
 *   queries_dev       concatenated query characters (A,C,G,T,N; other bytes = N);
 *                     16-byte aligned and readable up to the next multiple of 16 bytes
 *   offsets_dev       uint64[num_queries+1]; record i is [offsets[i], offsets[i+1])
 *   query_bytes       offsets[num_queries] (total characters; sizes the work-item tables: records longer than 4096
 *                     characters are cut into slices that different lanes scan)
 *   strand blocks     block b = 2*i + strand when both_strands, else b = i
 *   mems_dev          out: slamem_mem[mems_capacity], grouped by block, inside a
 *                     block in the reference's emission order (slamem.c:139-193)
 *   block_offsets_dev out: uint64[num_blocks+1]; block b owns [off[b], off[b+1])
 *   workspace_dev     scratch of slamem_find_mems_workspace_bytes() bytes
 *   total_out         number of MEMs found (also when SLAMEM_ERR_CAPACITY is returned)
 *   limit             a strand (or a 4096-position slice of a long one) may emit fewer than 2^28 MEMs: beyond that the call
 *                     fails with SLAMEM_ERR_ARG and says so (never wrong output)
 *   passes            a batch of reads is answered in one pass with one host round trip (the total).  The call does not look
 *                     at the record lengths first: when the batch turns out to hold a record of more than 4096 letters (it
 *                     needs slices) the work is done again with the item tables -- two passes, same answer
 *
 * Synchronous with respect to the stream on return (it has to read the total). */
int slamem_find_mems_workspace_bytes(uint32_t num_queries, int both_strands, uint64_t query_bytes,
                                     uint64_t mems_capacity, uint64_t *bytes_out);
int slamem_find_mems_device(const slamem_index *idx, const void *queries_dev, const uint64_t *offsets_dev,
                            uint32_t num_queries, uint64_t query_bytes, uint32_t min_len, int both_strands,
                            slamem_mem *mems_dev, uint64_t mems_capacity, uint64_t *block_offsets_dev,
                            void *workspace_dev, uint64_t workspace_bytes, void *stream, uint64_t *total_out);

/* The same batch in MAM mode (option -mam: matchType 1, slamem.c:131,657): only positions whose match is a single
 * BWT row are reported.  The reference skips the other positions with a `continue` that also skips its interval
 * bookkeeping (slamem.c:197-198), which makes later fall-backs start from a stale interval; the
 * result is defined by that behaviour and is reproduced exactly.  Same arguments, layout and errors as
 * slamem_find_mems_device; strands are scanned whole (one lane per strand). */
int slamem_find_mams_device(const slamem_index *idx, const void *queries_dev, const uint64_t *offsets_dev,
                            uint32_t num_queries, uint64_t query_bytes, uint32_t min_len, int both_strands,
                            slamem_mem *mems_dev, uint64_t mems_capacity, uint64_t *block_offsets_dev,
                            void *workspace_dev, uint64_t workspace_bytes, void *stream, uint64_t *total_out);

/* Host-buffer convenience used by the C front end: uploads the batch, runs
 * slamem_find_mems_device (growing the output buffer if needed) and returns
 * malloc()ed arrays the caller frees with slamem_host_free(). */
int slamem_find_mems_host(const slamem_index *idx, const char *queries, const uint64_t *offsets,
                          uint32_t num_queries, uint32_t min_len, int both_strands,
                          slamem_mem **mems_out, uint64_t **block_offsets_out, uint64_t *total_out);
int slamem_find_mams_host(const slamem_index *idx, const char *queries, const uint64_t *offsets,
                          uint32_t num_queries, uint32_t min_len, int both_strands,
                          slamem_mem **mems_out, uint64_t **block_offsets_out, uint64_t *total_out);
void slamem_host_free(void *p);

/* ---- (b') MEM retrieval, host to host, pipelined -------------------------------
 * The query loop of GetMatches (slamem.c:90-207) for a front end whose reads live in HOST memory -- the boundary SURVEY.md
 * 8(d) defines the path's metric on.  A stream is a three-stage pipeline (upload, search, download: one host thread
 * and one HIP stream each) over `slots` (2..8) sets of device + pinned result buffers: while the search kernels of batch b
 * run, the copy engines upload batch b+1 and download the MEMs of batch b-1, so the sustained rate is the kernels' rate,
 * not kernels + PCIe.  Four or five slots keep all three stages busy beside the result the caller is working on.
 *
 *   slamem_stream_create   max_batch_chars / max_batch_queries: what to reserve per slot (a larger batch makes its slot
 *                          grow); match_type 0 = MEM, 1 = MAM (-mam)
 *   slamem_stream_submit   record i of the batch is queries[offsets[i] .. offsets[i+1]) -- offsets[0] need not be 0, so
 *                          a front end passes its whole character buffer and a window of its offsets array.  Returns at
 *                          once; the characters and offsets must stay unchanged until the batch has been collected.
 *                          Uploads run at full PCIe rate when `queries` AND `offsets` are pinned memory
 *                          (slamem_pinned_alloc): 8 bytes of offsets per record go up with every batch.
 *                          Never blocks: SLAMEM_ERR_ARG when every slot is in use.
 *   slamem_stream_next     waits for the OLDEST submitted batch (results come back in submission order) and lends its
 *                          result: mems grouped by strand block in the reference's emission order and block offsets,
 *                          laid out as slamem_find_mems_device does, in pinned host memory owned by the stream, valid
 *                          until the next slamem_stream_next / slamem_stream_destroy call.  A failed batch returns its
 *                          error code here.  So one thread keeps slots - 1 batches in flight beside the one it works on:
 *                              submit(0..slots-2);  for b: next(b); submit(b + slots - 1); use result b
 *   slamem_stream_destroy  waits for the batches in flight, then frees everything.
 * No CPU fallback: every batch is searched on the GPU. */
typedef struct slamem_stream slamem_stream;
int slamem_stream_create(const slamem_index *idx, int slots, uint64_t max_batch_chars, uint32_t max_batch_queries,
                         int both_strands, int match_type, slamem_stream **out);
int slamem_stream_submit(slamem_stream *s, const char *queries, const uint64_t *offsets, uint32_t num_queries,
                         uint32_t min_len);
/* The same for reads the caller holds PACKED (ABI 4; no reference counterpart: the reference reads letters, sequence.c:89-270).
 * Since the search takes ~9 ms for 10 M reads the link bounds this path (1.5 GB of letters: 26 ms); packed reads are a third.
 *   planes   16-byte units {p0, p1} (two 64-bit words): bit i of p0 / p1 = low / high bit of letter 64u+i of the record
 *            (A,C,G,T = 0..3; a letter that is none of them: 0 and its bit in `other`).  A record starts a new unit: record i
 *            of `len` letters takes (len + 63) / 64 units, the units of the batch's records follow each other.
 *   other    per unit: bit i = letter 64u+i is not one of A,C,G,T (it is searched as N, sequence.c:61-81); NULL: no such letter
 *   offsets  as for slamem_stream_submit (in letters; only differences are used)
 *   num_units  the units of the batch (slamem_pack_reads counts them), or 0: counted from the offsets (1 ms per million records)
 * slamem_pack_reads makes the two arrays from letters (host, `threads` threads; units_out: their number).  16-byte aligned
 * `planes`; pinned memory for full link rate. */
int slamem_stream_submit_packed(slamem_stream *s, const void *planes, const uint64_t *other, const uint64_t *offsets,
                                uint32_t num_queries, uint64_t num_units, uint32_t min_len);
int slamem_pack_reads(const char *queries, const uint64_t *offsets, uint32_t num_queries, void *planes_out, uint64_t *other_out,
                      uint64_t *units_out, int threads);
int slamem_stream_next(slamem_stream *s, const slamem_mem **mems_out, const uint64_t **block_offsets_out,
                       uint64_t *total_out, uint32_t *num_queries_out, slamem_timings *timings_out);
int slamem_stream_destroy(slamem_stream *s);
/* Page-locked host memory for a front end's read buffers (hipHostMalloc / hipHostFree). */
int slamem_pinned_alloc(void **out, uint64_t bytes);
int slamem_pinned_free(void *p);
/* hipMemcpy device -> host (for a front end that fills its pinned buffers from device memory). */
int slamem_copy_to_host(void *dst_host, const void *src_dev, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* SLAMEM_HIP_H */
