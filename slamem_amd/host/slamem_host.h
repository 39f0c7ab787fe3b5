/* slamem_host.h -- host side of the slaMEM-compatible front end (plain C, no GPU code).
 *
 * Written from scratch against the reference's observable behaviour:
 *   - FASTA loading / normalisation / merging        sequence.c:61-81, 89-270, 272-301, 309-320
 *   - command-line conventions                        slamem.c:528-663, tools.c:31-79
 *   - the *-mems.txt text format                      slamem.c:98, 102, 144-148
 * libslamem_host.so carries these for the CPU tests; slaMEM-hip links them with libslamem_hip.so.
 */
#ifndef SLAMEM_HOST_H
#define SLAMEM_HOST_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    char *name;    /* full header line after '>' (sequence.c:213-217) */
    uint32_t size; /* normalised characters of this record            */
} slh_record;

typedef struct {
    slh_record *recs;
    int num;                /* accepted records                                                      */
    char *chars;            /* merge: rec0 'N' rec1 'N' ... ; otherwise the records back to back     */
    uint64_t total;         /* bytes in chars (merge: what the reference stores as allSequences[0]->size) */
    uint64_t *offsets;      /* !merge: num+1 start offsets into chars                                */
    uint32_t *merged_start; /* merge: start of record k in chars (sequence.c:260-262)                */
    long file_bytes;
    void *name_arena;       /* owns the name strings */
} slh_seqset;

/* LoadSequencesFromFile (sequence.c:89-270).  merge=1 for the reference file (records joined by 'N',
 * name filter applies), merge=0 for query files.  first_number: running record number for the
 * "# NN [name]" log lines; log may be NULL.  log_limit: at most this many per-record lines are printed
 * (the reference prints all; 0 = unlimited).  Returns the number of accepted records, 0 on any failure
 * (message printed to log like the reference does). */
int slh_load_file(const char *path, int merge, int acgt_only, uint32_t min_len, const char *name_filter,
                  int first_number, long log_limit, slh_seqset *out, FILE *log);
void slh_free_seqset(slh_seqset *s);
/* A query file handed out in consecutive pieces of about piece_bytes of the file each (the same records as
 * slh_load_file(path, 0, ...) would give in one set): a front end searches the first reads while the rest is parsed. */
typedef struct slh_pieces slh_pieces;
slh_pieces *slh_pieces_open(const char *path, int acgt_only, uint32_t min_len, int first_number, long log_limit,
                            long piece_bytes, FILE *log);
int slh_pieces_next(slh_pieces *p, slh_seqset *out); /* records in the piece; 0 at the end */
void slh_pieces_close(slh_pieces *p);
/* on: the page-cache mapping of every parsed piece is dropped at once (for inputs whose footprint matters more than the
 * address-space work this causes beside the other threads; off by default) */
void slh_pieces_release_parsed(slh_pieces *p, int on);
/* malloc for buffers of tens of MB and more: 2 MB aligned, marked for transparent huge pages; release with free() */
void *slh_big_malloc(size_t bytes);
/* host threads used for loading / formatting: SLAMEM_THREADS or the online CPUs, at most 32 */
int slh_thread_count(void);

/* GetSeqIdFromMergedSeqsPos (sequence.c:309-320). */
int slh_seq_id_from_merged_pos(const uint32_t *starts, int num, uint32_t *pos);

/* Options as the reference parses them (slamem.c:571-663; tools.c:31-62). */
typedef struct {
    int usage;          /* argc < 3                                          */
    int hidden_sort;    /* -s given: slh_sort_mems_file                      */
    int hidden_clean;   /* -c given: slh_clean_fasta                         */
    int image_arg;      /* index of the -v value, or -1: slh_mem_map_image   */
    int no_ns;          /* -n                                                */
    int min_seq_len;    /* -m, 0 if absent                                   */
    char *ref_name;     /* -r string (malloc'ed) or NULL                     */
    int ref_name_given; /* -r present                                        */
    int ref_name_empty; /* -r present without a string                       */
    int match_type;     /* 0 MEM, 1 MAM (-ma...)                             */
    int both_strands;   /* -b                                                */
    int min_mem_len;    /* -l, default 20                                    */
    int out_arg;        /* index of the -o value, or -1                      */
    int num_files;
    int *file_args;     /* indices into argv of the FASTA files, in order    */
} slh_options;

int slh_parse_options(int argc, char **argv, slh_options *o);
void slh_free_options(slh_options *o);
/* ParseArgument (tools.c:31-62) */
int slh_parse_argument(int argc, char **argv, const char *optionchars, int parse);
/* AppendToBasename (tools.c:65-79): everything before the last '.' of the whole path + extra */
char *slh_append_to_basename(const char *filename, const char *extra);

/* One strand block of the output file (slamem.c:98/102 header + :144-148 lines).
 * mems: triples (ref_pos, query_pos, length), 0-based; printed 1-based.  When num_refs > 1 every line is
 * prefixed by " <record name>\t" and ref_pos is made relative to the record.  Appends to buf (realloc'ed). */
typedef struct {
    char *data;
    size_t len, cap;
} slh_buffer;
int slh_format_block(slh_buffer *buf, const char *query_name, int reverse, const uint32_t *mems, uint64_t count,
                     const slh_record *refs, const uint32_t *merged_start, int num_refs, uint64_t *sum_len_out);
void slh_buffer_free(slh_buffer *b);
/* make room for `bytes` more characters in one step (slh_format_block grows the buffer by doubling otherwise) */
int slh_buffer_reserve(slh_buffer *b, size_t bytes);

/* The hidden utilities of the reference's command line: "-s <mems_file>" (SortMEMsFile, slamem.c:244-352) and
 * "-c <fasta_file>" (CleanFasta, slamem.c:455-523).  Messages go to log; the return value is the process status. */
int slh_sort_mems_file(const char *path, FILE *log);
int slh_clean_fasta(const char *path, FILE *log);

/* "-v <mems_file>" (CreateMemMapImage, slamem.c:354-452; graphics.c, bitmap.c): the picture of the MEMs of every query against
 * the one reference, written as <mems_file without extension>.bmp, the same bytes as the reference's.  seqs: the reference
 * record first, then every query record in loading order.  Returns the process status (0, or -1 after an error message). */
int slh_mem_map_image(const char *mems_path, const slh_record *seqs, int num_seqs, int num_refs, FILE *log);

/* (tests) an arbitrary 8-bit picture, rows top first, through the tool's file writer (palette of the tool; run-length coded, or
 * plain when that would not be shorter); 1 on success */
int slh_write_bmp8(const char *path, int width, int height, const uint8_t *rows_top_first);

/* number of progress dots the reference prints for a strand of this length (slamem.c:94,116-120) */
int slh_progress_dots(uint32_t textsize);

#ifdef __cplusplus
}
#endif
#endif
