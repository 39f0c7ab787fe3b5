/* slamem_host.c -- FASTA loading, option parsing and MEM text formatting of the slaMEM-compatible
 * front end.  Plain C, no GPU code; see slamem_host.h for the reference lines each piece restates. */
#include "slamem_host.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <sys/mman.h>

/* ------------------------------------------------------------------------------------------------ */
/* FASTA                                                                                             */
/* ------------------------------------------------------------------------------------------------ */

/* InitCharsTable (sequence.c:61-81): A/C/G/T any case -> upper; other letters -> 'N' unless acgt_only;
 * '>' ends a record; everything else (digits, blanks, '*', '-') is dropped. */
static void init_table(char table[256], int allow_ns) {
    int i;
    memset(table, 0, 256);
    if (allow_ns)
        for (i = 'A'; i <= 'Z'; i++) { table[i] = 'N'; table[i + 32] = 'N'; }
    table['A'] = table['a'] = 'A';
    table['C'] = table['c'] = 'C';
    table['G'] = table['g'] = 'G';
    table['T'] = table['t'] = 'T';
}

typedef struct {
    const unsigned char *p, *end;
} reader;

/* record names are carved out of large chunks (10 M reads = 10 M names: one malloc each would dominate) */
typedef struct name_chunk {
    struct name_chunk *next;
    size_t used, cap;
    char data[1];
} name_chunk;

static char *name_alloc(void **arena, size_t len) {
    name_chunk *c = (name_chunk *)*arena;
    if (!c || c->used + len > c->cap) {
        size_t cap = len > (1u << 20) ? len : (1u << 20);
        name_chunk *n = (name_chunk *)malloc(sizeof(name_chunk) + cap);
        if (!n) return NULL;
        n->next = c;
        n->used = 0;
        n->cap = cap;
        *arena = n;
        c = n;
    }
    c->used += len;
    return c->data + c->used - len;
}

static int rd(reader *r) { return r->p < r->end ? (int)*r->p++ : EOF; }

/* Buffers of tens of MB and more (sequence characters, record tables): 2 MB aligned and marked for transparent huge
 * pages, which takes most of the page-fault and unmap time out of a 1.5 GB read set.  free() releases them. */
void *slh_big_malloc(size_t bytes) {
    void *p = NULL;
    if (bytes < ((size_t)32 << 20)) return malloc(bytes ? bytes : 1);
    bytes = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
    if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) return NULL;
#ifdef MADV_HUGEPAGE
    (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
    return p;
}

static int grow(char **buf, uint64_t *cap, uint64_t need) {
    if (need <= *cap) return 0;
    uint64_t nc = *cap ? *cap : (1u << 20);
    while (nc < need) nc += nc / 2 + (1u << 20);
    if (*buf == NULL) { /* the loaders size their character buffer once, from the file size */
        char *fb = (char *)slh_big_malloc(nc);
        if (!fb) return -1;
        *buf = fb;
        *cap = nc;
        return 0;
    }
    char *nb = (char *)realloc(*buf, nc);
    if (!nb) return -1;
    *buf = nb;
    *cap = nc;
    return 0;
}

void slh_free_seqset(slh_seqset *s) {
    int i;
    if (!s) return;
    (void)i;
    while (s->name_arena) {
        name_chunk *c = (name_chunk *)s->name_arena;
        s->name_arena = c->next;
        free(c);
    }
    free(s->recs);
    free(s->chars);
    free(s->offsets);
    free(s->merged_start);
    memset(s, 0, sizeof(*s));
}

/* A whole line at once (sequence.c:61-81 for a line that holds nothing but letters): the `len` bytes at p are written to dst as
 * the table would write them -- A,C,G,T of either case as upper case; every other letter as 'N' (allow_ns), or none may occur
 * (!allow_ns: such a line takes the byte loop, which drops them).  Returns 0 -- whatever it wrote so far is to be ignored -- when
 * the line holds a byte the rule does not cover (a digit, a blank, '>', '*', ...): the byte loop then does the line.  A
 * soft-masked assembly (half of it lower case) and IUPAC letters stay on this path; 16 bytes per step with SSE2. */
static int line_of_letters(unsigned char *dst, const unsigned char *p, size_t len, int allow_ns) {
    size_t i = 0;
    static int slow = -1; /* SLAMEM_LOADER_BYTEWISE=1: every line through the byte loop (what the tests compare this path with) */
    if (slow < 0) { const char *v = getenv("SLAMEM_LOADER_BYTEWISE"); slow = v && atoi(v) != 0; }
    if (slow) return 0;
#if defined(__SSE2__)
    const __m128i fold = _mm_set1_epi8((char)0xDF), a = _mm_set1_epi8('A'), c = _mm_set1_epi8('C'), g = _mm_set1_epi8('G'),
                  t = _mm_set1_epi8('T'), n = _mm_set1_epi8('N'), z = _mm_set1_epi8(25);
    for (; i + 16 <= len; i += 16) {
        const __m128i x = _mm_loadu_si128((const __m128i *)(p + i));
        const __m128i u = _mm_and_si128(x, fold);
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(u, a), _mm_cmpeq_epi8(u, c)),
                                        _mm_or_si128(_mm_cmpeq_epi8(u, g), _mm_cmpeq_epi8(u, t)));
        if (allow_ns) {
            const __m128i v = _mm_sub_epi8(u, a);                              /* 0..25 for a letter (as unsigned bytes) */
            if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(v, z), v)) != 0xFFFF) return 0;
            _mm_storeu_si128((__m128i *)(dst + i), _mm_or_si128(_mm_and_si128(ok, u), _mm_andnot_si128(ok, n)));
        } else {
            if (_mm_movemask_epi8(ok) != 0xFFFF) return 0;
            _mm_storeu_si128((__m128i *)(dst + i), u);
        }
    }
#endif
    for (; i < len; i++) {
        const unsigned char u = (unsigned char)(p[i] & 0xDF);
        const int ok = u == 'A' || u == 'C' || u == 'G' || u == 'T';
        if (!ok && !(allow_ns && u >= 'A' && u <= 'Z')) return 0;
        dst[i] = ok ? u : (unsigned char)'N';
    }
    return 1;
}

/* the records of one memory range that starts with '>' */
static int load_mem(const unsigned char *data, long fsize, int merge, int acgt_only, uint32_t min_len,
                    const char *name_filter, int first_number, long log_limit, slh_seqset *out, FILE *log) {
    char table[256];
    reader r;
    int c, k, numseqs = 0, reccap = 0, matchpos, desclen;
    uint64_t seqlen = 0, maxseqlen = 0, cap = 0, offcap = 0;
    uint32_t seqsize;
    long logged = 0;
    char *chars = NULL;

    memset(out, 0, sizeof(*out));
    out->file_bytes = fsize;
    r.p = data;
    r.end = data + fsize;
    c = rd(&r);
    if (c != '>') {
        if (log) fprintf(log, "> WARNING: Invalid FASTA file\n");
        return 0;
    }
    init_table(table, !acgt_only);
    table['>'] = (char)0xFF; /* record terminator for the fast path below */
    if (grow(&chars, &cap, (uint64_t)fsize + 16)) goto oom; /* characters (+ separators) never outnumber the file's bytes */
    for (;;) { /* all records of the file (sequence.c:128) */
        const unsigned char *name_start;
        int quiet;
        while (c != EOF && c != '>') c = rd(&r);
        if (c == EOF) break;
        quiet = !log || (log_limit > 0 && logged >= log_limit);
        if (!quiet) fprintf(log, "# %02d [", first_number + numseqs);
        name_start = r.p;
        matchpos = 0;
        desclen = 0;
        if (quiet && !name_filter) { /* nothing to print or to match: find the end of the line in one step */
            const unsigned char *nl = (const unsigned char *)memchr(r.p, '\n', (size_t)(r.end - r.p));
            const unsigned char *stop = nl ? nl : r.end;
            const unsigned char *cr = (const unsigned char *)memchr(r.p, '\r', (size_t)(stop - r.p));
            if (cr) stop = cr;
            desclen = (int)(stop - r.p);
            c = stop < r.end ? (int)*stop : EOF;
            r.p = stop < r.end ? stop + 1 : stop;
        } else
        while ((c = rd(&r)) != EOF && c != '\n' && c != '\r') {
            if (!quiet && desclen < 50) fputc(c, log);
            if (name_filter && name_filter[matchpos] != '\0') { /* sequence.c:137-140 */
                if (name_filter[matchpos] == (char)c) matchpos++;
                else matchpos = 0;
            }
            desclen++;
        }
        if (!quiet) {
            for (k = desclen; k < 50; k++) fputc(' ', log);
            fprintf(log, "] ");
        }
        if (name_filter && name_filter[matchpos] != '\0') {
            if (!quiet) { fprintf(log, "NAME DOES NOT MATCH\n"); logged++; }
            continue;
        }
        seqsize = 0;
        if (!merge) {
            maxseqlen = seqlen; /* every query record is its own sequence */
        } else if (numseqs == 0) {
            seqlen = 0; /* sequence.c:151-155 */
            maxseqlen = 0;
        }
        {
            uint64_t rec_start = seqlen;
            if (!merge) { /* query records: tight, branch-light copy loop (the 'N' separator logic is reference-only) */
                const unsigned char *p = r.p, *pe = r.end;
                unsigned char *dst = (unsigned char *)chars + seqlen;
                while (p < pe && *p != '>') { /* line by line; a '>' anywhere ends the record (sequence.c:157) */
                    const unsigned char *nl = (const unsigned char *)memchr(p, '\n', (size_t)(pe - p));
                    const unsigned char *end = nl ? nl : pe;
                    const unsigned char *le = (end > p && end[-1] == '\r') ? end - 1 : end; /* (a CR is skipped like the LF) */
                    if (line_of_letters(dst, p, (size_t)(le - p), !acgt_only)) {
                        dst += le - p;
                        p = end;
                    } else {
                        while (p < end) {
                            unsigned char t = (unsigned char)table[*p];
                            if (t == 0xFF) break;
                            p++;
                            *dst = t;
                            dst += (t != 0);
                        }
                        if (p < end) break; /* stopped at a '>' inside the line */
                    }
                    if (nl) p = nl + 1; /* (the table maps the newline to "skip") */
                }
                seqsize = (uint32_t)(dst - ((unsigned char *)chars + seqlen));
                seqlen += seqsize;
                c = p < pe ? '>' : EOF;
                r.p = p < pe ? p + 1 : p;
                maxseqlen = seqlen;
            } else
            while ((c = rd(&r)) != '>' && c != EOF) {
                char t = table[c];
                if (t && seqsize != 0 && seqlen + 1 < maxseqlen) {
                    /* inside a record (its first letter, with the separator 'N' of sequence.c:163-165, is behind us) and not at
                       one of the steps of maxseqlen: this letter, then whole lines while they hold nothing but letters */
                    const unsigned char *p = r.p, *pe = r.end;
                    chars[seqlen++] = t;
                    seqsize++;
                    for (;;) {
                        const unsigned char *nl, *end, *le;
                        if (p < pe && (*p == '\n' || *p == '\r')) { p++; continue; }
                        if (p >= pe || *p == '>') break;
                        nl = (const unsigned char *)memchr(p, '\n', (size_t)(pe - p));
                        end = nl ? nl : pe;
                        le = (end > p && end[-1] == '\r') ? end - 1 : end;
                        if ((uint64_t)(le - p) + seqlen + 1 >= maxseqlen || (uint64_t)(le - p) + seqlen >= 0xFFFFFFF0ull) break;
                        if (!line_of_letters((unsigned char *)chars + seqlen, p, (size_t)(le - p), !acgt_only)) break;
                        seqlen += (uint64_t)(le - p);
                        seqsize += (uint32_t)(le - p);
                        p = end;
                    }
                    r.p = p;
                    continue;
                }
                if (t) {
                    if (seqlen == maxseqlen) { /* sequence.c:160-166 */
                        maxseqlen += (1u << 20);
                        if (merge && numseqs != 0 && seqsize == 0) chars[seqlen++] = 'N';
                    }
                    chars[seqlen++] = t;
                    seqsize++;
                    if (seqlen >= 0xFFFFFFF0ull) {
                        if (log) fprintf(log, "\n> WARNING: Sequence lengths of more than %u bp are not supported\n", UINT_MAX);
                        goto fail;
                    }
                }
            }
            if (merge) {
                if (seqlen != 0) maxseqlen = seqlen; /* sequence.c:172-176 */
            }
            if (seqsize == 0) {
                if (!quiet) { fprintf(log, "EMPTY\n"); logged++; }
                continue;
            }
            if (min_len != 0 && seqsize < min_len) {
                if (!quiet) { fprintf(log, "(%u bp) TOO SHORT\n", seqsize); logged++; }
                if (merge) seqlen -= seqsize; /* sequence.c:200: the separator 'N' stays */
                else seqlen = rec_start;
                continue;
            }
            if (!quiet) fprintf(log, "(%u bp) ", seqsize);
            if (numseqs == reccap) {
                int nc = reccap ? reccap * 2 : 64;
                slh_record *nr = (slh_record *)realloc(out->recs, (size_t)nc * sizeof(slh_record));
                if (!nr) goto oom;
                out->recs = nr;
                reccap = nc;
            }
            out->recs[numseqs].name = name_alloc(&out->name_arena, (size_t)desclen + 1);
            if (!out->recs[numseqs].name) goto oom;
            memcpy(out->recs[numseqs].name, name_start, (size_t)desclen);
            out->recs[numseqs].name[desclen] = '\0';
            out->recs[numseqs].size = seqsize;
            if (!merge) {
                if ((uint64_t)numseqs + 2 > offcap) {
                    uint64_t nc = offcap ? offcap * 2 : 1024;
                    uint64_t *no = (uint64_t *)realloc(out->offsets, nc * sizeof(uint64_t));
                    if (!no) goto oom;
                    out->offsets = no;
                    offcap = nc;
                }
                out->offsets[numseqs] = rec_start;
                out->offsets[numseqs + 1] = seqlen;
            }
            numseqs++;
            out->num = numseqs;
            if (!quiet) { fprintf(log, "OK\n"); logged++; }
            else if (log && log_limit > 0 && logged == log_limit) {
                fprintf(log, "# ... (further records of this file are loaded without a line each)\n");
                logged++;
            }
        }
    }
    if (numseqs == 0) {
        free(chars);
        slh_free_seqset(out);
        return 0;
    }
    if (grow(&chars, &cap, seqlen + 16)) goto oom;
    memset(chars + seqlen, 0, 16);
    out->chars = chars;
    out->total = seqlen;
    if (merge) { /* sequence.c:258-265 */
        out->merged_start = (uint32_t *)malloc((size_t)numseqs * sizeof(uint32_t));
        if (!out->merged_start) { chars = NULL; goto oom; }
        out->merged_start[0] = 0;
        for (k = 1; k < numseqs; k++) out->merged_start[k] = out->merged_start[k - 1] + out->recs[k - 1].size + 1;
    } else if (!out->offsets) {
        goto oom;
    }
    return numseqs;
oom:
    if (log) fprintf(log, "\n> ERROR: Out of memory while loading sequences\n");
fail:
    if (chars != out->chars) free(chars);
    slh_free_seqset(out);
    return 0;
}


/* ---- parallel loading of large query files ------------------------------------------------------- */
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

typedef struct {
    const unsigned char *data;
    long size;
    int acgt_only;
    uint32_t min_len;
    int first_number;
    long log_limit;
    FILE *log;
    slh_seqset set;
    int n;
    slh_seqset *dst;   /* second phase: this piece is copied to characters dst_cpos.. / records dst_rpos.. of dst */
    uint64_t dst_cpos;
    int dst_rpos;
} load_job;

static void *load_job_run(void *arg) {
    load_job *j = (load_job *)arg;
    j->n = j->size > 0 ? load_mem(j->data, j->size, 0, j->acgt_only, j->min_len, NULL, j->first_number, j->log_limit, &j->set, j->log) : 0;
    return NULL;
}

void slh_free_seqset(slh_seqset *s);

static void *concat_job_run(void *arg) {
    load_job *j = (load_job *)arg;
    const slh_seqset *s = &j->set;
    int i;
    if (j->n == 0) return NULL;
    memcpy(j->dst->chars + j->dst_cpos, s->chars, s->total);
    memcpy(j->dst->recs + j->dst_rpos, s->recs, (size_t)s->num * sizeof(slh_record));
    for (i = 0; i < s->num; i++) j->dst->offsets[j->dst_rpos + i] = s->offsets[i] + j->dst_cpos;
    slh_free_seqset(&j->set); /* its names were handed over before; unmapping 100 MB pieces is worth doing in parallel too */
    return NULL;
}

int slh_thread_count(void) {
    const char *e = getenv("SLAMEM_THREADS");
    long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    if (n > 32) n = 32;
    return (int)n;
}

/* Query files above 64 MB are cut at "newline + '>'" positions (always a true record start: header lines are single
 * lines, and in sequence context any '>' starts a record, sequence.c:157) and parsed by several threads; the
 * pieces are concatenated in order, so the result equals the sequential parse.  The per-record log lines come from
 * the first piece only (they are limited to the first log_limit records anyway). */
static int load_parallel(const unsigned char *data, long fsize, int acgt_only, uint32_t min_len, int first_number,
                         long log_limit, slh_seqset *out, FILE *log, int threads) {
    load_job *jobs = (load_job *)calloc((size_t)threads, sizeof(load_job));
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    long *cut = (long *)calloc((size_t)threads + 1, sizeof(long));
    int t, total = 0, ok = 1;
    uint64_t chars_total = 0;
    if (!jobs || !tid || !cut) { free(jobs); free(tid); free(cut); return -1; }
    cut[0] = 0;
    cut[threads] = fsize;
    for (t = 1; t < threads; t++) {
        long p = fsize / threads * t;
        if (p < cut[t - 1]) p = cut[t - 1];
        while (p < fsize && !(data[p] == '>' && p > 0 && (data[p - 1] == '\n' || data[p - 1] == '\r'))) p++;
        cut[t] = p;
    }
    for (t = 0; t < threads; t++) {
        jobs[t].data = data + cut[t];
        jobs[t].size = cut[t + 1] - cut[t];
        jobs[t].acgt_only = acgt_only;
        jobs[t].min_len = min_len;
        jobs[t].first_number = first_number;
        jobs[t].log_limit = log_limit;
        jobs[t].log = t == 0 ? log : NULL;
        if (pthread_create(&tid[t], NULL, load_job_run, &jobs[t]) != 0) { load_job_run(&jobs[t]); tid[t] = 0; }
    }
    for (t = 0; t < threads; t++) if (tid[t]) pthread_join(tid[t], NULL);
    for (t = 0; t < threads; t++) { total += jobs[t].n; chars_total += jobs[t].set.total; }
    memset(out, 0, sizeof(*out));
    out->file_bytes = fsize;
    if (total > 0) {
        out->recs = (slh_record *)slh_big_malloc((size_t)total * sizeof(slh_record));
        out->offsets = (uint64_t *)slh_big_malloc(((size_t)total + 1) * sizeof(uint64_t));
        out->chars = (char *)slh_big_malloc(chars_total + 16);
        if (!out->recs || !out->offsets || !out->chars) ok = 0;
        else {
            uint64_t cpos = 0;
            int rpos = 0;
            for (t = 0; t < threads; t++) { /* where every piece goes; then the pieces are copied by the threads again */
                slh_seqset *s = &jobs[t].set;
                jobs[t].dst = out;
                jobs[t].dst_cpos = cpos;
                jobs[t].dst_rpos = rpos;
                if (jobs[t].n == 0) continue;
                cpos += s->total;
                rpos += s->num;
                if (s->name_arena) { /* the names stay where they are: chain the arenas */
                    name_chunk *last = (name_chunk *)s->name_arena;
                    while (last->next) last = last->next;
                    last->next = (name_chunk *)out->name_arena;
                    out->name_arena = s->name_arena;
                    s->name_arena = NULL;
                }
            }
            for (t = 0; t < threads; t++)
                if (pthread_create(&tid[t], NULL, concat_job_run, &jobs[t]) != 0) { concat_job_run(&jobs[t]); tid[t] = 0; }
            for (t = 0; t < threads; t++) if (tid[t]) pthread_join(tid[t], NULL);
            out->offsets[total] = cpos;
            memset(out->chars + cpos, 0, 16);
            out->total = cpos;
            out->num = total;
        }
    }
    for (t = 0; t < threads; t++) slh_free_seqset(&jobs[t].set);
    free(jobs); free(tid); free(cut);
    if (!ok) { slh_free_seqset(out); return -1; }
    return total;
}

int slh_load_file(const char *path, int merge, int acgt_only, uint32_t min_len, const char *name_filter,
                  int first_number, long log_limit, slh_seqset *out, FILE *log) {
    FILE *f;
    unsigned char *data = NULL;
    long fsize;
    int n, threads, mapped = 0;
    memset(out, 0, sizeof(*out));
    if (log) fprintf(log, "> Loading sequences from file <%s> ... ", path);
    f = fopen(path, "rb");
    if (!f) {
        if (log) fprintf(log, "\n> WARNING: Sequence file not found\n");
        return 0;
    }
    fseek(f, 0L, SEEK_END);
    fsize = ftell(f);
    rewind(f);
    if (log) fprintf(log, "(%ld bytes)\n", fsize);
    /* map the file instead of copying it: the parser threads fault its pages in parallel */
    if (fsize > 0) {
        void *m = mmap(NULL, (size_t)fsize, PROT_READ, MAP_PRIVATE, fileno(f), 0);
        if (m != MAP_FAILED) { data = (unsigned char *)m; mapped = 1; }
    }
    if (!mapped) {
        data = (unsigned char *)malloc(fsize > 0 ? (size_t)fsize : 1);
        if (!data || (fsize > 0 && fread(data, 1, (size_t)fsize, f) != (size_t)fsize)) {
            if (log) fprintf(log, "> WARNING: Cannot read file\n");
            free(data);
            fclose(f);
            return 0;
        }
    }
    fclose(f);
    threads = slh_thread_count();
    if (!merge && threads > 1 && fsize > (64L << 20) && log_limit > 0 && data[0] == '>') {
        n = load_parallel(data, fsize, acgt_only, min_len, first_number, log_limit, out, log, threads);
        if (n < 0) n = load_mem(data, fsize, merge, acgt_only, min_len, name_filter, first_number, log_limit, out, log);
    } else {
        n = load_mem(data, fsize, merge, acgt_only, min_len, name_filter, first_number, log_limit, out, log);
    }
    if (mapped) munmap(data, (size_t)fsize);
    else free(data);
    return n;
}

/* ---- a query file in pieces ----------------------------------------------------------------------------------
 * The same records as slh_load_file(path, 0, ...), handed out as consecutive sequence sets of about piece_bytes of the
 * file each (cut at "newline + '>'"), so that a front end can search the first reads while the rest is still parsed.
 * Every piece is parsed by all host threads (load_parallel).  The "> Loading ..." line and the per-record lines (first
 * piece only; they are limited to the first log_limit records anyway) go to `log` as with slh_load_file. */
struct slh_pieces {
    unsigned char *data;
    long fsize, pos, piece_bytes;
    int mapped, acgt_only, first_number, threads, first;
    uint32_t min_len;
    long log_limit;
    FILE *log;
    int release_parsed; /* drop the file pages of a piece once it is parsed (slh_pieces_release_parsed) */
};

slh_pieces *slh_pieces_open(const char *path, int acgt_only, uint32_t min_len, int first_number, long log_limit,
                            long piece_bytes, FILE *log) {
    slh_pieces *p = (slh_pieces *)calloc(1, sizeof(slh_pieces));
    FILE *f;
    if (!p) return NULL;
    if (log) fprintf(log, "> Loading sequences from file <%s> ... ", path);
    f = fopen(path, "rb");
    if (!f) {
        if (log) fprintf(log, "\n> WARNING: Sequence file not found\n");
        free(p);
        return NULL;
    }
    fseek(f, 0L, SEEK_END);
    p->fsize = ftell(f);
    rewind(f);
    if (log) fprintf(log, "(%ld bytes)\n", p->fsize);
    if (p->fsize > 0) {
        void *m = mmap(NULL, (size_t)p->fsize, PROT_READ, MAP_PRIVATE, fileno(f), 0);
        if (m != MAP_FAILED) { p->data = (unsigned char *)m; p->mapped = 1; }
    }
    if (!p->mapped) {
        p->data = (unsigned char *)malloc(p->fsize > 0 ? (size_t)p->fsize : 1);
        if (!p->data || (p->fsize > 0 && fread(p->data, 1, (size_t)p->fsize, f) != (size_t)p->fsize)) {
            if (log) fprintf(log, "> WARNING: Cannot read file\n");
            free(p->data);
            free(p);
            fclose(f);
            return NULL;
        }
    }
    fclose(f);
    p->piece_bytes = piece_bytes > (1L << 20) ? piece_bytes : (1L << 20);
    p->acgt_only = acgt_only;
    p->min_len = min_len;
    p->first_number = first_number;
    p->log_limit = log_limit;
    p->log = log;
    p->threads = slh_thread_count();
    p->first = 1;
    return p;
}

/* the next piece: number of records (> 0), 0 at the end of the file (or when the file holds no record at all) */
int slh_pieces_next(slh_pieces *p, slh_seqset *out) {
    memset(out, 0, sizeof(*out));
    while (p->pos < p->fsize) {
        long start = p->pos, end = start + p->piece_bytes;
        int n;
        if (end >= p->fsize) end = p->fsize;
        else {
            while (end < p->fsize && !(p->data[end] == '>' && (p->data[end - 1] == '\n' || p->data[end - 1] == '\r'))) end++;
        }
        p->pos = end;
        if (p->first && p->data[start] != '>') {  /* as load_mem: a file that does not start with '>' is not FASTA */
            if (p->log) fprintf(p->log, "> WARNING: Invalid FASTA file\n");
            p->pos = p->fsize;
            return 0;
        }
        if (p->threads > 1 && end - start > (16L << 20) && p->log_limit > 0)
            n = load_parallel(p->data + start, end - start, p->acgt_only, p->min_len, p->first_number, p->log_limit, out,
                              p->first ? p->log : NULL, p->threads);
        else n = -1;
        if (n < 0)
            n = load_mem(p->data + start, end - start, 0, p->acgt_only, p->min_len, NULL, p->first_number, p->log_limit, out,
                         p->first ? p->log : NULL);
        p->first = 0;
        if (p->release_parsed && p->mapped && end - start > (1L << 20)) { /* the parsed part of the file is not read again */
            long a = (start + 4095) & ~4095L, b = end & ~4095L;
            if (b > a) (void)madvise(p->data + a, (size_t)(b - a), MADV_DONTNEED);
        }
        if (n > 0) { p->first_number += n; return n; }
        /* a piece without an accepted record (all too short / empty): go on */
    }
    return 0;
}

void slh_pieces_release_parsed(slh_pieces *p, int on) {
    if (p) p->release_parsed = on;
}

void slh_pieces_close(slh_pieces *p) {
    if (!p) return;
    if (p->mapped) munmap(p->data, (size_t)p->fsize);
    else free(p->data);
    free(p);
}

int slh_seq_id_from_merged_pos(const uint32_t *starts, int num, uint32_t *pos) {
    int lo = 0, hi = num - 1;
    while (lo != hi) { /* binary search for the last start <= pos */
        int mid = (lo + hi + 1) / 2;
        if (*pos >= starts[mid]) lo = mid;
        else hi = mid - 1;
    }
    *pos -= starts[lo];
    return lo;
}

/* ------------------------------------------------------------------------------------------------ */
/* options                                                                                           */
/* ------------------------------------------------------------------------------------------------ */

int slh_parse_argument(int argc, char **argv, const char *optionchars, int parse) {
    char uc[2] = {0, 0}, lc[2] = {0, 0};
    int i;
    for (i = 0; i < 2 && optionchars[i] != '\0'; i++) { /* only the first two characters of the option name count */
        char ch = optionchars[i];
        if (ch >= 'A' && ch <= 'Z') { uc[i] = ch; lc[i] = (char)(ch + 32); }
        else { uc[i] = (char)(ch - 32); lc[i] = ch; }
    }
    for (i = 1; i < argc; i++) {
        const char *a = argv[i];
        /* a one-letter option must be exactly two characters long: its third must equal the '\0' in uc[1] */
        if (a[0] == '-' && a[1] != '\0' && (a[1] == lc[0] || a[1] == uc[0]) && (a[2] == lc[1] || a[2] == uc[1])) {
            if (parse) {
                if (i == argc - 1) return -1; /* nothing in front of it */
                if (parse == 1) return atoi(argv[i + 1]);
                return i + 1;
            }
            return 1;
        }
    }
    return parse ? -1 : 0;
}

char *slh_append_to_basename(const char *filename, const char *extra) {
    int n = (int)strlen(filename), i;
    char *res;
    for (i = n - 1; i > 0; i--)
        if (filename[i] == '.') break;
    if (i <= 0) i = n;
    res = (char *)calloc((size_t)i + strlen(extra) + 1, 1);
    if (!res) return NULL;
    memcpy(res, filename, (size_t)i);
    strcat(res, extra);
    return res;
}

void slh_free_options(slh_options *o) {
    free(o->ref_name);
    free(o->file_args);
    memset(o, 0, sizeof(*o));
}

int slh_parse_options(int argc, char **argv, slh_options *o) {
    int i, j, n = 0, ref_arg;
    char oc;
    memset(o, 0, sizeof(*o));
    o->image_arg = o->out_arg = -1;
    o->min_mem_len = 20;
    if (argc < 3) { o->usage = 1; return 0; }
    o->hidden_sort = slh_parse_argument(argc, argv, "S", 0);
    o->hidden_clean = slh_parse_argument(argc, argv, "C", 0);
    o->file_args = (int *)calloc((size_t)argc, sizeof(int));
    if (!o->file_args) return -1;
    for (i = 1; i < argc; i++) { /* which arguments are FASTA files (slamem.c:574-600) */
        if (argv[i][0] == '-') {
            oc = argv[i][1];
            if (oc >= 'A' && oc <= 'Z') oc = (char)('a' + (oc - 'A'));
            if (oc == 'l' || oc == 'o' || oc == 'm' || oc == 'v') i++; /* any option starting with l/o/m/v eats the next argument */
            else if (oc == 'r') {
                i++;
                if (i == argc) break;
                j = 0;
                oc = argv[i][0];
                if (oc == '\'' || oc == '\"') j = 1;
                else oc = '\0';
                n = 0;
                while (argv[i][j] != oc) {
                    if (argv[i][j] == '\0') {
                        i++;
                        if (i == argc) break;
                        j = 0;
                    } else j++;
                    n++;
                }
            }
            continue;
        }
        o->file_args[o->num_files++] = i;
    }
    o->no_ns = slh_parse_argument(argc, argv, "N", 0);
    o->min_seq_len = slh_parse_argument(argc, argv, "M", 1);
    if (o->min_seq_len == -1) o->min_seq_len = 0;
    ref_arg = slh_parse_argument(argc, argv, "R", 2);
    if (ref_arg != -1) { /* slamem.c:605-629 */
        o->ref_name_given = 1;
        if (n == 0) o->ref_name_empty = 1;
        else {
            o->ref_name = (char *)calloc((size_t)n + 1, 1);
            if (!o->ref_name) return -1;
            i = ref_arg;
            j = 0;
            oc = argv[i][0];
            if (oc == '\'' || oc == '\"') j = 1;
            else oc = '\0';
            n = 0;
            while (argv[i][j] != oc) {
                if (argv[i][j] == '\0') {
                    i++;
                    if (i == argc) break;
                    j = 0;
                    o->ref_name[n] = ' ';
                } else {
                    o->ref_name[n] = argv[i][j];
                    j++;
                }
                n++;
            }
            o->ref_name[n] = '\0';
        }
    }
    o->image_arg = slh_parse_argument(argc, argv, "V", 2);
    o->match_type = slh_parse_argument(argc, argv, "MA", 0) ? 1 : 0;
    o->both_strands = slh_parse_argument(argc, argv, "B", 0);
    o->min_mem_len = slh_parse_argument(argc, argv, "L", 1);
    if (o->min_mem_len == -1) o->min_mem_len = 20;
    o->out_arg = slh_parse_argument(argc, argv, "O", 2);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* output                                                                                            */
/* ------------------------------------------------------------------------------------------------ */

void slh_buffer_free(slh_buffer *b) {
    free(b->data);
    b->data = NULL;
    b->len = b->cap = 0;
}

static int buf_reserve(slh_buffer *b, size_t extra) {
    if (b->len + extra <= b->cap) return 0;
    size_t nc = b->cap ? b->cap * 2 : (1u << 16);
    while (nc < b->len + extra) nc *= 2;
    if (b->data == NULL) { /* first reservation: callers that know their size ask for all of it (huge pages if large) */
        char *fd = (char *)slh_big_malloc(nc);
        if (!fd) return -1;
        b->data = fd;
        b->cap = nc;
        return 0;
    }
    char *nd = (char *)realloc(b->data, nc);
    if (!nd) return -1;
    b->data = nd;
    b->cap = nc;
    return 0;
}

int slh_buffer_reserve(slh_buffer *b, size_t bytes) { return buf_reserve(b, bytes); }

/* decimal digits, two at a time (24 M lines of three numbers each per 10 M reads: the formatter is the front end's
 * longest stage once the search is on the GPU) */
static const char DIGIT_PAIRS[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

static inline char *put_u32(char *p, uint32_t v) {
    char tmp[10];
    int n = 10;
    while (v >= 100) {
        uint32_t q = v / 100, r = v - q * 100;
        n -= 2;
        memcpy(tmp + n, DIGIT_PAIRS + 2 * r, 2);
        v = q;
    }
    if (v >= 10) { n -= 2; memcpy(tmp + n, DIGIT_PAIRS + 2 * v, 2); }
    else tmp[--n] = (char)('0' + v);
    memcpy(p, tmp + n, (size_t)(10 - n));
    return p + (10 - n);
}

int slh_format_block(slh_buffer *buf, const char *query_name, int reverse, const uint32_t *mems, uint64_t count,
                     const slh_record *refs, const uint32_t *merged_start, int num_refs, uint64_t *sum_len_out) {
    size_t nl = strlen(query_name);
    uint64_t i, sum = 0;
    char *p;
    /* room for the whole block at once when its lines have a known bound (one reference record: three numbers of at
       most ten digits, two tabs, a newline) */
    if (buf_reserve(buf, nl + 16 + (num_refs == 1 ? (size_t)count * 33 : 0))) return -1;
    p = buf->data + buf->len;
    *p++ = '>';
    memcpy(p, query_name, nl);
    p += nl;
    if (reverse) { memcpy(p, " Reverse", 8); p += 8; } /* slamem.c:102 */
    *p++ = '\n';
    if (num_refs == 1) { /* slamem.c:148 */
        for (i = 0; i < count; i++) {
            const uint32_t ln = mems[3 * i + 2];
            p = put_u32(p, mems[3 * i] + 1);
            *p++ = '\t';
            p = put_u32(p, mems[3 * i + 1] + 1);
            *p++ = '\t';
            p = put_u32(p, ln);
            *p++ = '\n';
            sum += ln;
        }
        buf->len = (size_t)(p - buf->data);
        if (sum_len_out) *sum_len_out = sum;
        return 0;
    }
    buf->len = (size_t)(p - buf->data);
    for (i = 0; i < count; i++) { /* slamem.c:144-148: several reference records, every line names its record */
        uint32_t rp = mems[3 * i], qp = mems[3 * i + 1], ln = mems[3 * i + 2];
        int id = slh_seq_id_from_merged_pos(merged_start, num_refs, &rp);
        const char *rname = refs[id].name;
        size_t namelen = strlen(rname);
        if (buf_reserve(buf, namelen + 48)) return -1;
        p = buf->data + buf->len;
        *p++ = ' ';
        memcpy(p, rname, namelen);
        p += namelen;
        *p++ = '\t';
        p = put_u32(p, rp + 1);
        *p++ = '\t';
        p = put_u32(p, qp + 1);
        *p++ = '\t';
        p = put_u32(p, ln);
        *p++ = '\n';
        buf->len = (size_t)(p - buf->data);
        sum += ln;
    }
    if (sum_len_out) *sum_len_out = sum;
    return 0;
}

int slh_progress_dots(uint32_t textsize) {
    uint32_t step = textsize / 10; /* slamem.c:94 */
    return (int)(textsize / (step + 1)); /* one dot each time the counter reaches the step (slamem.c:116-120) */
}

/* ------------------------------------------------------------------------------------------------
 * The two hidden utilities of the reference's command line (slamem.c:555-570).  Plain host text
 * processing; kept so that every invocation of the reference has a counterpart here.
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    char ref_name[65]; /* first word of the reference name, at most 64 characters (slamem.c:221,330) */
    int ref_pos, query_pos, size;
    size_t order;      /* input order: ties keep it (glibc's qsort is a merge sort) */
} sorted_mem;

static int sorted_mem_cmp(const void *pa, const void *pb) { /* MEMInfoSortFunction, slamem.c:227-239 */
    const sorted_mem *a = (const sorted_mem *)pa, *b = (const sorted_mem *)pb;
    const char *ca = a->ref_name, *cb = b->ref_name;
    int diff = 0;
    while ((diff = (int)(*ca) - (int)(*cb)) == 0 && *ca != '\0') { ca++; cb++; }
    if (diff == 0) {
        diff = a->ref_pos - b->ref_pos;
        if (diff == 0) diff = a->query_pos - b->query_pos;
    }
    if (diff == 0) diff = a->order < b->order ? -1 : (a->order > b->order ? 1 : 0);
    return diff;
}

/* a byte source with the reference's `char c = fgetc()` view: byte 0xFF reads as end of input */
typedef struct { const unsigned char *p, *end; } byte_src;
static int src_get(byte_src *s) {
    if (s->p >= s->end) return -1;
    if (*s->p == 0xFF) { s->p = s->end; return -1; }
    return *s->p++;
}
static int is_space_c(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }
static void skip_space(byte_src *s) { while (s->p < s->end && is_space_c(*s->p)) s->p++; }
/* scanf("%d"): optional sign, digits; returns 0 when no number starts here */
static int scan_int(byte_src *s, int *out) {
    const unsigned char *q;
    long long v = 0;
    int neg = 0, digits = 0;
    skip_space(s);
    q = s->p;
    if (q < s->end && (*q == '+' || *q == '-')) { neg = *q == '-'; q++; }
    while (q < s->end && *q >= '0' && *q <= '9') { if (v < (1LL << 40)) v = v * 10 + (*q - '0'); q++; digits++; }
    if (!digits) return 0;
    s->p = q;
    *out = (int)(neg ? -v : v);
    return 1;
}

static char *read_whole_file(const char *path, size_t *size_out) {
    FILE *f = fopen(path, "rb");
    char *buf;
    long sz;
    if (!f) return NULL;
    if (fseek(f, 0, SEEK_END) != 0 || (sz = ftell(f)) < 0) { fclose(f); return NULL; }
    rewind(f);
    buf = (char *)malloc((size_t)sz + 1);
    if (!buf) { fclose(f); return NULL; }
    if (sz && fread(buf, 1, (size_t)sz, f) != (size_t)sz) { free(buf); fclose(f); return NULL; }
    fclose(f);
    *size_out = (size_t)sz;
    return buf;
}

/* SortMEMsFile (slamem.c:244-352): every block of the MEMs file sorted by (reference name, reference position, query
 * position) and written in DESCENDING order (the reference walks its sorted array from the end, :310-317) to
 * <basename>-sorted.txt.  Returns the process status: 0, or 255 after an error message. */
int slh_sort_mems_file(const char *path, FILE *log) {
    size_t size = 0, cap = 0, num = 0, seqs = 0, k;
    char *data, *out_name, seqname[256];
    sorted_mem *arr = NULL;
    byte_src s;
    FILE *out;
    int c, fields = 0;
    fprintf(log, "> Sorting MEMs from <%s> ", path);
    fflush(log);
    data = read_whole_file(path, &size);
    if (!data) { fprintf(log, "\n> ERROR: Cannot read input file\n"); return 255; }
    s.p = (const unsigned char *)data;
    s.end = s.p + size;
    c = src_get(&s);
    if (c != '>') { fprintf(log, "\n> ERROR: Invalid MEMs file\n"); free(data); return 255; }
    while (c == '>') { /* :261-265: skip the header lines in front of the first MEM */
        c = src_get(&s);
        while (c != '\n' && c != -1) c = src_get(&s);
        c = src_get(&s);
    }
    if (c == -1) { fprintf(log, "\n> ERROR: No MEMs inside file\n"); free(data); return 255; }
    for (;;) { /* :271-278: number of fields of that line */
        while (c == ' ' || c == '\t') c = src_get(&s);
        if (c != '\n') {
            fields++;
            while (c != ' ' && c != '\t' && c != '\n' && c != -1) c = src_get(&s);
        }
        if (c == '\n' || c == -1) break;
    }
    if (fields != 3 && fields != 4) { fprintf(log, "\n> ERROR: Invalid MEMs file format\n"); free(data); return 255; }
    s.p = (const unsigned char *)data;
    s.end = s.p + size;
    if (fields == 4) fprintf(log, "(multiple references) ");
    fprintf(log, "...\n");
    out_name = slh_append_to_basename(path, "-sorted.txt");
    if (!out_name || (out = fopen(out_name, "w")) == NULL) {
        fprintf(log, "> ERROR: Cannot write output file\n");
        free(out_name); free(data);
        return 255;
    }
    seqname[0] = '\0';
    for (;;) {
        c = src_get(&s);
        if (c == '>' || c == -1) {
            if (seqs != 0) { /* :304-318 */
                fprintf(log, "(%d MEMs)\n", (int)num);
                fflush(log);
                if (num) qsort(arr, num, sizeof(sorted_mem), sorted_mem_cmp);
                fprintf(out, ">%s\n", seqname);
                for (k = num; k-- > 0;) {
                    if (fields == 4) fprintf(out, " %s\t", arr[k].ref_name);
                    fprintf(out, "%d\t%d\t%d\n", arr[k].ref_pos, arr[k].query_pos, arr[k].size);
                }
            }
            if (c == -1) break;
            num = 0;
            { /* fscanf(" %255[^\n]\n") :322 */
                size_t n = 0;
                skip_space(&s);
                while (s.p < s.end && *s.p != '\n' && n < 255) seqname[n++] = (char)*s.p++;
                seqname[n] = '\0';
                skip_space(&s);
            }
            fprintf(log, ":: '%s' ... ", seqname);
            fflush(log);
            seqs++;
            continue;
        }
        s.p--; /* ungetc */
        if (num == cap) {
            sorted_mem *na = (sorted_mem *)realloc(arr, (cap + 1024) * sizeof(sorted_mem));
            if (!na) { fprintf(log, "\n> ERROR: Not enough memory\n"); fclose(out); free(out_name); free(arr); free(data); return 255; }
            arr = na;
            cap += 1024;
        }
        if (fields == 4) { /* fscanf(" %64[^\t ]") :330 */
            size_t n = 0;
            skip_space(&s);
            while (s.p < s.end && *s.p != '\t' && *s.p != ' ' && n < 64) arr[num].ref_name[n++] = (char)*s.p++;
            arr[num].ref_name[n] = '\0';
        } else arr[num].ref_name[0] = '\0';
        if (!scan_int(&s, &arr[num].ref_pos) || !scan_int(&s, &arr[num].query_pos) || !scan_int(&s, &arr[num].size)) {
            fprintf(log, "\n> ERROR: Invalid format\n"); /* :333-337 (the reference also waits for a key here) */
            fclose(out); free(out_name); free(arr); free(data);
            return 255;
        }
        skip_space(&s);
        arr[num].order = num;
        num++;
    }
    fprintf(log, "> Saving sorted MEMs to <%s> ...\n", out_name);
    fflush(log);
    fclose(out);
    free(out_name);
    free(arr);
    free(data);
    fprintf(log, "> Done!\n");
    return 0;
}

/* CleanFasta (slamem.c:455-523): one record named after the file, holding the A/C/G/T letters (upper-cased) of all
 * records of the input, 100 per line, written to <basename>-clean.fasta. */
int slh_clean_fasta(const char *path, FILE *log) {
    size_t size = 0;
    char *data, *out_name;
    byte_src s;
    FILE *out;
    unsigned int chars = 0, invalid = 0, seqs = 0, line = 0;
    int c;
    fprintf(log, "> Opening FASTA file <%s> ... ", path);
    fflush(log);
    data = read_whole_file(path, &size);
    if (!data) { fprintf(log, "\n> ERROR: FASTA file not found\n"); return 255; }
    s.p = (const unsigned char *)data;
    s.end = s.p + size;
    c = src_get(&s);
    if (c != '>') { fprintf(log, "\n> ERROR: Invalid FASTA file\n"); free(data); return 255; }
    fprintf(log, "OK\n");
    out_name = slh_append_to_basename(path, "-clean.fasta");
    fprintf(log, "> Creating clean FASTA file <%s> ... ", out_name ? out_name : "");
    fflush(log);
    if (!out_name || (out = fopen(out_name, "w")) == NULL) {
        fprintf(log, "\n> ERROR: Can't write clean FASTA file\n");
        free(out_name); free(data);
        return 255;
    }
    fprintf(out, ">%s\n", path); /* the file name is the label (:479) */
    while (c != -1) {
        int up = (c == 'a' || c == 'c' || c == 'g' || c == 't') ? c - 32 : c;
        if (up == 'A' || up == 'C' || up == 'G' || up == 'T') {
            fputc(up, out);
            chars++;
            if (++line == 100) { fputc('\n', out); line = 0; }
        } else if (c == '>') {
            while (c != '\n' && c != -1) c = src_get(&s); /* skip the description */
            seqs++;
        } else if (c > 32 && c < 127) invalid++;
        c = src_get(&s);
    }
    if (line != 0) fputc('\n', out);
    fclose(out);
    free(out_name);
    free(data);
    fprintf(log, " OK\n");
    fprintf(log, ":: %u total chars", chars);
    if (invalid != 0) fprintf(log, " (%u non ACGT chars removed)", invalid);
    if (seqs > 1) fprintf(log, " ; %u sequences merged", seqs);
    fprintf(log, "\n");
    fprintf(log, "> Done!\n");
    return 0;
}
