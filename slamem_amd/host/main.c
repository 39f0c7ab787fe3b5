/* main.c -- slaMEM-compatible command line in front of the MI355X engine.
 *
 *   slaMEM-hip (<options>) <reference_file> <query_file(s)>
 *
 * Keeps the reference's options, stdout shape and *-mems.txt format (slamem.c:528-672, 37-218); the
 * index build and the MEM search run on the GPU through the C ABI of include/slamem_hip.h.
 * There is no CPU fallback: without a usable GPU the program prints the library's error and exits 255.
 *
 * Environment: SLAMEM_DEVICE (default 0) selects the GPU; SLAMEM_VERBOSE=1 prints one line per loaded
 * record and per strand for any number of records, as the reference does (default: first 100 only);
 * SLAMEM_BATCH_MB bounds the query characters sent to the GPU per batch (default 1024).
 */
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <pthread.h>

#include "../../include/slamem_hip.h"
#include "../../include/slamem_rccl.h"
#include "slamem_host.h"

#define VERSION "0.8.2"
static const char MATCH_TYPE_CHAR[] = "EAU";

/* the device warm-up thread (see main) is joined before the process ends, whichever way it ends */
static pthread_t g_warm_tid;
static int g_warm_started = 0;
static void join_warmup(void) {
    if (g_warm_started) { g_warm_started = 0; pthread_join(g_warm_tid, NULL); }
}

static void exit_message(const char *msg) { /* tools.c:21-25 */
    printf("> ERROR: %s\n", msg);
    join_warmup();
    exit(-1);
}

static void gpu_fail(const char *what, int rc) {
    printf("\n> ERROR: %s failed: %s (%s)\n", what, slamem_strerror(rc), slamem_last_error_message());
    join_warmup();
    exit(-1);
}

static void gpu_fail_msg(const char *what, int rc, const char *detail) { /* the detail was captured on another thread */
    printf("\n> ERROR: %s failed: %s (%s)\n", what, slamem_strerror(rc), detail);
    join_warmup();
    exit(-1);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* formatting of a range of strand blocks of one batch, one range per thread */
typedef struct {
    const slh_seqset *q;
    const slh_seqset *ref;
    const slamem_mem *mems;
    const uint64_t *boff;
    int first_rec, strands;
    uint64_t b0, b1; /* strand blocks [b0,b1) of the batch */
    slh_buffer buf;
    long long matches, sum;
    int failed;
} fmt_job;

static void *fmt_run(void *arg) {
    fmt_job *j = (fmt_job *)arg;
    uint64_t b;
    /* about 36 characters per MEM line and a header per block: reserve once instead of doubling on the way */
    (void)slh_buffer_reserve(&j->buf, (size_t)(j->boff[j->b1] - j->boff[j->b0]) * 36 + (size_t)(j->b1 - j->b0) * 48 + 4096);
    for (b = j->b0; b < j->b1; b++) {
        int i = j->first_rec + (int)(b / (uint64_t)j->strands), s = (int)(b % (uint64_t)j->strands);
        uint64_t cnt = j->boff[b + 1] - j->boff[b], sum = 0;
        if (slh_format_block(&j->buf, j->q->recs[i].name, s, (const uint32_t *)(j->mems + j->boff[b]), cnt, j->ref->recs,
                             j->ref->merged_start, j->ref->num, &sum)) { j->failed = 1; return NULL; }
        j->matches += (long long)cnt;
        j->sum += (long long)sum;
    }
    return NULL;
}

/* one contiguous share of a batch's records, searched on one GPU by one host thread */
typedef struct {
    slamem_index *idx;
    const char *chars;
    uint64_t *offs;
    int first, last; /* records [first,last) of the query set */
    uint32_t min_len;
    int both, mam;
    slamem_mem *mems;
    uint64_t *boff, total;
    int rc;
    char err[512];
} gpu_part;

static void *gpu_part_run(void *arg) {
    gpu_part *g = (gpu_part *)arg;
    g->rc = (g->mam ? slamem_find_mams_host : slamem_find_mems_host)(g->idx, g->chars, g->offs, (uint32_t)(g->last - g->first),
                                                                     g->min_len, g->both, &g->mems, &g->boff, &g->total);
    if (g->rc != SLAMEM_OK) snprintf(g->err, sizeof(g->err), "%s", slamem_last_error_message()); /* the message is per thread */
    return NULL;
}

/* The index is built on the GPU by its own host thread while the main thread parses the query files: the two do not
 * depend on each other (the reference does them one after the other, slamem.c:635-651 then :73-74). */
typedef struct {
    const char *text;
    uint32_t n;
    int device;
    slamem_index *idx;
    slamem_index_info info;
    slamem_timings tm;
    slamem_sslcp_stats st;
    int rc, rc_stats;
    double seconds;
    char err[512];
} build_job;

static void *build_run(void *arg) {
    build_job *b = (build_job *)arg;
    double t0 = now_s();
    b->rc = slamem_index_build(b->text, b->n, b->device, &b->idx);
    if (b->rc == SLAMEM_OK) {
        slamem_index_get_info(b->idx, &b->info);
        slamem_get_timings(&b->tm); /* timings and error text are per thread: take them here */
        b->rc_stats = slamem_index_sampled_lcp_stats(b->idx, &b->st);
    }
    if (b->rc != SLAMEM_OK || b->rc_stats != SLAMEM_OK) snprintf(b->err, sizeof(b->err), "%s", slamem_last_error_message());
    b->seconds = now_s() - t0;
    return NULL;
}

/* One batch of query records: its shares are searched on the GPUs by their own host threads while the main thread
 * formats and writes the batch before it (GPU(b+1) overlaps format(b)). */
typedef struct {
    const slh_seqset *q;
    int first, last, nparts;
    gpu_part parts[16];
    pthread_t tid[16];
    int threaded[16];
} batch_t;

static batch_t *batch_start(const slh_seqset *q, int first, int last, slamem_index **gpus, int ngpu, uint32_t min_len,
                            int both, int mam) {
    batch_t *b = (batch_t *)calloc(1, sizeof(batch_t));
    uint64_t base = q->offsets[first], tot = q->offsets[last] - base;
    int part, r0 = first, i;
    if (!b) return NULL;
    b->q = q; b->first = first; b->last = last; b->nparts = ngpu;
    for (part = 0; part < ngpu; part++) { /* contiguous shares with (almost) equal numbers of bases, one per GPU */
        uint64_t target = base + tot * (uint64_t)(part + 1) / (uint64_t)ngpu;
        int r1 = r0;
        gpu_part *g = &b->parts[part];
        if (part == ngpu - 1) r1 = last;
        else while (r1 < last && q->offsets[r1 + 1] <= target) r1++;
        g->idx = gpus[part];
        g->first = r0;
        g->last = r1;
        g->chars = q->chars + q->offsets[r0];
        g->offs = (uint64_t *)malloc(((size_t)(r1 - r0) + 1) * sizeof(uint64_t));
        if (!g->offs) return NULL;
        for (i = r0; i <= r1; i++) g->offs[i - r0] = q->offsets[i] - q->offsets[r0];
        g->min_len = min_len;
        g->both = both;
        g->mam = mam;
        r0 = r1;
    }
    for (part = 0; part < ngpu; part++) {
        b->threaded[part] = pthread_create(&b->tid[part], NULL, gpu_part_run, &b->parts[part]) == 0;
        if (!b->threaded[part]) gpu_part_run(&b->parts[part]);
    }
    return b;
}

static void batch_join(batch_t *b, int device) {
    int part;
    for (part = 0; part < b->nparts; part++) {
        if (b->threaded[part]) pthread_join(b->tid[part], NULL);
        b->threaded[part] = 0;
    }
    for (part = 0; part < b->nparts; part++)
        if (b->parts[part].rc != SLAMEM_OK) {
            printf("\n> ERROR: MEM search on GPU %d failed: %s (%s)\n", device + part, slamem_strerror(b->parts[part].rc),
                   b->parts[part].err);
            exit(-1);
        }
}

static void *warmup_run(void *arg) { /* HIP runtime + context start-up, hidden behind the parsing of the reference file */
    (void)slamem_device_warmup(*(int *)arg);
    return NULL;
}

static void usage(const char *prog) { /* slamem.c:533-553 */
    printf("Usage:\n");
    printf("\t%s (<options>) <reference_file> <query_file(s)>\n", prog);
    printf("Options:\n");
    printf("\t-mem\tfind MEMs: any number of occurrences in both ref and query (default)\n");
    printf("\t-mam\tfind MAMs: unique in ref but any number in query\n");
    printf("\t-l\tminimum match length (default=20)\n");
    printf("\t-o\toutput file name (default=\"*-mems.txt\")\n");
    printf("\t-b\tprocess both forward and reverse strands\n");
    printf("\t-n\tdiscard 'N' characters in the sequences\n");
    printf("\t-m\tminimum sequence size (e.g. to ignore small scaffolds)\n");
    printf("\t-r\tload only the reference(s) whose name(s) contain(s) this string\n");
    printf("Example:\n");
    printf("\t%s -b -l 10 ./ref.fna ./query.fna\n", prog);
}

int main(int argc, char **argv) {
    slh_options o;
    slh_seqset ref, *qsets;
    int i, f, num_qsets = 0, total_queries = 0, device = 0, numbering = 1;
    long log_limit = 100;
    uint64_t batch_bytes = 256ull << 20; /* query characters per batch and GPU */
    const char *env;
    char *out_name;
    FILE *out;
    slamem_index *idx = NULL, *gpus[16];
    int rc, ngpu = 1;
    double t0;
    long long total_matches = 0, total_sum = 0;
    slh_buffer buf = {0, 0, 0};
    build_job bj;
    pthread_t build_tid;
    int build_async = 0;

    printf("[ slaMEM v%s ]\n\n", VERSION);
    if (slh_parse_options(argc, argv, &o) != 0) exit_message("Out of memory");
    if (o.usage) { usage(argv[0]); slh_free_options(&o); return -1; }
    if (o.hidden_sort) { /* slamem.c:555-562 */
        slh_free_options(&o);
        if (argc != 3) { printf("Usage: %s -s <mems_file>\n\n", argv[0]); return -1; }
        return slh_sort_mems_file(argv[2], stdout);
    }
    if (o.hidden_clean) { /* slamem.c:563-570 */
        slh_free_options(&o);
        if (argc != 3) { printf("Usage: %s -c <fasta_file>\n\n", argv[0]); return -1; }
        return slh_clean_fasta(argv[2], stdout);
    }
    if (o.num_files < 2) exit_message("Not enough input sequence files provided");
    if (o.ref_name_given && o.ref_name_empty) exit_message("No reference name string provided");
    if (o.image_arg != -1) exit_message("The -v image tool is not part of this front end (use the reference's on the *-mems.txt output)");
    if ((env = getenv("SLAMEM_DEVICE")) != NULL) device = atoi(env);
    if ((env = getenv("SLAMEM_VERBOSE")) != NULL && atoi(env) != 0) log_limit = 0;
    if ((env = getenv("SLAMEM_BATCH_MB")) != NULL && atoll(env) > 0) batch_bytes = (uint64_t)atoll(env) << 20;

    {
        static int warm_device;
        warm_device = device;
        g_warm_started = pthread_create(&g_warm_tid, NULL, warmup_run, &warm_device) == 0;
    }
    double t_start = now_s(), t_load = 0, t_build = 0, t_gpu = 0, t_format = 0, t_write = 0;
    int timing = getenv("SLAMEM_TIMING") != NULL;
    /* load everything (slamem.c:635-651) */
    memset(&ref, 0, sizeof(ref));
    memset(&bj, 0, sizeof(bj));
    qsets = (slh_seqset *)calloc((size_t)o.num_files, sizeof(slh_seqset));
    if (!qsets) exit_message("Out of memory");
    {
        int have_ref = 0;
        for (f = 0; f < o.num_files; f++) {
            const char *path = argv[o.file_args[f]];
            if (!have_ref) {
                int n = slh_load_file(path, 1, o.no_ns, (uint32_t)o.min_seq_len, o.ref_name, numbering, log_limit, &ref, stdout);
                if (n == 0) exit_message("No valid sequences found in reference file");
                have_ref = 1;
                numbering += n;
                o.file_args[0] = o.file_args[f]; /* remember which argument was the reference */
                bj.text = ref.chars;
                bj.n = (uint32_t)ref.total;
                bj.device = device;
                build_async = pthread_create(&build_tid, NULL, build_run, &bj) == 0;
            } else {
                int n = slh_load_file(path, 0, o.no_ns, (uint32_t)o.min_seq_len, NULL, numbering, log_limit, &qsets[num_qsets], stdout);
                if (n != 0) { numbering += n; total_queries += n; num_qsets++; }
            }
        }
    }
    t_load = now_s() - t_start;
    if (build_async) pthread_join(build_tid, NULL); /* before any exit: never leave the process with a build in flight */
    else build_run(&bj);
    join_warmup();
    if (num_qsets == 0) exit_message("No query files provided"); /* slamem.c:648 */
    printf("> %d reference%s and %d quer%s successfully loaded\n", ref.num, ref.num == 1 ? "" : "s", total_queries,
           total_queries == 1 ? "y" : "ies");
    if (o.min_mem_len < 1) exit_message("Minimum match length must be at least 1");

    if (o.out_arg == -1) out_name = slh_append_to_basename(argv[o.file_args[0]], "-mems.txt");
    else out_name = argv[o.out_arg];

    /* GetMatches (slamem.c:37-218) */
    printf("> Using options: minimum M%cM length = %d ; strand = %s\n", MATCH_TYPE_CHAR[o.match_type], o.min_mem_len,
           o.both_strands == 0 ? "forward only" : "forward + reverse");
    out = fopen(out_name, "w");
    if (!out) {
        printf("\n> ERROR: Cannot create output file <%s>\n", out_name);
        exit(-1);
    }
    printf("> Building index for reference sequence");
    if (ref.num == 1) printf(" \"%s\"", ref.recs[0].name);
    else printf("s");
    printf(" (%u Mbp) ...\n", (unsigned)(ref.total / 1000000U));
    fflush(stdout);
    t0 = now_s();
    idx = bj.idx;
    if (bj.rc != SLAMEM_OK) gpu_fail_msg("index construction on the GPU", bj.rc, bj.err);
    {
        slamem_index_info info = bj.info;
        slamem_timings tm = bj.tm;
        printf("> Suffix sort + BWT + LCP + parent links on GPU %d ... OK (%.3f s, overlapped with the loading of the queries; device %.1f ms: sort %.1f, BWT %.1f, LCP %.1f, links %.1f; %u doubling rounds)\n",
               device, bj.seconds, tm.build_total_ms, tm.build_sort_ms, tm.build_bwt_ms, tm.build_lcp_ms, tm.build_links_ms,
               info.sort_rounds);
        printf(":: Index size = %.1f MB in HBM (FM blocks + 16 B per row)\n", (double)info.arena_bytes / 1e6);
        {   /* the statistics lines of BuildSampledLCPArray (lcparray.c:709-711, 999-1000), from the per-row records */
            slamem_sslcp_stats st = bj.st;
            unsigned bwt_len = info.bwt_size;
            if (bj.rc_stats != SLAMEM_OK) gpu_fail_msg("LCP sampling statistics", bj.rc_stats, bj.err);
            printf(":: %.2lf%% samples (%u of %u)\n", ((double)st.num_samples / (double)bwt_len) * 100.0, (unsigned)st.num_samples, bwt_len);
            printf(":: %.2lf%% oversized samples (%d of %u)\n", ((double)st.num_oversized_lcp / (double)st.num_samples) * 100.0,
                   (int)st.num_oversized_lcp, (unsigned)st.num_samples);
            printf(":: Average LCP value = %d (max=%lld)\n", (int)(st.sum_lcp / (long long)bwt_len), (long long)st.max_lcp);
            printf(":: %.2lf%% oversized values (%d of %u)\n", ((double)st.num_oversized_links / (double)st.num_samples) * 100.0,
                   (int)st.num_oversized_links, (unsigned)st.num_samples);
            printf(":: Average SV distance = %.2lf (max=%lld)\n", (double)st.sum_link_distance / (double)st.num_samples,
                   (long long)st.max_link_distance);
        }
    }
    /* SLAMEM_GPUS=N: replicate the index to N GPUs (device, device+1, ...) with one RCCL broadcast of its arena over
       xGMI; every batch is then split by bases into N contiguous shares, one per GPU (no data-path collective). */
    gpus[0] = idx;
    if ((env = getenv("SLAMEM_GPUS")) != NULL) {
        int avail = 0;
        slamem_device_count(&avail);
        ngpu = atoi(env) <= 0 ? avail - device : atoi(env);
        if (ngpu < 1) ngpu = 1;
        if (ngpu > 16) ngpu = 16;
        if (device + ngpu > avail) exit_message("SLAMEM_GPUS asks for more GPUs than are visible");
    }
    if (ngpu > 1 || getenv("SLAMEM_REPLICATE_SELFTEST") != NULL) {
        int devs[16], g;
        double tr = now_s();
        for (g = 0; g < ngpu; g++) devs[g] = device + g;
        rc = slamem_index_replicate(idx, devs, ngpu, ngpu == 1, gpus);
        if (rc != SLAMEM_OK) gpu_fail("index replication over RCCL", rc);
        if (ngpu == 1) { slamem_index_free(idx); idx = gpus[0]; } /* self-test: search on the broadcast copy */
        printf("> Index replicated to %d GPU%s by RCCL broadcast ... OK (%.3f s)\n", ngpu, ngpu == 1 ? " (self-test copy)" : "s", now_s() - tr);
    }
    t_build = bj.seconds + (now_s() - t0);
    free(ref.chars); /* the reference frees the text here too (slamem.c:75-77) */
    ref.chars = NULL;
    printf("> Matching query sequences against index ...\n");
    fflush(stdout);

    {
        int strands = o.both_strands ? 2 : 1;
        long printed = 0;
        /* the list of batches (records [first,last) of one query file each), then a two-stage pipeline over it */
        typedef struct { int f, first, last; } batch_range;
        batch_range *ranges = NULL;
        size_t nranges = 0, cap_ranges = 0, bi;
        batch_t *cur = NULL, *nxt = NULL;
        for (f = 0; f < num_qsets; f++) {
            slh_seqset *q = &qsets[f];
            int first = 0;
            while (first < q->num) {
                int last = first;
                uint64_t base = q->offsets[first];
                while (last < q->num && (last == first || q->offsets[last + 1] - base <= batch_bytes * (uint64_t)ngpu)) last++;
                if (nranges == cap_ranges) {
                    cap_ranges = cap_ranges ? cap_ranges * 2 : 16;
                    ranges = (batch_range *)realloc(ranges, cap_ranges * sizeof(batch_range));
                    if (!ranges) exit_message("Out of memory");
                }
                ranges[nranges].f = f; ranges[nranges].first = first; ranges[nranges].last = last;
                nranges++;
                first = last;
            }
        }
        if (nranges) {
            cur = batch_start(&qsets[ranges[0].f], ranges[0].first, ranges[0].last, gpus, ngpu, (uint32_t)o.min_mem_len,
                              o.both_strands, o.match_type == 1);
            if (!cur) exit_message("Out of memory");
        }
        for (bi = 0; bi < nranges; bi++) {
            slh_seqset *q = &qsets[ranges[bi].f];
            gpu_part *parts;
            int part;
            double tg = now_s();
            batch_join(cur, device);
            if (bi + 1 < nranges) { /* the next batch goes to the GPUs while this one is formatted and written */
                nxt = batch_start(&qsets[ranges[bi + 1].f], ranges[bi + 1].first, ranges[bi + 1].last, gpus, ngpu,
                                  (uint32_t)o.min_mem_len, o.both_strands, o.match_type == 1);
                if (!nxt) exit_message("Out of memory");
            } else nxt = NULL;
            parts = cur->parts;
            {
                t_gpu += now_s() - tg; /* time the main thread waited for the GPUs */
                tg = now_s();
                for (part = 0; part < ngpu; part++) {
                    const int pfirst = parts[part].first, plast = parts[part].last;
                    slamem_mem *mems = parts[part].mems;
                    uint64_t *boff = parts[part].boff;
#define first pfirst
#define last plast
                    /* the first strand blocks get their ':: "name" ....' line (slamem.c:97,101,203) and are formatted here;
                       the rest of the batch is formatted by all host threads and written in order */
                    uint64_t nblk = (uint64_t)(last - first) * strands, bseq = 0, b;
                    int nthr = slh_thread_count(), t;
                    if (log_limit == 0) bseq = nblk;
                    else if (printed < log_limit) bseq = (uint64_t)(log_limit - printed) < nblk ? (uint64_t)(log_limit - printed) : nblk;
                    for (b = 0; b < bseq; b++) {
                        int ri = first + (int)(b / strands), s = (int)(b % strands), d, dots;
                        uint64_t cnt = boff[b + 1] - boff[b], sum = 0;
                        if (slh_format_block(&buf, q->recs[ri].name, s, (const uint32_t *)(mems + boff[b]), cnt, ref.recs,
                                             ref.merged_start, ref.num, &sum))
                            exit_message("Out of memory");
                        total_matches += (long long)cnt;
                        total_sum += (long long)sum;
                        dots = slh_progress_dots(q->recs[ri].size);
                        printf(":: \"%s%s\" ", q->recs[ri].name, s ? " Reverse" : "");
                        for (d = 0; d < dots; d++) putchar('.');
                        printf(" (%d M%cMs ; avg size = %d bp)\n", (int)cnt, MATCH_TYPE_CHAR[o.match_type], (int)(cnt ? sum / cnt : 0));
                        printed++;
                    }
                    if (buf.len) {
                        double tw = now_s();
                        if (fwrite(buf.data, 1, buf.len, out) != buf.len) exit_message("Cannot write output file");
                        buf.len = 0;
                        t_write += now_s() - tw;
                    }
                    if (bseq < nblk) {
                        fmt_job *jobs;
                        pthread_t *tid;
                        uint64_t per;
                        if ((nblk - bseq) < 4096 || nthr < 1) nthr = 1;
                        jobs = (fmt_job *)calloc((size_t)nthr, sizeof(fmt_job));
                        tid = (pthread_t *)calloc((size_t)nthr, sizeof(pthread_t));
                        if (!jobs || !tid) exit_message("Out of memory");
                        per = (nblk - bseq + (uint64_t)nthr - 1) / (uint64_t)nthr;
                        for (t = 0; t < nthr; t++) {
                            jobs[t].q = q; jobs[t].ref = &ref; jobs[t].mems = mems; jobs[t].boff = boff;
                            jobs[t].first_rec = first; jobs[t].strands = strands;
                            jobs[t].b0 = bseq + per * (uint64_t)t < nblk ? bseq + per * (uint64_t)t : nblk;
                            jobs[t].b1 = jobs[t].b0 + per < nblk ? jobs[t].b0 + per : nblk;
                            if (t == nthr - 1 || pthread_create(&tid[t], NULL, fmt_run, &jobs[t]) != 0) { fmt_run(&jobs[t]); tid[t] = 0; }
                        }
                        for (t = 0; t < nthr; t++) {
                            double tw;
                            if (tid[t]) pthread_join(tid[t], NULL);
                            if (jobs[t].failed) exit_message("Out of memory");
                            total_matches += jobs[t].matches;
                            total_sum += jobs[t].sum;
                            tw = now_s();
                            if (jobs[t].buf.len && fwrite(jobs[t].buf.data, 1, jobs[t].buf.len, out) != jobs[t].buf.len)
                                exit_message("Cannot write output file");
                            t_write += now_s() - tw;
                            slh_buffer_free(&jobs[t].buf);
                        }
                        free(jobs);
                        free(tid);
                    }
#undef first
#undef last
                    slamem_host_free(mems);
                    slamem_host_free(boff);
                    free(parts[part].offs);
                }
                t_format += now_s() - tg;
            }
            free(cur);
            cur = nxt;
        }
        free(ranges);
        if (log_limit != 0 && (long)total_queries * strands > log_limit)
            printf(":: ... (%ld more strand blocks matched; set SLAMEM_VERBOSE=1 for a line each)\n",
                   (long)total_queries * strands - log_limit);
    }
    double t_end0 = now_s(), t_end1, t_end2;
    for (i = 0; i < ngpu; i++) slamem_index_free(gpus[i]);
    t_end1 = now_s();
    if (total_queries != 1) /* slamem.c:210-212 (the reference divides by zero when nothing matched) */
        printf(":: Average %d M%cMs found per query sequence (total = %lld, avg size = %d bp)\n",
               (int)(total_matches / total_queries), MATCH_TYPE_CHAR[o.match_type], total_matches,
               (int)(total_matches ? total_sum / total_matches : 0));
    fflush(stdout);
    printf("> Saving M%cMs to <%s> ... ", MATCH_TYPE_CHAR[o.match_type], out_name);
    if (fclose(out) != 0) exit_message("Cannot write output file");
    t_end2 = now_s();
    printf("OK\n");
    if (o.out_arg == -1) free(out_name);
    slh_buffer_free(&buf);
    slh_free_seqset(&ref);
    for (f = 0; f < num_qsets; f++) slh_free_seqset(&qsets[f]);
    free(qsets);
    slh_free_options(&o);
    printf("> Done!\n");
    if (timing)
        fprintf(stderr, "[timing] load %.3f s (index build of %.3f s overlapped), waiting for the GPU (search + transfers, overlapped with formatting) %.3f s, format %.3f s + write %.3f s, index free %.3f s, close %.3f s, host free %.3f s, total %.3f s\n",
                t_load, t_build, t_gpu, t_format - t_write, t_write, t_end1 - t_end0, t_end2 - t_end1, now_s() - t_end2, now_s() - t_start);
    return 0;
}
