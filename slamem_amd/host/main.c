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
 * SLAMEM_BATCH_MB bounds the query characters sent to the GPU per batch (default 256); SLAMEM_OVERLAP_MB (default 256):
 * query files above this size in total are parsed in pieces of SLAMEM_PIECE_MB (128) beside the search, -1 = never;
 * SLAMEM_NO_BIND=1 leaves the host threads where the scheduler puts them (default: on the CPUs local to the GPU);
 * SLAMEM_FULL_TEARDOWN=1 frees every buffer and the index before returning (default: one process that ends with _exit
 * once the results are written -- the command returns when its GPU memory is back, so a job started right behind it
 * finds the device free, and schedulers see the job end when it ends).  SLAMEM_DETACH_TEARDOWN=1 (opt-in): the work runs
 * in a forked worker and the command returns as soon as the results are written, while the worker still gives back its
 * HBM and pinned memory behind the caller's back (0.3-0.4 s for the reference-sized run); never use it under a profiler
 * (the fork would follow the profiler's GPU initialisation).  SLAMEM_WAIT_HBM_S (default 30): how long start-up waits
 * for the HBM the index needs to become free (the tail of such a detached worker, or any other job) before it builds.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <pthread.h>
#include <dlfcn.h>
#include <dirent.h>
#include <sched.h>
#include <ctype.h>
#include <errno.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>

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

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Large query files are parsed in pieces by their own thread while the main thread already searches the first pieces
 * (slh_pieces_*): the FASTA parse (0.2 s for the reference-sized run) no longer stands in front of the search. */
typedef struct {
    char **argv;
    const int *file_args;
    int first_file, num_files, acgt_only, numbering;
    uint32_t min_len;
    long log_limit, piece_bytes;
    slh_seqset *sets;  /* room for every piece */
    int cap;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int ready, done, total_queries;
    int release_early; /* footprint before speed: see main */
    int failed;        /* more pieces than main made room for */
    double seconds;
} loader_t;

static void *loader_run(void *arg) {
    loader_t *ld = (loader_t *)arg;
    double t0 = now_s();
    int f;
    for (f = ld->first_file; f < ld->num_files; f++) {
        slh_pieces *p = slh_pieces_open(ld->argv[ld->file_args[f]], ld->acgt_only, ld->min_len, ld->numbering, ld->log_limit,
                                        ld->piece_bytes, stdout);
        if (!p) continue;
        slh_pieces_release_parsed(p, ld->release_early);
        for (;;) {
            slh_seqset s;
            int n = slh_pieces_next(p, &s);
            if (n <= 0) break;
            pthread_mutex_lock(&ld->mu);
            if (ld->ready < ld->cap) {
                ld->sets[ld->ready++] = s;
                ld->total_queries += n;
                ld->numbering += n;
            } else { /* cannot happen (a piece is at least piece_bytes of its file): never drop reads silently */
                ld->failed = 1;
                slh_free_seqset(&s);
            }
            pthread_cond_broadcast(&ld->cv);
            pthread_mutex_unlock(&ld->mu);
        }
        slh_pieces_close(p);
    }
    pthread_mutex_lock(&ld->mu);
    ld->done = 1;
    ld->seconds = now_s() - t0;
    pthread_cond_broadcast(&ld->cv);
    pthread_mutex_unlock(&ld->mu);
    return NULL;
}

/* While the loader thread still parses query files (and prints its "# NN [name]" lines), the main thread's lines are held
 * back in memory; release_stdout joins the loader, prints the reference's "successfully loaded" line and then the held
 * text, so that stdout reads as if everything had been loaded first (slamem.c:635-651 before :37-218).  Every exit path
 * calls it first. */
static loader_t g_ld;
static pthread_t g_ld_tid;
static int g_ld_started = 0;
static FILE *g_mo = NULL; /* the memory stream, NULL: print directly */
static char *g_mo_buf = NULL;
static size_t g_mo_len = 0;
static int g_num_refs = 0;
#define say(...) fprintf(g_mo ? g_mo : stdout, __VA_ARGS__)
static void release_stdout(void) {
    if (g_ld_started) {
        g_ld_started = 0;
        pthread_join(g_ld_tid, NULL);
        if (g_ld.ready > 0)
            printf("> %d reference%s and %d quer%s successfully loaded\n", g_num_refs, g_num_refs == 1 ? "" : "s", g_ld.total_queries,
                   g_ld.total_queries == 1 ? "y" : "ies");
    }
    if (g_mo) {
        FILE *m = g_mo;
        g_mo = NULL;
        fclose(m);
        if (g_mo_buf && g_mo_len) fwrite(g_mo_buf, 1, g_mo_len, stdout);
        free(g_mo_buf);
        g_mo_buf = NULL;
    }
}


static void exit_message(const char *msg) { /* tools.c:21-25 */
    release_stdout();
    printf("> ERROR: %s\n", msg);
    join_warmup();
    exit(-1);
}

static void gpu_fail(const char *what, int rc) {
    release_stdout();
    printf("\n> ERROR: %s failed: %s (%s)\n", what, slamem_strerror(rc), slamem_last_error_message());
    join_warmup();
    exit(-1);
}

static void gpu_fail_msg(const char *what, int rc, const char *detail) { /* the detail was captured on another thread */
    release_stdout();
    printf("\n> ERROR: %s failed: %s (%s)\n", what, slamem_strerror(rc), detail);
    join_warmup();
    exit(-1);
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
    double t_start, t_end; /* (SLAMEM_TIMING) when the thread worked */
} fmt_job;

/* Text buffers go round: formatter -> writer -> pool -> formatter.  A fresh 20 MB buffer costs its page faults every
 * time (the output of the reference-sized run is 638 MB); a recycled one is already mapped. */
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static slh_buffer g_pool[512];
static int g_pool_n = 0;
static long g_pool_hits = 0, g_pool_misses = 0; /* (SLAMEM_TIMING) */
static double g_fmt_busy = 0, g_fmt_longest = 0, g_fmt_latest_start = 0; /* (SLAMEM_TIMING) formatter threads: CPU time, per-batch maxima */
static void pool_put(slh_buffer *b) {
    pthread_mutex_lock(&g_pool_mu);
    if (b->data && g_pool_n < 512) { b->len = 0; g_pool[g_pool_n++] = *b; b->data = NULL; }
    pthread_mutex_unlock(&g_pool_mu);
    if (b->data) slh_buffer_free(b);
    b->data = NULL; b->len = b->cap = 0;
}
static void pool_get(slh_buffer *b, size_t need) {
    int i, best = -1;
    pthread_mutex_lock(&g_pool_mu);
    for (i = 0; i < g_pool_n; i++)
        if (g_pool[i].cap >= need && (best < 0 || g_pool[i].cap < g_pool[best].cap)) best = i;
    if (best >= 0) { *b = g_pool[best]; g_pool[best] = g_pool[--g_pool_n]; g_pool_hits++; }
    else g_pool_misses++;
    pthread_mutex_unlock(&g_pool_mu);
}

static void *fmt_run(void *arg) {
    fmt_job *j = (fmt_job *)arg;
    uint64_t b;
    j->t_start = now_s();
    /* about 36 characters per MEM line and a header per block: reserve once instead of doubling on the way */
    const size_t need = (size_t)(j->boff[j->b1] - j->boff[j->b0]) * 36 + (size_t)(j->b1 - j->b0) * 48 + 4096;
    pool_get(&j->buf, need);
    (void)slh_buffer_reserve(&j->buf, need);
    for (b = j->b0; b < j->b1; b++) {
        int i = j->first_rec + (int)(b / (uint64_t)j->strands), s = (int)(b % (uint64_t)j->strands);
        uint64_t cnt = j->boff[b + 1] - j->boff[b], sum = 0;
        if (slh_format_block(&j->buf, j->q->recs[i].name, s, (const uint32_t *)(j->mems + j->boff[b]), cnt, j->ref->recs,
                             j->ref->merged_start, j->ref->num, &sum)) { j->failed = 1; return NULL; }
        j->matches += (long long)cnt;
        j->sum += (long long)sum;
    }
    j->t_end = now_s();
    return NULL;
}

/* the formatter threads of a batch take its chunks (ranges of strand blocks) from a shared counter: equal shares per
 * thread left the batch waiting for its slowest thread (measured: twice the mean, on a box whose CPU share is smaller
 * than the number of threads) */
typedef struct {
    fmt_job *jobs;
    int njobs;
    int next; /* atomic */
} fmt_queue;

static void *fmt_worker(void *arg) {
    fmt_queue *q = (fmt_queue *)arg;
    for (;;) {
        int c = __atomic_fetch_add(&q->next, 1, __ATOMIC_RELAXED);
        if (c >= q->njobs) return NULL;
        fmt_run(&q->jobs[c]);
    }
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

/* A job that starts right behind another one (a detached worker still giving back its arena, any other tenant of the
 * GPU) finds less HBM free than it will have a moment later: wait, with a bound, until the full layout's build peak
 * fits -- or the compact layout's when the device could never hold the full one -- instead of failing or silently
 * taking the slower layout.  The reference frees before it reports (slamem.c:208-216); this is the other half. */
static void wait_for_hbm(const build_job *b) {
    uint64_t arena = 0, peak_full = 0, peak_compact = 0, free_b = 0, total_b = 0, want;
    const char *env = getenv("SLAMEM_WAIT_HBM_S");
    double limit = env ? atof(env) : 30.0, t0 = now_s();
    int told = 0;
    if (limit <= 0) return;
    if (slamem_index_build_bytes(b->n, SLAMEM_LAYOUT_FULL, &arena, &peak_full) != SLAMEM_OK) return;
    if (slamem_index_build_bytes(b->n, SLAMEM_LAYOUT_COMPACT, &arena, &peak_compact) != SLAMEM_OK) return;
    for (;;) {
        if (slamem_device_mem_info(b->device, &free_b, &total_b) != SLAMEM_OK) return;
        want = (peak_full + (1ull << 30) <= total_b ? peak_full : peak_compact) + (256ull << 20);
        if (free_b >= want || want > total_b || now_s() - t0 > limit) break;
        if (!told) {
            fprintf(stderr, "> Waiting for HBM on GPU %d: %.1f GB free, %.1f GB needed to build the index (another job is still releasing its memory?)\n",
                    b->device, (double)free_b / 1e9, (double)want / 1e9);
            told = 1;
        }
        usleep(50000);
    }
    if (told) fprintf(stderr, "> Waited %.2f s for HBM (%.1f GB free now)\n", now_s() - t0, (double)free_b / 1e9);
}

static void *build_run(void *arg) {
    build_job *b = (build_job *)arg;
    double t0;
    wait_for_hbm(b);
    t0 = now_s();
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

/* The output file is written by its own thread: the main thread hands over formatted buffers in order and goes on
 * formatting the next ones (GPU(b+1), format(b) and write(b-1) overlap; the reference does everything in one loop,
 * slamem.c:90-207). */
typedef struct wnode {
    slh_buffer buf;
    struct wnode *next;
} wnode;
typedef struct {
    FILE *out;
    pthread_t tid;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    wnode *head, *tail;
    size_t queued; /* bytes waiting */
    int done, failed, started;
    double seconds;
} writer_t;

static void *writer_run(void *arg) {
    writer_t *w = (writer_t *)arg;
    for (;;) {
        wnode *n;
        double t0;
        pthread_mutex_lock(&w->mu);
        while (!w->head && !w->done) pthread_cond_wait(&w->cv, &w->mu);
        n = w->head;
        if (n) { w->head = n->next; if (!w->head) w->tail = NULL; }
        pthread_mutex_unlock(&w->mu);
        if (!n) return NULL;
        t0 = now_s();
        if (n->buf.len && !w->failed && fwrite(n->buf.data, 1, n->buf.len, w->out) != n->buf.len) w->failed = 1;
        w->seconds += now_s() - t0;
        pthread_mutex_lock(&w->mu);
        w->queued -= n->buf.len;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        pool_put(&n->buf);
        free(n);
    }
}

/* takes ownership of *buf (and leaves it empty); waits while more than 1 GiB of text is queued */
static int writer_push(writer_t *w, slh_buffer *buf) {
    wnode *n;
    if (!buf->len) return 0;
    n = (wnode *)calloc(1, sizeof(wnode));
    if (!n) return -1;
    n->buf = *buf;
    buf->data = NULL; buf->len = buf->cap = 0;
    if (!w->started) { /* no thread: write here */
        int bad = fwrite(n->buf.data, 1, n->buf.len, w->out) != n->buf.len;
        if (bad) w->failed = 1;
        slh_buffer_free(&n->buf);
        free(n);
        return 0;
    }
    pthread_mutex_lock(&w->mu);
    while (w->queued > ((size_t)1 << 30)) pthread_cond_wait(&w->cv, &w->mu);
    if (w->tail) w->tail->next = n; else w->head = n;
    w->tail = n;
    w->queued += n->buf.len;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    return 0;
}

static void writer_finish(writer_t *w) {
    if (!w->started) return;
    pthread_mutex_lock(&w->mu);
    w->done = 1;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    pthread_join(w->tid, NULL);
    w->started = 0;
}

/* every exit after the search has started goes through here: batches in flight finish first (slamem_stream_destroy
 * waits for them), the writer drains -- never leave the process with kernels running or a thread writing */
static slamem_stream *g_streams[16];
static int g_nstreams = 0;
static writer_t g_writer;
static void shutdown_pipeline(void) {
    int g;
    for (g = 0; g < g_nstreams; g++) { slamem_stream_destroy(g_streams[g]); g_streams[g] = NULL; }
    g_nstreams = 0;
    writer_finish(&g_writer);
}
static void pipeline_fail(const char *msg) {
    shutdown_pipeline();
    exit_message(msg);
}

/* A piece of the query files whose batches are all formatted can be released by a thread of its own while the main thread
 * goes on.  Measured on the reference-sized run (1.5 GB of reads): it takes 0.10 s off the end of the process and adds
 * 0.10 s to the formatting beside it (the address-space work stalls the formatter threads), so it is only done when the
 * footprint matters: query files above SLAMEM_RELEASE_EARLY_GB (default 16) in total. */
static pthread_t *g_reap_tid = NULL;
static int g_reap_n = 0;
static void *reap_run(void *arg) {
    slh_free_seqset((slh_seqset *)arg);
    free(arg);
    return NULL;
}
static void reap_set(slh_seqset *s) {
    slh_seqset *copy = (slh_seqset *)malloc(sizeof(slh_seqset));
    if (!copy || !g_reap_tid) { free(copy); return; } /* stays allocated until the process ends */
    *copy = *s;
    if (pthread_create(&g_reap_tid[g_reap_n], NULL, reap_run, copy) != 0) { free(copy); return; }
    g_reap_n++;
    memset(s, 0, sizeof(*s));
}

/* The host threads (FASTA parsing, formatting, the copies behind uploads from ordinary memory) work on memory that the GPU's
 * NUMA node holds or receives: on a two-socket box they ran 1.5 times longer when the scheduler spread them over both
 * sockets (tools/cli_numa_probe.sh: 0.47-0.56 s -> 0.42 s for the reference-sized run).  Once the runtime is up, every thread
 * of the process is confined to the CPUs that are local to the GPU -- if the process may use CPUs of more than one node,
 * and unless SLAMEM_NO_BIND is set.  Threads created later inherit the mask. */
static void bind_to_gpu_node(int device) {
    char bdf[64], path[160], list[4096];
    cpu_set_t local, allowed, both;
    FILE *f;
    DIR *d;
    struct dirent *e;
    const char *p;
    size_t i;
    if (getenv("SLAMEM_NO_BIND") != NULL) return;
    if (slamem_device_pci_bus_id(device, bdf, (int)sizeof(bdf)) != SLAMEM_OK) return;
    for (i = 0; bdf[i]; i++) bdf[i] = (char)tolower((unsigned char)bdf[i]);
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    f = fopen(path, "r");
    if (!f) return;
    if (!fgets(list, (int)sizeof(list), f)) { fclose(f); return; }
    fclose(f);
    CPU_ZERO(&local);
    for (p = list; *p;) { /* "0-63,128-191" */
        char *end;
        long a, b;
        if (!isdigit((unsigned char)*p)) { p++; continue; }
        a = strtol(p, &end, 10);
        b = a;
        if (*end == '-') b = strtol(end + 1, &end, 10);
        for (; a <= b && a < CPU_SETSIZE; a++) CPU_SET((int)a, &local);
        p = end;
    }
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    CPU_AND(&both, &local, &allowed);
    if (CPU_COUNT(&both) == 0 || CPU_COUNT(&both) == CPU_COUNT(&allowed)) return; /* nothing local allowed / already local */
    {   /* not onto a handful of CPUs: the loader and formatter threads need room (a cpuset that leaves one or two local
           CPUs would serialise them), and a mask the user narrowed on purpose is left alone */
        int want = slh_thread_count(), half = CPU_COUNT(&allowed) / 2;
        long online = sysconf(_SC_NPROCESSORS_ONLN);
        if (want > half) want = half;
        if (CPU_COUNT(&both) < want) return;
        if (online > 0 && CPU_COUNT(&allowed) < (int)online && getenv("SLAMEM_BIND") == NULL) return;
    }
    d = opendir("/proc/self/task");
    if (!d) { (void)sched_setaffinity(0, sizeof(both), &both); return; }
    while ((e = readdir(d)) != NULL)
        if (isdigit((unsigned char)e->d_name[0])) (void)sched_setaffinity((pid_t)atol(e->d_name), sizeof(both), &both);
    closedir(d);
}

static int g_warm_ok = 0;
static void *warmup_run(void *arg) { /* HIP runtime + context start-up, hidden behind the parsing of the reference file */
    g_warm_ok = slamem_device_warmup(*(int *)arg) == SLAMEM_OK;
    return NULL;
}

/* The command returns when its results are complete, not when the kernel has taken back the gigabytes behind them.
 * The work runs in a child process (forked before any thread or GPU call exists); the parent only waits for one byte
 * that the child sends once the output file and stdout are written, and leaves with status 0 then.  What follows in
 * the child -- the end of a process that holds the index in HBM, pinned buffers and the queries, 0.35 s for the
 * reference-sized run -- happens behind the caller's back.  A child that ends without that byte (any error exit, a
 * signal) is waited for and its status passed on.  SLAMEM_FOREGROUND=1: one process, as before. */
static int g_done_fd = -1;
static pid_t g_child = 0;
static void forward_signal(int sig) {
    if (g_child > 0) kill(g_child, sig);
}
/* has the process that forked us gone already?  (PR_SET_PDEATHSIG only covers a death AFTER the prctl call.)  Compared
 * with the pid the front had before the fork: getppid() == 1 says nothing when the front itself is pid 1 of a container,
 * and never happens under a subreaper. */
static int front_is_gone(pid_t front) { return getppid() != front; }
static void split_off_worker(void) {
    int pfd[2];
    pid_t pid, front = getpid();
    if (getenv("SLAMEM_DETACH_TEARDOWN") == NULL || getenv("SLAMEM_FOREGROUND") != NULL || pipe(pfd) != 0) return;
    fflush(stdout);
    fflush(stderr);
    pid = fork();
    if (pid < 0) { close(pfd[0]); close(pfd[1]); return; }
    if (pid == 0) { /* the worker */
        close(pfd[0]);
        g_done_fd = pfd[1];
        prctl(PR_SET_PDEATHSIG, SIGTERM); /* no worker without its front */
        if (front_is_gone(front)) _exit(255);
        return;
    }
    close(pfd[1]);
    g_child = pid;
    signal(SIGINT, forward_signal);
    signal(SIGTERM, forward_signal);
    signal(SIGHUP, forward_signal);
    for (;;) {
        char c;
        ssize_t n = read(pfd[0], &c, 1);
        int st;
        if (n == 1) _exit(0); /* everything is written */
        if (n < 0 && errno == EINTR) continue;
        while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
        _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0));
    }
}
static void leave_now(void) { /* results complete: tell the front, then end without returning memory piece by piece */
    const char *linger = getenv("SLAMEM_TEST_LINGER_MS"); /* tests: hold the GPU memory this long after "done" */
    fflush(stdout);
    fflush(stderr);
    if (g_done_fd >= 0) {
        char c = 0;
        close(1);
        close(2);
        if (linger && atoi(linger) > 0) prctl(PR_SET_PDEATHSIG, 0); /* (the front's exit would end the worker at once) */
        if (write(g_done_fd, &c, 1) != 1) _exit(255);
        if (linger && atoi(linger) > 0) usleep((useconds_t)atoi(linger) * 1000u);
    }
    _exit(0);
}

/* "-v <mems_file>" (slamem.c:630-655): the sequences are loaded for their names and sizes only, then the picture is drawn */
static int image_tool(int argc, char **argv, slh_options *o, long log_limit) {
    slh_seqset ref, *sets;
    slh_record *seqs;
    int f, k, numbering = 1, loaded = 0, queries = 0, at = 0, status;
    (void)argc;
    memset(&ref, 0, sizeof(ref));
    sets = (slh_seqset *)calloc((size_t)o->num_files + 1, sizeof(slh_seqset));
    if (!sets) exit_message("Out of memory");
    for (f = 0; f < o->num_files; f++) {
        const char *path = argv[o->file_args[f]];
        int n = slh_load_file(path, loaded == 0, o->no_ns, (uint32_t)o->min_seq_len, loaded == 0 ? o->ref_name : NULL, numbering,
                              log_limit, loaded == 0 ? &ref : &sets[loaded], stdout);
        if (n != 0) { loaded++; numbering += n; if (loaded > 1) queries += n; }
        if (loaded == 0) exit_message("No valid sequences found in reference file");
    }
    if (loaded == 1) exit_message("No query files provided");
    if (queries == 0) exit_message("No valid query sequences found");
    printf("> %d reference%s and %d quer%s successfully loaded\n", ref.num, ref.num == 1 ? "" : "s", queries, queries == 1 ? "y" : "ies");
    seqs = (slh_record *)calloc((size_t)queries + 1, sizeof(slh_record));
    if (!seqs) exit_message("Out of memory");
    seqs[at].name = ref.recs[0].name;
    seqs[at++].size = (uint32_t)ref.total; /* one record: its size; several: refused by the tool */
    for (f = 1; f < loaded; f++)
        for (k = 0; k < sets[f].num; k++) seqs[at++] = sets[f].recs[k];
    status = slh_mem_map_image(argv[o->image_arg], seqs, at, ref.num, stdout);
    fflush(stdout);
    free(seqs);
    for (f = 1; f < loaded; f++) slh_free_seqset(&sets[f]);
    free(sets);
    slh_free_seqset(&ref);
    slh_free_options(o);
    return status;
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
    printf("Extra:\n");
    printf("\t-v\tgenerate MEMs map image from this MEMs file\n");
    printf("Example:\n");
    printf("\t%s -b -l 10 ./ref.fna ./query.fna\n", prog);
    printf("\t%s -v ./ref-mems.txt ./ref.fna ./query.fna\n", prog);
}

int main(int argc, char **argv) {
    slh_options o;
    slh_seqset ref, *qsets;
    int i, f, total_queries = 0, device = 0, numbering = 1;
    long log_limit = 100;
    uint64_t batch_bytes = 256ull << 20; /* query characters per batch and GPU */
    long piece_bytes = 128L << 20;       /* bytes of a query file per piece of the loader thread */
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
    if ((env = getenv("SLAMEM_DEVICE")) != NULL) device = atoi(env);
    if ((env = getenv("SLAMEM_VERBOSE")) != NULL && atoi(env) != 0) log_limit = 0;
    if (o.image_arg != -1) return image_tool(argc, argv, &o, log_limit); /* slamem.c:652-655; host only, before any GPU call */
    if ((env = getenv("SLAMEM_BATCH_MB")) != NULL && atoll(env) > 0) batch_bytes = (uint64_t)atoll(env) << 20;

    if (getenv("SLAMEM_FULL_TEARDOWN") == NULL) split_off_worker();
    {
        static int warm_device;
        warm_device = device;
        g_warm_started = pthread_create(&g_warm_tid, NULL, warmup_run, &warm_device) == 0;
    }
    double t_start = now_s(), t_load = 0, t_build = 0, t_gpu = 0, t_format = 0, t_write = 0, t_wait_load = 0;
    int timing = getenv("SLAMEM_TIMING") != NULL;
    /* load everything (slamem.c:635-651).  Query files above SLAMEM_OVERLAP_MB (default 256) in total are parsed in pieces by
       a loader thread while the search already runs; stdout keeps the reference's order: what the main thread prints meanwhile
       is held back until the "successfully loaded" line can be printed */
    memset(&ref, 0, sizeof(ref));
    memset(&bj, 0, sizeof(bj));
    loader_t *const ld = &g_ld;
    int overlap = 0, ref_file = -1;
    {
        uint64_t qbytes_total = 0, pieces_cap = 0;
        long thr_mb = 256;
        if ((env = getenv("SLAMEM_OVERLAP_MB")) != NULL) thr_mb = atol(env);
        if ((env = getenv("SLAMEM_PIECE_MB")) != NULL && atol(env) > 0) piece_bytes = atol(env) << 20;
        for (f = 1; f < o.num_files; f++) {
            FILE *qf = fopen(argv[o.file_args[f]], "rb");
            if (qf) {
                long sz;
                fseek(qf, 0L, SEEK_END);
                sz = ftell(qf);
                fclose(qf);
                if (sz > 0) { qbytes_total += (uint64_t)sz; pieces_cap += (uint64_t)sz / (uint64_t)piece_bytes + 2; }
            }
        }
        overlap = thr_mb >= 0 && qbytes_total > ((uint64_t)thr_mb << 20);
        memset(ld, 0, sizeof(*ld));
        ld->release_early = qbytes_total > ((uint64_t)((env = getenv("SLAMEM_RELEASE_EARLY_GB")) != NULL ? atol(env) : 16) << 30);
        ld->cap = (int)(overlap ? pieces_cap + (uint64_t)o.num_files : (uint64_t)o.num_files);
        qsets = (slh_seqset *)calloc((size_t)ld->cap + 1, sizeof(slh_seqset));
        g_reap_tid = (pthread_t *)calloc((size_t)ld->cap + 1, sizeof(pthread_t));
        if (!qsets || !g_reap_tid) exit_message("Out of memory");
        ld->sets = qsets;
        pthread_mutex_init(&ld->mu, NULL);
        pthread_cond_init(&ld->cv, NULL);
    }
    for (f = 0; f < o.num_files; f++) {
        const char *path = argv[o.file_args[f]];
        if (ref_file < 0) {
            int n = slh_load_file(path, 1, o.no_ns, (uint32_t)o.min_seq_len, o.ref_name, numbering, log_limit, &ref, stdout);
            if (n == 0) exit_message("No valid sequences found in reference file");
            ref_file = f;
            numbering += n;
            g_num_refs = ref.num;
            o.file_args[0] = o.file_args[f]; /* remember which argument was the reference */
            bj.text = ref.chars;
            bj.n = (uint32_t)ref.total;
            bj.device = device;
            build_async = pthread_create(&build_tid, NULL, build_run, &bj) == 0;
            if (overlap) break;
        } else {
            int n = slh_load_file(path, 0, o.no_ns, (uint32_t)o.min_seq_len, NULL, numbering, log_limit, &qsets[ld->ready], stdout);
            if (n != 0) { numbering += n; ld->total_queries += n; ld->ready++; }
        }
    }
    if (overlap) {
        ld->argv = argv; ld->file_args = o.file_args; ld->first_file = ref_file + 1; ld->num_files = o.num_files;
        ld->acgt_only = o.no_ns; ld->min_len = (uint32_t)o.min_seq_len; ld->numbering = numbering; ld->log_limit = log_limit;
        ld->piece_bytes = piece_bytes;
        fflush(stdout);
        g_mo = open_memstream(&g_mo_buf, &g_mo_len);
        g_ld_started = g_mo != NULL && pthread_create(&g_ld_tid, NULL, loader_run, ld) == 0;
        if (!g_ld_started) { /* no thread: load here, as for small inputs */
            if (g_mo) release_stdout();
            overlap = 0;
            loader_run(ld);
        }
    } else {
        ld->done = 1;
    }
    t_load = now_s() - t_start;
    double t_join0 = now_s();
    if (build_async) pthread_join(build_tid, NULL); /* before any exit: never leave the process with a build in flight */
    else build_run(&bj);
    join_warmup();
    if (g_warm_ok) bind_to_gpu_node(device); /* from the main thread, before the worker threads below exist: they inherit the mask */
    double t_join = now_s() - t_join0, t_streams = 0;
    if (!overlap) {
        if (ld->ready == 0) exit_message("No query files provided"); /* slamem.c:648 */
        printf("> %d reference%s and %d quer%s successfully loaded\n", ref.num, ref.num == 1 ? "" : "s", ld->total_queries,
               ld->total_queries == 1 ? "y" : "ies");
    }
    if (o.min_mem_len < 1) exit_message("Minimum match length must be at least 1");

    if (o.out_arg == -1) out_name = slh_append_to_basename(argv[o.file_args[0]], "-mems.txt");
    else out_name = argv[o.out_arg];

    /* GetMatches (slamem.c:37-218) */
    say("> Using options: minimum M%cM length = %d ; strand = %s\n", MATCH_TYPE_CHAR[o.match_type], o.min_mem_len,
           o.both_strands == 0 ? "forward only" : "forward + reverse");
    out = fopen(out_name, "w");
    if (!out) {
        release_stdout();
        printf("\n> ERROR: Cannot create output file <%s>\n", out_name);
        join_warmup();
        exit(-1);
    }
    say("> Building index for reference sequence");
    if (ref.num == 1) say(" \"%s\"", ref.recs[0].name);
    else say("s");
    say(" (%u Mbp) ...\n", (unsigned)(ref.total / 1000000U));
    if (!g_mo) fflush(stdout);
    t0 = now_s();
    idx = bj.idx;
    if (bj.rc != SLAMEM_OK) gpu_fail_msg("index construction on the GPU", bj.rc, bj.err);
    {
        slamem_index_info info = bj.info;
        slamem_timings tm = bj.tm;
        say("> Suffix sort + BWT + LCP + parent links on GPU %d ... OK (%.3f s, overlapped with the loading of the queries; device %.1f ms: sort %.1f, BWT %.1f, LCP %.1f, links %.1f; %u doubling rounds)\n",
               device, bj.seconds, tm.build_total_ms, tm.build_sort_ms, tm.build_bwt_ms, tm.build_lcp_ms, tm.build_links_ms,
               info.sort_rounds);
        say(":: Index size = %.1f MB in HBM (%s layout)\n", (double)info.arena_bytes / 1e6,
            info.layout == SLAMEM_LAYOUT_COMPACT ? "compact" : "full");
        {   /* the statistics lines of BuildSampledLCPArray (lcparray.c:709-711, 999-1000), from the per-row records */
            slamem_sslcp_stats st = bj.st;
            unsigned bwt_len = info.bwt_size;
            if (bj.rc_stats != SLAMEM_OK) gpu_fail_msg("LCP sampling statistics", bj.rc_stats, bj.err);
            say(":: %.2lf%% samples (%u of %u)\n", ((double)st.num_samples / (double)bwt_len) * 100.0, (unsigned)st.num_samples, bwt_len);
            say(":: %.2lf%% oversized samples (%d of %u)\n", ((double)st.num_oversized_lcp / (double)st.num_samples) * 100.0,
                   (int)st.num_oversized_lcp, (unsigned)st.num_samples);
            say(":: Average LCP value = %d (max=%lld)\n", (int)(st.sum_lcp / (long long)bwt_len), (long long)st.max_lcp);
            say(":: %.2lf%% oversized values (%d of %u)\n", ((double)st.num_oversized_links / (double)st.num_samples) * 100.0,
                   (int)st.num_oversized_links, (unsigned)st.num_samples);
            say(":: Average SV distance = %.2lf (max=%lld)\n", (double)st.sum_link_distance / (double)st.num_samples,
                   (long long)st.max_link_distance);
        }
    }
    /* SLAMEM_GPUS=N: replicate the index to N GPUs (device, device+1, ...) with one RCCL broadcast of its arena over
       xGMI; every batch is then split by bases into N contiguous shares, one per GPU (no data-path collective). */
    gpus[0] = idx;
    int logical = 0; /* SLAMEM_LOGICAL_GPUS=N (self-test): N "GPUs" that are all this device, each with its own copy of the index */
    if ((env = getenv("SLAMEM_GPUS")) != NULL) {
        int avail = 0;
        slamem_device_count(&avail);
        ngpu = atoi(env) <= 0 ? avail - device : atoi(env);
        if (ngpu < 1) ngpu = 1;
        if (ngpu > 16) ngpu = 16;
        if (device + ngpu > avail) exit_message("SLAMEM_GPUS asks for more GPUs than are visible");
    } else if ((env = getenv("SLAMEM_LOGICAL_GPUS")) != NULL && atoi(env) > 1) {
        ngpu = atoi(env) > 16 ? 16 : atoi(env);
        logical = 1;
    }
    if (ngpu > 1 || getenv("SLAMEM_REPLICATE_SELFTEST") != NULL) {
        int devs[16], g;
        double tr = now_s();
        for (g = 0; g < ngpu; g++) devs[g] = logical ? device : device + g;
        {   /* libslamem_rccl.so (and RCCL behind it, hundreds of MB of code) is loaded only when it is needed: a
               one-GPU run never pays for it at start-up */
            typedef int (*replicate_fn)(const slamem_index *, const int *, int, int, slamem_index **);
            void *h = dlopen("libslamem_rccl.so", RTLD_NOW | RTLD_GLOBAL);
            replicate_fn fn = h ? (replicate_fn)dlsym(h, "slamem_index_replicate") : NULL;
            if (!fn) {
                release_stdout();
                printf("\n> ERROR: cannot load libslamem_rccl.so (%s)\n", dlerror());
                join_warmup();
                exit(-1);
            }
            if (logical) { /* one one-rank broadcast into a fresh arena per logical GPU: the scheduling below is the real N-GPU one */
                for (g = 1, rc = SLAMEM_OK; g < ngpu && rc == SLAMEM_OK; g++) rc = fn(idx, devs, 1, 1, &gpus[g]);
            } else {
                rc = fn(idx, devs, ngpu, ngpu == 1, gpus);
            }
        }
        if (rc != SLAMEM_OK) gpu_fail("index replication over RCCL", rc);
        if (ngpu == 1) { slamem_index_free(idx); idx = gpus[0]; } /* self-test: search on the broadcast copy */
        say("> Index replicated to %d %sGPU%s by RCCL broadcast ... OK (%.3f s)\n", ngpu, logical ? "logical " : "",
            ngpu == 1 ? " (self-test copy)" : "s", now_s() - tr);
    }
    t_build = bj.seconds + (now_s() - t0);
    free(ref.chars); /* the reference frees the text here too (slamem.c:75-77) */
    ref.chars = NULL;
    say("> Matching query sequences against index ...\n");
    if (!g_mo) fflush(stdout);

    {
        int strands = o.both_strands ? 2 : 1;
        long printed = 0;
        /* the list of batches (records [first,last) of one query file each), then a pipeline over it: the GPUs search
           batches b+1.. (slamem_stream_*: upload, search and download of neighbouring batches overlap), the main thread
           formats batch b with all host threads, the writer thread writes batch b-1 */
        typedef struct { int f, first, last; } batch_range;
        batch_range *ranges = NULL;
        size_t nranges = 0, cap_ranges = 0, bi, submitted = 0;
        uint64_t max_chars = 1;
        uint32_t max_recs = 1;
        int inflight[16], g;
        const int slots = 6; /* upload, prepare, search and download of four batches beside the one being formatted */
        int sets_seen = 0, sets_reaped = 0, sets_ready = ld->ready; /* !overlap: everything is there */
#define ADD_RANGES_OF_NEW_SETS()                                                                                              \
        for (; sets_seen < sets_ready; sets_seen++) {                                                                          \
            slh_seqset *q = &qsets[sets_seen];                                                                                 \
            int first = 0;                                                                                                     \
            while (first < q->num) {                                                                                           \
                int last = first;                                                                                              \
                uint64_t base = q->offsets[first];                                                                             \
                /* the first batches are short, so that the search starts early */                                             \
                uint64_t limit = nranges < (size_t)ngpu ? batch_bytes / 4 : nranges < 2 * (size_t)ngpu ? batch_bytes / 2 : batch_bytes; \
                while (last < q->num && (last == first || q->offsets[last + 1] - base <= limit)) last++;                       \
                if (nranges == cap_ranges) {                                                                                   \
                    cap_ranges = cap_ranges ? cap_ranges * 2 : 16;                                                             \
                    ranges = (batch_range *)realloc(ranges, cap_ranges * sizeof(batch_range));                                 \
                    if (!ranges) pipeline_fail("Out of memory");                                                               \
                }                                                                                                              \
                ranges[nranges].f = sets_seen; ranges[nranges].first = first; ranges[nranges].last = last;                     \
                if (q->offsets[last] - base > max_chars) max_chars = q->offsets[last] - base;                                  \
                if ((uint32_t)(last - first) > max_recs) max_recs = (uint32_t)(last - first);                                  \
                nranges++;                                                                                                     \
                first = last;                                                                                                  \
            }                                                                                                                  \
        }
        ADD_RANGES_OF_NEW_SETS();
        if (overlap) { /* the batches are not known yet: reserve for a piece's worth, the slots grow on demand */
            uint64_t guess = batch_bytes < (uint64_t)piece_bytes ? batch_bytes : (uint64_t)piece_bytes;
            if (max_chars < guess) max_chars = guess;
            if (max_recs < max_chars / 64) max_recs = (uint32_t)(max_chars / 64);
        }
        memset(&g_writer, 0, sizeof(g_writer));
        g_writer.out = out;
        pthread_mutex_init(&g_writer.mu, NULL);
        pthread_cond_init(&g_writer.cv, NULL);
        g_writer.started = pthread_create(&g_writer.tid, NULL, writer_run, &g_writer) == 0;
        double ts0 = now_s();
        for (g = 0; g < ngpu && (nranges || overlap); g++) { /* batch b is searched on GPU b mod ngpu: no data-path collective */
            rc = slamem_stream_create(gpus[g], slots, max_chars, max_recs, o.both_strands, o.match_type == 1 ? 1 : 0, &g_streams[g]);
            if (rc != SLAMEM_OK) { shutdown_pipeline(); gpu_fail("setting up the search pipeline", rc); }
            g_nstreams = g + 1;
            inflight[g] = 0;
        }
        t_streams = now_s() - ts0;
        for (bi = 0;; bi++) {
            if (overlap) { /* take in the pieces parsed meanwhile; wait for one only when there is nothing else to do */
                double tw = now_s();
                pthread_mutex_lock(&ld->mu);
                while (bi >= nranges && sets_seen == ld->ready && !ld->done) pthread_cond_wait(&ld->cv, &ld->mu);
                sets_ready = ld->ready;
                pthread_mutex_unlock(&ld->mu);
                ADD_RANGES_OF_NEW_SETS();
                t_wait_load += now_s() - tw;
                if (bi >= nranges) { /* nothing new: either the loader is done or a piece without batches came in */
                    int finished;
                    pthread_mutex_lock(&ld->mu);
                    finished = ld->done && sets_seen == ld->ready;
                    pthread_mutex_unlock(&ld->mu);
                    if (finished) break;
                    bi--;
                    continue;
                }
            } else if (bi >= nranges) break;
            slh_seqset *q = &qsets[ranges[bi].f];
            const int first = ranges[bi].first, last = ranges[bi].last;
            for (; ld->release_early && sets_reaped < ranges[bi].f; sets_reaped++) /* every batch of the earlier sets is formatted */
                if (qsets[sets_reaped].chars) reap_set(&qsets[sets_reaped]);
            const slamem_mem *mems = NULL;
            const uint64_t *boff = NULL;
            uint64_t total = 0;
            double tg = now_s();
            /* keep every GPU's pipeline full: slots - 1 batches in flight beside the result being formatted */
            while (submitted < nranges && inflight[submitted % (size_t)ngpu] < slots - 1) {
                slh_seqset *qs = &qsets[ranges[submitted].f];
                rc = slamem_stream_submit(g_streams[submitted % (size_t)ngpu], qs->chars, qs->offsets + ranges[submitted].first,
                                          (uint32_t)(ranges[submitted].last - ranges[submitted].first), (uint32_t)o.min_mem_len);
                if (rc != SLAMEM_OK) { shutdown_pipeline(); gpu_fail("MEM search on the GPU", rc); }
                inflight[submitted % (size_t)ngpu]++;
                submitted++;
            }
            rc = slamem_stream_next(g_streams[bi % (size_t)ngpu], &mems, &boff, &total, NULL, NULL);
            if (rc != SLAMEM_OK) {
                char detail[512];
                snprintf(detail, sizeof(detail), "%s", slamem_last_error_message());
                shutdown_pipeline();
                release_stdout();
                printf("\n> ERROR: MEM search on GPU %d failed: %s (%s)\n", device + (int)(bi % (size_t)ngpu), slamem_strerror(rc), detail);
                join_warmup();
                exit(-1);
            }
            inflight[bi % (size_t)ngpu]--;
            t_gpu += now_s() - tg; /* time the main thread waited for the GPUs */
            tg = now_s();
            {
                /* the first strand blocks get their ':: "name" ....' line (slamem.c:97,101,203) and are formatted here;
                   the rest of the batch is formatted by all host threads and handed to the writer in order */
                uint64_t nblk = (uint64_t)(last - first) * strands, bseq = 0, b;
                int nthr = slh_thread_count(), t;
                if (log_limit == 0) bseq = nblk;
                else if (printed < log_limit) bseq = (uint64_t)(log_limit - printed) < nblk ? (uint64_t)(log_limit - printed) : nblk;
                for (b = 0; b < bseq; b++) {
                    int ri = first + (int)(b / strands), sidx = (int)(b % strands), d, dots;
                    uint64_t cnt = boff[b + 1] - boff[b], sum = 0;
                    if (slh_format_block(&buf, q->recs[ri].name, sidx, (const uint32_t *)(mems + boff[b]), cnt, ref.recs,
                                         ref.merged_start, ref.num, &sum))
                        pipeline_fail("Out of memory");
                    total_matches += (long long)cnt;
                    total_sum += (long long)sum;
                    dots = slh_progress_dots(q->recs[ri].size);
                    say(":: \"%s%s\" ", q->recs[ri].name, sidx ? " Reverse" : "");
                    for (d = 0; d < dots; d++) fputc('.', g_mo ? g_mo : stdout);
                    say(" (%d M%cMs ; avg size = %d bp)\n", (int)cnt, MATCH_TYPE_CHAR[o.match_type], (int)(cnt ? sum / cnt : 0));
                    printed++;
                }
                if (writer_push(&g_writer, &buf)) pipeline_fail("Out of memory");
                if (bseq < nblk) {
                    fmt_job *jobs;
                    pthread_t *tid;
                    fmt_queue fq;
                    uint64_t per;
                    int njobs;
                    double fmt_t0 = now_s(), fmt_longest = 0, fmt_latest_start = 0;
                    if ((nblk - bseq) < 4096 || nthr < 1) nthr = 1;
                    /* chunks of 32 k strand blocks, at least one per thread */
                    njobs = (int)((nblk - bseq + 32767) / 32768);
                    if (njobs < nthr) njobs = nthr;
                    jobs = (fmt_job *)calloc((size_t)njobs, sizeof(fmt_job));
                    tid = (pthread_t *)calloc((size_t)nthr, sizeof(pthread_t));
                    if (!jobs || !tid) pipeline_fail("Out of memory");
                    per = (nblk - bseq + (uint64_t)njobs - 1) / (uint64_t)njobs;
                    for (t = 0; t < njobs; t++) {
                        jobs[t].q = q; jobs[t].ref = &ref; jobs[t].mems = mems; jobs[t].boff = boff;
                        jobs[t].first_rec = first; jobs[t].strands = strands;
                        jobs[t].b0 = bseq + per * (uint64_t)t < nblk ? bseq + per * (uint64_t)t : nblk;
                        jobs[t].b1 = jobs[t].b0 + per < nblk ? jobs[t].b0 + per : nblk;
                    }
                    fq.jobs = jobs; fq.njobs = njobs; fq.next = 0;
                    for (t = 0; t + 1 < nthr; t++)
                        if (pthread_create(&tid[t], NULL, fmt_worker, &fq) != 0) tid[t] = 0;
                    fmt_worker(&fq); /* the main thread takes chunks too */
                    for (t = 0; t + 1 < nthr; t++)
                        if (tid[t]) pthread_join(tid[t], NULL);
                    for (t = 0; t < njobs; t++) {
                        if (jobs[t].failed) pipeline_fail("Out of memory");
                        g_fmt_busy += jobs[t].t_end - jobs[t].t_start;
                        if (jobs[t].t_end - jobs[t].t_start > fmt_longest) fmt_longest = jobs[t].t_end - jobs[t].t_start;
                        if (jobs[t].t_start - fmt_t0 > fmt_latest_start) fmt_latest_start = jobs[t].t_start - fmt_t0;
                        total_matches += jobs[t].matches;
                        total_sum += jobs[t].sum;
                        if (writer_push(&g_writer, &jobs[t].buf)) pipeline_fail("Out of memory");
                    }
                    free(jobs);
                    free(tid);
                    g_fmt_longest += fmt_longest;
                    g_fmt_latest_start += fmt_latest_start;
                }
                if (g_writer.failed) pipeline_fail("Cannot write output file");
            }
            t_format += now_s() - tg;
        }
#undef ADD_RANGES_OF_NEW_SETS
        free(ranges);
        release_stdout(); /* the loader is done: the "successfully loaded" line, then what was held back */
        total_queries = ld->total_queries;
        if (ld->failed) pipeline_fail("Internal error: the query files came in more pieces than planned");
        if (ld->ready == 0) { /* slamem.c:648 (an overlapped run only knows it now) */
            shutdown_pipeline();
            fclose(out);
            remove(out_name);
            exit_message("No query files provided");
        }
        writer_finish(&g_writer);
        t_write = g_writer.seconds;
        if (g_writer.failed) pipeline_fail("Cannot write output file");
        if (log_limit != 0 && (long)total_queries * strands > log_limit)
            say(":: ... (%ld more strand blocks matched; set SLAMEM_VERBOSE=1 for a line each)\n",
                   (long)total_queries * strands - log_limit);
    }
    double t_end0 = now_s(), t_end1;
    if (total_queries != 1) /* slamem.c:210-212 (the reference divides by zero when nothing matched) */
        printf(":: Average %d M%cMs found per query sequence (total = %lld, avg size = %d bp)\n",
               (int)(total_matches / total_queries), MATCH_TYPE_CHAR[o.match_type], total_matches,
               (int)(total_matches ? total_sum / total_matches : 0));
    fflush(stdout);
    printf("> Saving M%cMs to <%s> ... ", MATCH_TYPE_CHAR[o.match_type], out_name);
    if (fflush(out) != 0 || ferror(out)) exit_message("Cannot write output file");
    if (getenv("SLAMEM_FULL_TEARDOWN") != NULL && fclose(out) != 0) exit_message("Cannot write output file");
    t_end1 = now_s();
    printf("OK\n");
    printf("> Done!\n");
    char overlap_note[96] = "";
    if (g_ld.seconds > 0)
        snprintf(overlap_note, sizeof(overlap_note), " + queries parsed in pieces beside the search in %.3f s (%.3f s waited for)", g_ld.seconds, t_wait_load);
    if (timing)
        fprintf(stderr, "[timing] formatter threads: busy %.3f s in all; per batch the longest chunk %.3f s, the last chunk started %.3f s after the batch began (sums over the batches)\n",
                g_fmt_busy, g_fmt_longest, g_fmt_latest_start);
    if (timing)
        fprintf(stderr, "[timing] load %.3f s%s (index build of %.3f s overlapped; %.3f s more waiting for it), pipeline set-up %.3f s, "
                        "waiting for the GPU (upload + search + download, overlapped with formatting) %.3f s, format %.3f s "
                        "(writer thread busy %.3f s, overlapped; text buffers: %ld recycled, %ld fresh), close %.3f s, total %.3f s\n",
                t_load, overlap_note, t_build, t_join, t_streams, t_gpu, t_format, t_write, g_pool_hits, g_pool_misses, t_end1 - t_end0,
                now_s() - t_start);
    fflush(stdout);
    fflush(stderr);
    if (getenv("SLAMEM_FULL_TEARDOWN") == NULL) {
        /* everything is written and nothing runs on the GPU any more: leave without returning gigabytes of buffers and
           HBM piece by piece (0.15-0.25 s of the reference-sized run); the kernel driver reclaims them with the process */
        leave_now();
    }
    {
        double a = now_s(), b, c, d;
        shutdown_pipeline();
        b = now_s();
        for (i = 0; i < ngpu; i++) slamem_index_free(gpus[i]);
        c = now_s();
        if (o.out_arg == -1) free(out_name);
        slh_buffer_free(&buf);
        slh_free_seqset(&ref);
        for (f = 0; f < g_reap_n; f++) pthread_join(g_reap_tid[f], NULL);
        free(g_reap_tid);
        for (f = 0; f < g_ld.ready; f++) slh_free_seqset(&qsets[f]);
        free(qsets);
        pthread_mutex_lock(&g_pool_mu);
        while (g_pool_n > 0) slh_buffer_free(&g_pool[--g_pool_n]);
        pthread_mutex_unlock(&g_pool_mu);
        d = now_s();
        if (timing)
            fprintf(stderr, "[timing] teardown: search pipeline (pinned buffers, device work space) %.3f s, index %.3f s, host buffers %.3f s\n",
                    b - a, c - b, d - c);
    }
    slh_free_options(&o);
    return 0;
}
