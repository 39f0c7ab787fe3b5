// stream.hip -- host-to-host MEM retrieval, pipelined: the form of GetMatches' query loop (slamem.c:90-207) that a
// front end with its reads in HOST memory calls.  SURVEY.md 8(d) defines the path's metric on exactly this boundary
// ("reads resident in host memory -> MEM triples in host memory").
//
// A slamem_stream is a four-stage pipeline over `slots` (2-8) sets of buffers, each stage a host thread with its own HIP
// stream, each taking the batches in submission order:
//   UPLOAD    the batch's characters and offsets to the device (straight from the caller's memory: at full PCIe rate when
//             that memory is pinned, slamem_pinned_alloc)
//   PREPARE   work-item tables (the stage's one host round trip), K8a prefilter, work list, K7q packing
//   SEARCH    K8 and K9 of ALL batches back to back on one stream, with no host round trip in between: the thread only
//             enqueues (the preparation is awaited on the device, hipStreamWaitEvent) and never waits for a result
//   DOWNLOAD  waits for the batch's K9 (event), reads its totals, copies MEMs and block offsets into pinned host memory
// K8 is a persistent kernel that takes every wave slot of the chip; what it leaves idle is its tail (the last strands of
// its waves, ~1 ms whatever the batch size).  The preparation of the NEXT batches runs on its own stream, so the hardware
// puts its workgroups exactly where K8's waves retire, and the next K8 is already queued behind K9.  (Round 2 ran each
// batch's whole chain, with three host round trips, on one of two alternating search threads: the second thread's small
// kernels and memsets sat behind the first one's K8 until its tail and the chain after them was exposed -- 3.7 ms per
// million-read batch where the kernels need 2.9; measured with rocprofv3 --kernel-trace, profiles/r03_host_leg_timeline.txt.
// One thread per SLOT doing everything in turn was measured before that: the slots fall into lockstep, 62 ms.)
// No CPU fallback: every batch is searched on the GPU.
#include "common.h"
#include "prims.h"

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

namespace slamem {

namespace {

enum SlotState { FREE = 0, QUEUED = 1, UPLOADED = 2, PREPARED = 3, LAUNCHED = 4, DONE = 5, RETURNED = 6,
                 K8_ISSUED = 7 };  // K8 enqueued with its lanes passed on; K9 follows behind the next batch's K8 (then LAUNCHED)
constexpr int kMaxSlots = 8;
constexpr int kThreads = 4;  // upload, prepare, search, download
enum { T_UP = 0, T_PREP = 1, T_SEARCH = 2, T_DOWN = 3 };

struct Slot {
    int state = FREE;
    // device
    void* d_q = nullptr;
    uint64_t* d_off = nullptr;
    uint64_t* d_boff = nullptr;
    slamem_mem* d_mems = nullptr;
    void* d_ws = nullptr;
    uint64_t cap = 0, ws_bytes = 0;
    uint64_t cap_chars = 0;   // room of d_q (characters), of d_off / d_boff / h_boff (records): grown when a batch needs more
    uint32_t cap_q = 0;
    // pinned host
    uint64_t* h_boff = nullptr;
    slamem_mem* h_mems = nullptr;
    uint64_t h_cap = 0, h_boff_cap = 0;
    // the batch
    uint64_t seq = 0;
    const char* chars = nullptr;
    const uint64_t* offs = nullptr;
    uint32_t nq = 0, min_len = 0;
    bool all_short = false;  // no record longer than a slice (seen by the upload stage while the copy engines work)
    // a batch handed in as bit-planes (slamem_stream_submit_packed): what goes up, and the device side of it
    const void* planes = nullptr;       // 16-byte units {p0, p1}: letters 64u .. 64u+63 of a record; a record starts a new unit
    const uint64_t* other = nullptr;    // per unit: letters that are not A,C,G,T (nullptr: none)
    void* d_planes = nullptr;
    uint64_t* d_other = nullptr;
    uint32_t* d_ucnt = nullptr;         // units per record, then their prefix sums (and the scan's scratch behind them)
    uint64_t cap_units = 0;
    uint64_t units_hint = 0;            // the batch's units as the caller counted them (0: count here)
    // its result
    uint64_t total = 0;
    int rc = SLAMEM_OK;
    char err[512] = "";
    slamem_timings tm;
    // the batch on its way through the search (mem_search.hip), the events the stages hand it over with, and the pinned
    // words K9's stream copies the batch's totals into
    SearchJob* job = nullptr;
    hipEvent_t ev_prep = nullptr, ev_done = nullptr, ev_k8 = nullptr;
    unsigned long long* h_scal = nullptr;
};

}  // namespace

}  // namespace slamem

using namespace slamem;

struct slamem_stream {
    const slamem_index* idx = nullptr;
    int nslots = 0, both = 0, match_type = 0;
    uint64_t max_chars = 0;
    uint32_t max_q = 0;
    Slot slot[kMaxSlots];
    std::thread th[kThreads];
    hipStream_t st[kThreads] = {};
    hipStream_t st_upx[3] = {nullptr, nullptr, nullptr};  // more copy streams of the upload stage (SLAMEM_STREAM_UPLOAD_SPLIT = 2..4)
    int upload_split = 2;
    int nthreads = kThreads;
    // SLAMEM_STREAM_CARRY=1: K8 without its tail (below).  Off by default: it does not pay -- every result then waits for the
    // NEXT batch's K8, and the carried lanes finish their strands at the head of that launch instead of the tail of this one
    // (headline workload, round 3: 39.5-40.0 ms against 38.5; a stream of 30 M reads 113.8 against 108.6 ms; the "steady rate" of
    // 868 M MEMs/s the first runs showed was an artefact of where the results' time stamps fall; profiles/r03_host_leg.jsonl)
    bool carry = false;
    Slot* pending = nullptr;     // search stage only: the batch whose K8 has passed its unfinished lanes on
    hipStream_t st_place = nullptr;    // K9 of every batch: behind the K8 that finished it, but not in front of the next K8
    hipStream_t st_search2 = nullptr;  // SLAMEM_STREAM_SEARCH_STREAMS=2: odd batches' K8 + K9 on a second stream (their K8 starts in the tail of the even one's)
    int search_streams = 1;
    uint32_t k8_waves = 2560;  // two search streams: waves of a K8 that has another batch behind it (SLAMEM_STREAM_K8_WAVES; the chip holds 4096)
    double mems_per_char = 0;  // the densest batch so far: sizes a slot's first output buffers (written by the download stage)
    std::mutex mu;
    std::condition_variable cv;
    uint64_t submitted = 0, returned = 0;  // batches handed in / handed back
    bool stop = false, trace = false;
    std::chrono::steady_clock::time_point t0;
};

namespace slamem {
namespace {

// Buffers are allocated by the stage that uses them, on a slot's first batch (and again if a batch needs more room):
// setting a stream up costs nothing, the allocations (pinned host memory above all: ~0.1 ms per MB) overlap with the
// other stages' work, and slots that are never used are never allocated.
int grow_outputs(slamem_stream* s, Slot& sl, uint64_t need_cap) {  // search stage: device output + workspace
    if (sl.d_mems) (void)hipFree(sl.d_mems);
    if (sl.d_ws) (void)hipFree(sl.d_ws);
    sl.d_mems = nullptr; sl.d_ws = nullptr;
    sl.cap = need_cap;
    sl.ws_bytes = find_mems_workspace_bytes(sl.cap_q, s->both, sl.cap_chars, sl.cap);
    SLAMEM_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_mems), sl.cap * sizeof(slamem_mem) + 16));
    SLAMEM_HIP(hipMalloc(&sl.d_ws, sl.ws_bytes));
    return SLAMEM_OK;
}

// stage 0: characters and offsets to the device.  The offsets go up as the caller holds them (absolute): the characters land
// at d_q + kFront + (base mod 16) and the kernels get the pointer that record offset `base` maps to -- 16-byte aligned like
// the caller's buffer start, so nothing is rebased on the host (a loop over a million offsets per batch made this stage
// the pipeline's bottleneck).
constexpr uint64_t kFront = 32;  // bytes in front of the characters: the kernels read whole aligned words around a record
inline const char* device_queries(const Slot& sl) {
    const uint64_t base = sl.offs[0];
    return static_cast<const char*>(sl.d_q) + kFront + (base & 15u) - base;
}
// ---- reads that come as bit-planes (slamem_stream_submit_packed) -------------------------------------------------------------
// The link is what bounds the host-to-host path since the search takes 9 ms for 10 M reads (1.5 GB of letters: 26 ms at 57
// GB/s); a caller that holds its reads packed -- two bits a letter in the layout of the index's text planes, 48 bytes per 150
// letters -- sends a third of that.  On the device the letters are written out again (k_unpack_reads: 0.5 ms per 1.5 GB) and the
// batch goes the way of every other; letters that are not A,C,G,T travel as a third plane (or not at all: other = nullptr).
__global__ void __launch_bounds__(256) k_unit_counts(const uint64_t* __restrict__ offsets, uint32_t nq, uint32_t* __restrict__ cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nq) return;
    cnt[i] = i < nq ? (uint32_t)((offsets[i + 1] - offsets[i] + 63u) >> 6) : 0u;  // cnt[nq] = 0: the scan leaves the total there
}
constexpr uint32_t kUnpackRecords = 128, kUnpackBytes = 32768;
// a block writes the letters of kUnpackRecords consecutive records: decoded into LDS (16 letters per lane and step), then out in
// aligned 16-byte pieces; records too long for the buffer are written letter by letter
__global__ void __launch_bounds__(256) k_unpack_reads(const uint4* __restrict__ planes, const uint64_t* __restrict__ other,
                                                      const uint32_t* __restrict__ uoff, const uint64_t* __restrict__ offsets, uint32_t nq,
                                                      char* __restrict__ out /* where record offset offsets[0] goes */) {
    __shared__ __attribute__((aligned(16))) char buf[kUnpackBytes + 16];
    const uint32_t r0 = blockIdx.x * kUnpackRecords, r1 = r0 + kUnpackRecords < nq ? r0 + kUnpackRecords : nq;
    if (r0 >= nq) return;
    const uint64_t base = offsets[0], lo = offsets[r0], hi = offsets[r1];
    char* dst = out + (lo - base);
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);  // buf[shift + x] is byte x of the range
    const bool staged = hi - lo <= kUnpackBytes;
    auto letter = [&](uint64_t p0, uint64_t p1, uint64_t ot, uint32_t bit) -> char {
        const uint32_t c = (uint32_t)((p0 >> bit) & 1ull) | ((uint32_t)((p1 >> bit) & 1ull) << 1);
        return ((ot >> bit) & 1ull) ? 'N' : "ACGT"[c];
    };
    // 2 lanes per record; a lane takes 16 letters at a time, 32 apart
    for (uint32_t r = r0 + (threadIdx.x >> 1); r < r1; r += 128u) {
        const uint64_t ro = offsets[r];
        const uint32_t len = (uint32_t)(offsets[r + 1] - ro);
        const uint32_t u0 = uoff[r];
        for (uint32_t x = (threadIdx.x & 1u) * 16u; x < len; x += 32u) {
            const uint4 pu = planes[u0 + (x >> 6)];
            const uint64_t p0 = ((uint64_t)pu.y << 32) | pu.x, p1 = ((uint64_t)pu.w << 32) | pu.z;
            const uint64_t ot = other ? other[u0 + (x >> 6)] : 0ull;
            const uint32_t n = len - x < 16u ? len - x : 16u;
            char* w = staged ? buf + shift + (uint32_t)(ro - lo) + x : dst + (ro - lo) + x;
            for (uint32_t i = 0; i < n; i++) w[i] = letter(p0, p1, ot, (x & 63u) + i);
        }
    }
    if (!staged) return;
    __syncthreads();
    const uint32_t nbytes = (uint32_t)(hi - lo);
    // bytes [0, nbytes) of the range sit at buf[shift ..]; global address dst - shift is 16-byte aligned
    const uint32_t first_full = shift ? 16u - shift : 0u;  // range bytes in front of the first aligned piece
    for (uint32_t i = threadIdx.x; i < first_full && i < nbytes; i += 256u) dst[i] = buf[shift + i];
    if (nbytes > first_full) {
        const uint32_t pieces = (nbytes - first_full) >> 4;
        uint4* g = reinterpret_cast<uint4*>(dst + first_full);
        const uint4* l = reinterpret_cast<const uint4*>(buf + shift + first_full);
        for (uint32_t i = threadIdx.x; i < pieces; i += 256u) g[i] = l[i];
        for (uint32_t i = first_full + (pieces << 4) + threadIdx.x; i < nbytes; i += 256u) dst[i] = buf[shift + i];
    }
}

int stage_upload_packed(slamem_stream* s, Slot& sl, char* dst) {
    // the units of the batch: as the caller says, or counted here (a pass over the offsets: ~1 ms per million records, in front
    // of the copies -- a caller that knows the number from slamem_pack_reads saves it)
    uint64_t units = sl.units_hint;
    if (units == 0)
        for (uint32_t i = 0; i < sl.nq; i++) units += (sl.offs[i + 1] - sl.offs[i] + 63u) >> 6;
    if (units >= 0xFFFFFFFFull) { set_error("slamem_stream_submit_packed: fewer than 2^32 units (64 letters) per batch"); return SLAMEM_ERR_ARG; }
    if (!sl.d_planes || units > sl.cap_units) {
        if (sl.d_planes) (void)hipFree(sl.d_planes);
        if (sl.d_other) (void)hipFree(sl.d_other);
        sl.d_planes = nullptr; sl.d_other = nullptr; sl.cap_units = 0;
        const uint64_t want = units > (s->max_chars >> 6) + s->max_q ? units : (s->max_chars >> 6) + s->max_q;
        SLAMEM_HIP(hipMalloc(&sl.d_planes, (want + 1) * 16));
        SLAMEM_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_other), (want + 1) * 8));
        sl.cap_units = want;
    }
    if (!sl.d_ucnt) SLAMEM_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_ucnt), ((uint64_t)sl.cap_q + 2 + scan_u32_tmp_words((uint64_t)sl.cap_q + 1)) * 4));
    hipStream_t st = s->st[0];
    SLAMEM_HIP(hipMemcpyAsync(sl.d_off, sl.offs, ((uint64_t)sl.nq + 1) * 8, hipMemcpyHostToDevice, st));
    // the planes on the copy streams side by side (as the letters of an ordinary batch)
    const uint64_t pbytes = units * 16;
    const int ways = (s->upload_split > 1 && pbytes >= (8u << 20)) ? s->upload_split : 1;
    uint64_t at = 0;
    for (int wy = 0; wy < ways; wy++) {
        const uint64_t end = wy + 1 == ways ? pbytes : ((pbytes * (uint64_t)(wy + 1) / (uint64_t)ways) & ~(uint64_t)4095);
        hipStream_t cs = wy == 0 ? st : s->st_upx[wy - 1];
        if (end > at) SLAMEM_HIP(hipMemcpyAsync(static_cast<char*>(sl.d_planes) + at, static_cast<const char*>(sl.planes) + at, end - at, hipMemcpyHostToDevice, cs));
        at = end;
    }
    if (sl.other && units) SLAMEM_HIP(hipMemcpyAsync(sl.d_other, sl.other, units * 8, hipMemcpyHostToDevice, st));
    if (sl.nq) {
        hipLaunchKernelGGL(k_unit_counts, dim3((unsigned)(((uint64_t)sl.nq + 1 + 255) / 256)), dim3(256), 0, st, (const uint64_t*)sl.d_off, sl.nq, sl.d_ucnt);
        SLAMEM_HIP(exclusive_scan_u32(sl.d_ucnt, sl.d_ucnt, (uint64_t)sl.nq + 1, sl.d_ucnt + sl.cap_q + 2, st));
    }
    {   // while the copy engines work: is any record longer than a slice?  (as in stage_upload)
        uint64_t longest = 0;
        for (uint32_t i = 0; i < sl.nq; i++) {
            const uint64_t len = sl.offs[i + 1] - sl.offs[i];
            longest = len > longest ? len : longest;
        }
        sl.all_short = longest <= kSearchSliceLen;
    }
    for (int wy = 1; wy < ways; wy++) SLAMEM_HIP(hipStreamSynchronize(s->st_upx[wy - 1]));  // (the letters are written on st: every piece must be there)
    if (sl.nq) {
        hipLaunchKernelGGL(k_unpack_reads, dim3((sl.nq + kUnpackRecords - 1) / kUnpackRecords), dim3(256), 0, st, (const uint4*)sl.d_planes,
                           sl.other ? (const uint64_t*)sl.d_other : (const uint64_t*)nullptr, (const uint32_t*)sl.d_ucnt, (const uint64_t*)sl.d_off, sl.nq, dst);
        SLAMEM_HIP(hipGetLastError());
    }
    SLAMEM_HIP(hipStreamSynchronize(st));
    return SLAMEM_OK;
}

int stage_upload(slamem_stream* s, Slot& sl) {
    const uint64_t base = sl.offs[0], qbytes = sl.offs[sl.nq] - base;
    if (!sl.d_q || qbytes > sl.cap_chars || sl.nq > sl.cap_q) {
        // first batch of this slot, or a batch larger than the reservation (max_batch_* of slamem_stream_create are a hint)
        const uint64_t nchars = qbytes > s->max_chars ? qbytes : s->max_chars;
        const uint32_t nrec = sl.nq > s->max_q ? sl.nq : s->max_q;
        const uint64_t nb = (uint64_t)nrec * (s->both ? 2 : 1);
        if (sl.d_q) (void)hipFree(sl.d_q);
        if (sl.d_off) (void)hipFree(sl.d_off);
        if (sl.d_boff) (void)hipFree(sl.d_boff);
        if (sl.d_mems) (void)hipFree(sl.d_mems);
        if (sl.d_ws) (void)hipFree(sl.d_ws);
        if (sl.d_ucnt) (void)hipFree(sl.d_ucnt);
        sl.d_ucnt = nullptr;
        sl.d_q = nullptr; sl.d_off = nullptr; sl.d_boff = nullptr; sl.d_mems = nullptr; sl.d_ws = nullptr;  // (the prepare stage sizes its own)
        sl.cap_chars = 0; sl.cap_q = 0;  // until all three are there: a failed allocation must not leave stale room behind
        SLAMEM_HIP(hipMalloc(&sl.d_q, nchars + 2 * kFront + 32));
        SLAMEM_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_off), ((uint64_t)nrec + 1) * 8));
        SLAMEM_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_boff), (nb + 1) * 8));
        sl.cap_chars = nchars;
        sl.cap_q = nrec;
    }
    char* dst = static_cast<char*>(sl.d_q) + kFront + (base & 15u);
    if (sl.planes) return stage_upload_packed(s, sl, dst);
    const char* src = sl.chars + base;
    // Beside K8 -- which keeps the memory system at its random-request ceiling -- ONE copy engine moves 42 GB/s instead of the
    // link's 57 (measured: 3.56 ms per 158 MB instead of 2.76, which made the upload the pipeline's period); the pieces of a
    // batch go up on `upload_split` copy streams side by side
    const int ways = (s->upload_split > 1 && qbytes >= (8u << 20)) ? s->upload_split : 1;
    uint64_t at = 0;
    for (int wy = 0; wy < ways; wy++) {
        const uint64_t end = wy + 1 == ways ? qbytes : ((qbytes * (uint64_t)(wy + 1) / (uint64_t)ways) & ~(uint64_t)4095);
        hipStream_t cs = wy == 0 ? s->st[0] : s->st_upx[wy - 1];
        if (end > at) SLAMEM_HIP(hipMemcpyAsync(dst + at, src + at, end - at, hipMemcpyHostToDevice, cs));
        at = end;
    }
    SLAMEM_HIP(hipMemcpyAsync(sl.d_off, sl.offs, ((uint64_t)sl.nq + 1) * 8, hipMemcpyHostToDevice, s->st[0]));
    {   // while the copy engines work: is any record longer than a slice?  (if not, the prepare stage knows the number of work
        // items without asking the device -- its one host round trip per batch goes away)
        uint64_t longest = 0;
        for (uint32_t i = 0; i < sl.nq; i++) {
            const uint64_t len = sl.offs[i + 1] - sl.offs[i];
            longest = len > longest ? len : longest;
        }
        sl.all_short = longest <= kSearchSliceLen;
    }
    SLAMEM_HIP(hipStreamSynchronize(s->st[0]));
    for (int wy = 1; wy < ways; wy++) SLAMEM_HIP(hipStreamSynchronize(s->st_upx[wy - 1]));
    return SLAMEM_OK;
}

// stage 1: work-item tables (one small host round trip on this stage's stream), then K8a, the work list and K7q, asynchronous
int job_setup(slamem_stream* s, Slot& sl) {
    const uint64_t qbytes = sl.offs[sl.nq] - sl.offs[0];
    if (!sl.d_ws) {
        // first guess of the room for MEMs: what the densest batch so far would need, and never less than a MEM per 32
        // characters; grown when a batch needs more (SLAMEM_ERR_CAPACITY tells how much)
        const uint64_t nb = (uint64_t)sl.cap_q * (s->both ? 2 : 1);
        uint64_t guess = sl.cap_chars / 32 + nb + 1024;
        const uint64_t seen = (uint64_t)(1.25 * s->mems_per_char * (double)sl.cap_chars) + 1024;
        int rc = grow_outputs(s, sl, seen > guess ? seen : guess);
        if (rc != SLAMEM_OK) return rc;
    }
    if (!sl.job && !(sl.job = search_job_new())) { set_error("out of host memory"); return SLAMEM_ERR_NOMEM; }
    if (!sl.ev_prep) SLAMEM_HIP(hipEventCreateWithFlags(&sl.ev_prep, hipEventDisableTiming));
    if (!sl.ev_done) SLAMEM_HIP(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
    if (!sl.ev_k8) SLAMEM_HIP(hipEventCreateWithFlags(&sl.ev_k8, hipEventDisableTiming));
    if (!sl.h_scal) SLAMEM_HIP(hipHostMalloc(reinterpret_cast<void**>(&sl.h_scal), 16 * sizeof(unsigned long long), hipHostMallocDefault));
    return search_job_init(sl.job, s->idx, device_queries(sl), sl.d_off, sl.nq, qbytes, sl.min_len, s->both, s->match_type, sl.d_mems,
                           sl.cap, sl.d_boff, sl.d_ws, sl.ws_bytes, sl.h_scal);
}
int stage_prepare(slamem_stream* s, Slot& sl) {
    int rc = job_setup(s, sl);
    if (rc == SLAMEM_OK && sl.all_short) search_job_slices_hint(sl.job, sl.nq);
    if (rc == SLAMEM_OK) rc = search_job_tables(sl.job, s->st[T_PREP]);
    if (rc == SLAMEM_OK) rc = search_job_prep(sl.job, s->st[T_PREP]);
    if (rc == SLAMEM_OK) SLAMEM_HIP(hipEventRecord(sl.ev_prep, s->st[T_PREP]));
    return rc;
}

// stage 2: K8 + K9 behind the preparation, enqueued only: the search stream holds the K8s and K9s of all batches in flight,
// one behind the other.  K8 WITHOUT ITS TAIL: when another batch has been submitted behind this one, K8 ends as soon as its
// work list is empty and passes the unfinished lanes on (search_job_k8 carry_out); the next batch's K8 takes them in first,
// and only then this batch's K9 follows -- the chip never drains between two batches of a stream.  The last batch (nothing
// submitted behind it) runs to its end as before.  `failed`: the batch comes from an earlier stage with an error and only
// the batch waiting for it has to be finished.
void finish_pending(slamem_stream* s, hipStream_t st) {  // K9 of the batch whose lanes are all through now; hand it to the download stage
    Slot* p = s->pending;
    s->pending = nullptr;
    hipStream_t ps = s->st_place ? s->st_place : st;
    if (ps != st) { (void)hipEventRecord(p->ev_k8, st); (void)hipStreamWaitEvent(ps, p->ev_k8, 0); }
    int rc = search_job_place(p->job, ps);
    (void)hipEventRecord(p->ev_done, ps);
    if (rc != SLAMEM_OK) snprintf(p->err, sizeof(p->err), "%s", slamem_last_error_message());
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (rc != SLAMEM_OK) p->rc = rc;
        p->state = LAUNCHED;
    }
    s->cv.notify_all();
}
int stage_search(slamem_stream* s, Slot& sl, bool failed, bool* issued_only) {
    hipStream_t st = (s->search_streams > 1 && (sl.seq & 1u)) ? s->st_search2 : s->st[T_SEARCH];
    *issued_only = false;
    if (failed) {
        if (s->pending) {
            (void)search_job_flush(s->pending->job, st);
            finish_pending(s, st);
        }
        return SLAMEM_OK;
    }
    bool more, follows;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        follows = s->submitted > sl.seq + 1;
        more = s->carry && s->search_streams == 1 && follows;
    }
    // two search streams: a batch that has another one behind it takes only part of the chip for its K8, so that the next
    // batch's preparation and then its K8 start beside it instead of in its tail; the last batch of a stream takes all of it
    search_job_k8_wave_cap(sl.job, (s->search_streams > 1 && follows) ? s->k8_waves : 0u);
    SLAMEM_HIP(hipStreamWaitEvent(st, sl.ev_prep, 0));
    int rc;
    if (s->pending && search_job_can_carry_into(s->pending->job, sl.job)) {
        rc = search_job_k8(sl.job, st, s->pending->job, more);
        if (rc != SLAMEM_OK) (void)search_job_flush(s->pending->job, st);
        finish_pending(s, st);
    } else {
        if (s->pending) {
            (void)search_job_flush(s->pending->job, st);
            finish_pending(s, st);
        }
        rc = search_job_k8(sl.job, st, nullptr, more);
    }
    if (rc == SLAMEM_OK && search_job_carried_out(sl.job)) {
        s->pending = &sl;
        *issued_only = true;  // its K9 and its event follow behind the next batch's K8
        return SLAMEM_OK;
    }
    hipStream_t ps = s->st_place ? s->st_place : st;
    if (ps != st) { SLAMEM_HIP(hipEventRecord(sl.ev_k8, st)); SLAMEM_HIP(hipStreamWaitEvent(ps, sl.ev_k8, 0)); }
    if (rc == SLAMEM_OK) rc = search_job_place(sl.job, ps);
    // (recorded even after a failed launch: the download stage waits for whatever did get onto the stream)
    SLAMEM_HIP(hipEventRecord(sl.ev_done, ps));
    return rc;
}

// stage 3: the batch's totals, then MEMs and block offsets to pinned host memory
int stage_download(slamem_stream* s, Slot& sl) {
    hipStream_t st = s->st[T_DOWN];
    (void)slamem_reset_timings();
    SLAMEM_HIP(hipEventSynchronize(sl.ev_done));
    int rc = search_job_collect(sl.job, &sl.total);
    for (int attempt = 0; rc == SLAMEM_ERR_CAPACITY && sl.total > sl.cap && attempt < 2; attempt++) {
        // rare (the first guess was too small): more room, and the batch once more, start to end, on this stage's stream
        rc = grow_outputs(s, sl, sl.total + sl.total / 8 + 1024);
        if (rc == SLAMEM_OK) rc = job_setup(s, sl);
        if (rc == SLAMEM_OK) rc = search_job_tables(sl.job, st);
        if (rc == SLAMEM_OK) rc = search_job_prep(sl.job, st);
        if (rc == SLAMEM_OK) rc = search_job_search(sl.job, st);
        SLAMEM_HIP(hipStreamSynchronize(st));
        if (rc == SLAMEM_OK) rc = search_job_collect(sl.job, &sl.total);
    }
    (void)slamem_get_timings(&sl.tm);
    if (rc != SLAMEM_OK) return rc;
    {
        const uint64_t chars = sl.offs[sl.nq] - sl.offs[0];
        const double d = chars ? (double)sl.total / (double)chars : 0.0;
        std::lock_guard<std::mutex> lk(s->mu);
        if (d > s->mems_per_char) s->mems_per_char = d;
    }
    const uint64_t nb = (uint64_t)sl.nq * (s->both ? 2 : 1);
    if (!sl.h_boff || sl.h_boff_cap < nb + 1) {
        if (sl.h_boff) (void)hipHostFree(sl.h_boff);
        sl.h_boff = nullptr;
        const uint64_t want = (uint64_t)sl.cap_q * (s->both ? 2 : 1) + 1;
        sl.h_boff_cap = want > nb + 1 ? want : nb + 1;
        SLAMEM_HIP(hipHostMalloc(reinterpret_cast<void**>(&sl.h_boff), sl.h_boff_cap * 8, hipHostMallocDefault));
    }
    if (sl.h_cap < sl.cap || !sl.h_mems) {
        if (sl.h_mems) (void)hipHostFree(sl.h_mems);
        sl.h_mems = nullptr;
        SLAMEM_HIP(hipHostMalloc(reinterpret_cast<void**>(&sl.h_mems), sl.cap * sizeof(slamem_mem) + 16, hipHostMallocDefault));
        sl.h_cap = sl.cap;
    }
    if (sl.total) SLAMEM_HIP(hipMemcpyAsync(sl.h_mems, sl.d_mems, sl.total * sizeof(slamem_mem), hipMemcpyDeviceToHost, st));
    SLAMEM_HIP(hipMemcpyAsync(sl.h_boff, sl.d_boff, (nb + 1) * 8, hipMemcpyDeviceToHost, st));
    SLAMEM_HIP(hipStreamSynchronize(st));
    return SLAMEM_OK;
}

// one thread per stage; each takes its batches in submission order
void worker(slamem_stream* s, int t) {
    static const int kWant[kThreads] = {QUEUED, UPLOADED, PREPARED, LAUNCHED};
    static const int kDone[kThreads] = {UPLOADED, PREPARED, LAUNCHED, DONE};
    static const char* const kName[kThreads] = {"upload", "prepare", "search", "download"};
    const int want = kWant[t], done = kDone[t];
    (void)hipSetDevice(s->idx->device);
    for (uint64_t seq = 0;; seq++) {
        Slot& sl = s->slot[seq % (uint64_t)s->nslots];
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->stop || (sl.state == want && sl.seq == seq && seq < s->submitted); });
            if (s->stop) return;
        }
        int rc = sl.rc;  // a batch that failed in an earlier stage passes through untouched
        bool issued_only = false;
        const auto t_begin = std::chrono::steady_clock::now();
        if (rc == SLAMEM_OK) {
            rc = t == T_UP ? stage_upload(s, sl) : t == T_PREP ? stage_prepare(s, sl) : t == T_SEARCH ? stage_search(s, sl, false, &issued_only)
                                                                                                    : stage_download(s, sl);
            if (rc != SLAMEM_OK) snprintf(sl.err, sizeof(sl.err), "%s", slamem_last_error_message());  // the text is per thread
        } else if (t == T_SEARCH) {
            (void)stage_search(s, sl, true, &issued_only);
        } else if (t == T_DOWN && sl.ev_done) {
            (void)hipEventSynchronize(sl.ev_done);  // nothing of a failed batch may still run when its slot is handed back
        }
        if (s->trace) {  // SLAMEM_STREAM_TRACE=1: when every stage worked on every batch (ms since the stream was created)
            const auto t_end = std::chrono::steady_clock::now();
            fprintf(stderr, "[stream] batch %3llu %-8s %9.3f .. %9.3f ms\n", (unsigned long long)seq, kName[t],
                    std::chrono::duration<double, std::milli>(t_begin - s->t0).count(),
                    std::chrono::duration<double, std::milli>(t_end - s->t0).count());
        }
        {
            std::lock_guard<std::mutex> lk(s->mu);
            sl.rc = rc;
            sl.state = issued_only ? (int)K8_ISSUED : done;
        }
        s->cv.notify_all();
    }
}

void free_slot(Slot& sl) {
    if (sl.d_q) (void)hipFree(sl.d_q);
    if (sl.d_off) (void)hipFree(sl.d_off);
    if (sl.d_boff) (void)hipFree(sl.d_boff);
    if (sl.d_mems) (void)hipFree(sl.d_mems);
    if (sl.d_ws) (void)hipFree(sl.d_ws);
    if (sl.d_planes) (void)hipFree(sl.d_planes);
    if (sl.d_other) (void)hipFree(sl.d_other);
    if (sl.d_ucnt) (void)hipFree(sl.d_ucnt);
    if (sl.h_boff) (void)hipHostFree(sl.h_boff);
    if (sl.h_mems) (void)hipHostFree(sl.h_mems);
    if (sl.h_scal) (void)hipHostFree(sl.h_scal);
    if (sl.ev_prep) (void)hipEventDestroy(sl.ev_prep);
    if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    if (sl.ev_k8) (void)hipEventDestroy(sl.ev_k8);
    if (sl.job) search_job_delete(sl.job);
}

}  // namespace
}  // namespace slamem

extern "C" {

int slamem_pinned_alloc(void** out, uint64_t bytes) {
    if (!out) return SLAMEM_ERR_ARG;
    *out = nullptr;
    SLAMEM_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return SLAMEM_OK;
}

// letters -> bit-planes on the host (what a caller without packed reads of its own puts in front of slamem_stream_submit_packed)
int slamem_pack_reads(const char* queries, const uint64_t* offsets, uint32_t num_queries, void* planes_out, uint64_t* other_out,
                      uint64_t* units_out, int threads) {
    if (!offsets || (num_queries && (!queries || !planes_out))) { set_error("slamem_pack_reads: null argument"); return SLAMEM_ERR_ARG; }
    // the first unit of every record: a prefix sum (one pass), then the records in ranges, a thread each
    std::vector<uint64_t> first((size_t)num_queries + 1);
    uint64_t units = 0;
    for (uint32_t i = 0; i < num_queries; i++) { first[i] = units; units += (offsets[i + 1] - offsets[i] + 63u) >> 6; }
    first[num_queries] = units;
    if (units_out) *units_out = units;
    uint64_t* pl = static_cast<uint64_t*>(planes_out);
    auto work = [&](uint32_t a, uint32_t b) {
        for (uint32_t r = a; r < b; r++) {
            const unsigned char* q = reinterpret_cast<const unsigned char*>(queries) + offsets[r];
            const uint64_t len = offsets[r + 1] - offsets[r];
            for (uint64_t u = 0; u * 64 < len; u++) {
                uint64_t p0 = 0, p1 = 0, ot = 0;
                const uint64_t n = len - u * 64 < 64 ? len - u * 64 : 64;
                for (uint64_t i = 0; i < n; i++) {
                    const unsigned c = q[u * 64 + i] & 0xDFu;  // upper case
                    const unsigned x = (c >> 1) & 3u, code = x ^ (x >> 1);  // A 0, C 1, G 2, T 3
                    const bool ok = c == 'A' || c == 'C' || c == 'G' || c == 'T';
                    if (ok) { p0 |= (uint64_t)(code & 1u) << i; p1 |= (uint64_t)(code >> 1) << i; } else ot |= 1ull << i;
                }
                pl[2 * (first[r] + u)] = p0;
                pl[2 * (first[r] + u) + 1] = p1;
                if (other_out) other_out[first[r] + u] = ot;
            }
        }
    };
    const int nt = threads < 1 ? 1 : threads > 64 ? 64 : threads;
    if (nt == 1 || num_queries < 4096u) work(0, num_queries);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back(work, (uint32_t)((uint64_t)num_queries * t / nt), (uint32_t)((uint64_t)num_queries * (t + 1) / nt));
        for (auto& x : th) x.join();
    }
    return SLAMEM_OK;
}

// device -> host copy for a front end that fills its (pinned) read buffers from device memory (bench.py, tools): one DMA into
// page-locked memory instead of the runtime's staged copy of pageable memory
int slamem_copy_to_host(void* dst_host, const void* src_dev, uint64_t bytes) {
    if (bytes && (!dst_host || !src_dev)) { set_error("slamem_copy_to_host: null argument"); return SLAMEM_ERR_ARG; }
    if (bytes) SLAMEM_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return SLAMEM_OK;
}

int slamem_pinned_free(void* p) {
    if (p) SLAMEM_HIP(hipHostFree(p));
    return SLAMEM_OK;
}

int slamem_stream_destroy(slamem_stream* s) {
    if (!s) return SLAMEM_OK;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        // batches in flight finish first: never tear buffers down under a running kernel
        s->cv.wait(lk, [&] {
            for (int k = 0; k < s->nslots; k++)
                if (s->slot[k].state == QUEUED || s->slot[k].state == UPLOADED || s->slot[k].state == PREPARED ||
                    s->slot[k].state == LAUNCHED || s->slot[k].state == K8_ISSUED) return false;
            return true;
        });
        s->stop = true;
    }
    s->cv.notify_all();
    for (int k = 0; k < s->nthreads; k++)
        if (s->th[k].joinable()) s->th[k].join();
    (void)hipSetDevice(s->idx->device);
    // every copy and kernel of the stream's own HIP streams is through before buffers and streams go (the stages wait for their
    // work batch by batch; this is for whatever a failed batch left behind, and for tools that watch the copies: a profiler waited
    // 30 s at exit for completion callbacks of copies whose streams were destroyed under it, profiles/README.md)
    for (int k = 0; k < kThreads; k++)
        if (s->st[k]) (void)hipStreamSynchronize(s->st[k]);
    for (int k = 0; k < 3; k++)
        if (s->st_upx[k]) (void)hipStreamSynchronize(s->st_upx[k]);
    if (s->st_search2) (void)hipStreamSynchronize(s->st_search2);
    if (s->st_place) (void)hipStreamSynchronize(s->st_place);
    for (int k = 0; k < s->nslots; k++) free_slot(s->slot[k]);
    for (int k = 0; k < kThreads; k++)
        if (s->st[k]) (void)hipStreamDestroy(s->st[k]);
    for (int k = 0; k < 3; k++)
        if (s->st_upx[k]) (void)hipStreamDestroy(s->st_upx[k]);
    if (s->st_search2) (void)hipStreamDestroy(s->st_search2);
    if (s->st_place) (void)hipStreamDestroy(s->st_place);
    delete s;
    return SLAMEM_OK;
}

int slamem_stream_create(const slamem_index* idx, int slots, uint64_t max_batch_chars, uint32_t max_batch_queries,
                         int both_strands, int match_type, slamem_stream** out) {
    if (!idx || !out || slots < 2 || slots > kMaxSlots || max_batch_queries == 0 || (match_type != 0 && match_type != 1)) {
        set_error("slamem_stream_create: bad argument (2..8 slots, at least one query per batch, match type 0 or 1)");
        return SLAMEM_ERR_ARG;
    }
    *out = nullptr;
    SLAMEM_HIP(hipSetDevice(idx->device));
    slamem_stream* s = new (std::nothrow) slamem_stream();
    if (!s) { set_error("out of host memory"); return SLAMEM_ERR_NOMEM; }
    s->idx = idx;
    s->trace = getenv("SLAMEM_STREAM_TRACE") != nullptr;
    s->t0 = std::chrono::steady_clock::now();
    s->nslots = slots;
    s->both = both_strands ? 1 : 0;
    s->match_type = match_type;
    s->max_chars = max_batch_chars;
    s->max_q = max_batch_queries;
    int rc = SLAMEM_OK;
    {
        int lo = 0, hi = 0;  // (numerically lower = higher priority)
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char* v = getenv("SLAMEM_STREAM_CARRY");
        if (v) s->carry = atoi(v) != 0;
        v = getenv("SLAMEM_STREAM_SEARCH_STREAMS");
        const bool two = v && atoi(v) == 2 && !s->carry;
        v = getenv("SLAMEM_STREAM_PREP_PRIORITY");
        const bool prio = v ? atoi(v) != 0 : two;
        for (int k = 0; k < s->nthreads; k++) {
            // the preparation's workgroups go first where K8's waves retire (it is what the next K8 waits for)
            hipError_t e = (prio && k == T_PREP) ? hipStreamCreateWithPriority(&s->st[k], hipStreamNonBlocking, hi)
                                                 : hipStreamCreateWithFlags(&s->st[k], hipStreamNonBlocking);
            if (e != hipSuccess) { rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); break; }
        }
        // K9 on a stream of its own (SLAMEM_STREAM_PLACE_STREAM=1) was measured and lost: behind a K8 that holds every wave
        // slot its small kernels wait for the next tail (43-44 ms against 38.5)
        v = getenv("SLAMEM_STREAM_PLACE_STREAM");
        if (rc == SLAMEM_OK && v && atoi(v) != 0) {
            hipError_t e = hipStreamCreateWithFlags(&s->st_place, hipStreamNonBlocking);
            if (e != hipSuccess) rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
        }
        v = getenv("SLAMEM_STREAM_K8_WAVES");
        if (v && atoi(v) > 0) s->k8_waves = (uint32_t)atoi(v);
        // SLAMEM_STREAM_SEARCH_STREAMS=2 (measured at the end of round 3, not the default): even / odd batches on two search
        // streams, each K8 on 2560 of the chip's 4096 wave slots while another batch follows, the preparation on a high-priority
        // stream -- K8 (b+1) fills in as K8 (b) drains and the preparation of b+2 finds slots beside them.  Million-read batches
        // gain (headline stream 38.3 -> 37.5 ms, 30 M reads 109.0 -> 102.5 ms), two-million-read batches lose (102.3 -> 103.9 ms),
        // the command line does not move: profiles/r03_host_leg_two_streams.jsonl.
        if (rc == SLAMEM_OK && two) {
            hipError_t e = hipStreamCreateWithFlags(&s->st_search2, hipStreamNonBlocking);
            if (e != hipSuccess) rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
            else s->search_streams = 2;
        }
    }
    {
        const char* v = getenv("SLAMEM_STREAM_UPLOAD_SPLIT");
        if (v && atoi(v) >= 1 && atoi(v) <= 4) s->upload_split = atoi(v);
        for (int k = 0; rc == SLAMEM_OK && k + 1 < s->upload_split; k++) {
            hipError_t e = hipStreamCreateWithFlags(&s->st_upx[k], hipStreamNonBlocking);
            if (e != hipSuccess) rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
        }
    }
    if (rc != SLAMEM_OK) {
        for (int k = 0; k < 3; k++)
            if (s->st_upx[k]) (void)hipStreamDestroy(s->st_upx[k]);
        for (int k = 0; k < slots; k++) free_slot(s->slot[k]);
        for (int k = 0; k < kThreads; k++)
            if (s->st[k]) (void)hipStreamDestroy(s->st[k]);
        delete s;
        return rc;
    }
    for (int k = 0; k < s->nthreads; k++) s->th[k] = std::thread(worker, s, k);
    *out = s;
    return SLAMEM_OK;
}

int slamem_stream_submit(slamem_stream* s, const char* queries, const uint64_t* offsets, uint32_t num_queries, uint32_t min_len) {
    if (!s || !offsets || (num_queries && !queries)) { set_error("slamem_stream_submit: null argument"); return SLAMEM_ERR_ARG; }
    if (min_len < 1) { set_error("slamem_stream_submit: minimum MEM length must be >= 1"); return SLAMEM_ERR_ARG; }
    std::unique_lock<std::mutex> lk(s->mu);
    Slot& sl = s->slot[s->submitted % (uint64_t)s->nslots];
    if (sl.state != FREE) {
        // a slot is released by the slamem_stream_next call AFTER the one that handed its result out: at most slots - 1
        // batches may be in flight beside the result the caller is working on.  Never blocks (a caller that both submits
        // and collects on one thread would wait for itself).
        set_error("slamem_stream_submit: all %d slots are in use (collect a result with slamem_stream_next first)", s->nslots);
        return SLAMEM_ERR_ARG;
    }
    sl.seq = s->submitted;
    sl.chars = queries;
    sl.planes = nullptr;
    sl.other = nullptr;
    sl.offs = offsets;
    sl.nq = num_queries;
    sl.min_len = min_len;
    sl.rc = SLAMEM_OK;
    sl.err[0] = 0;
    sl.total = 0;
    sl.state = QUEUED;
    s->submitted++;
    lk.unlock();
    s->cv.notify_all();
    return SLAMEM_OK;
}

int slamem_stream_submit_packed(slamem_stream* s, const void* planes, const uint64_t* other, const uint64_t* offsets, uint32_t num_queries,
                                uint64_t num_units, uint32_t min_len) {
    if (!s || !offsets || (num_queries && !planes)) { set_error("slamem_stream_submit_packed: null argument"); return SLAMEM_ERR_ARG; }
    if (((uintptr_t)planes & 15u) != 0) { set_error("slamem_stream_submit_packed: planes must be 16-byte aligned"); return SLAMEM_ERR_ARG; }
    if (min_len < 1) { set_error("slamem_stream_submit_packed: minimum MEM length must be >= 1"); return SLAMEM_ERR_ARG; }
    std::unique_lock<std::mutex> lk(s->mu);
    Slot& sl = s->slot[s->submitted % (uint64_t)s->nslots];
    if (sl.state != FREE) {
        set_error("slamem_stream_submit_packed: all %d slots are in use (collect a result with slamem_stream_next first)", s->nslots);
        return SLAMEM_ERR_ARG;
    }
    sl.seq = s->submitted;
    sl.chars = nullptr;
    sl.planes = planes;
    sl.other = other;
    sl.units_hint = num_units;
    sl.offs = offsets;
    sl.nq = num_queries;
    sl.min_len = min_len;
    sl.rc = SLAMEM_OK;
    sl.err[0] = 0;
    sl.total = 0;
    sl.state = QUEUED;
    s->submitted++;
    lk.unlock();
    s->cv.notify_all();
    return SLAMEM_OK;
}

int slamem_stream_next(slamem_stream* s, const slamem_mem** mems_out, const uint64_t** block_offsets_out, uint64_t* total_out,
                       uint32_t* num_queries_out, slamem_timings* timings_out) {
    if (!s || !mems_out || !block_offsets_out || !total_out) { set_error("slamem_stream_next: null argument"); return SLAMEM_ERR_ARG; }
    std::unique_lock<std::mutex> lk(s->mu);
    // the result handed out by the previous call goes back to the pool
    if (s->returned > 0) {
        Slot& prev = s->slot[(s->returned - 1) % (uint64_t)s->nslots];
        if (prev.state == RETURNED) { prev.state = FREE; s->cv.notify_all(); }
    }
    if (s->returned == s->submitted) { set_error("slamem_stream_next: no batch is pending"); return SLAMEM_ERR_ARG; }
    Slot& sl = s->slot[s->returned % (uint64_t)s->nslots];
    s->cv.wait(lk, [&] { return sl.state == DONE; });
    sl.state = RETURNED;
    s->returned++;
    *mems_out = sl.h_mems;
    *block_offsets_out = sl.h_boff;
    *total_out = sl.total;
    if (num_queries_out) *num_queries_out = sl.nq;
    if (timings_out) *timings_out = sl.tm;
    if (sl.rc != SLAMEM_OK) set_error("%s", sl.err);
    return sl.rc;
}

}  // extern "C"
