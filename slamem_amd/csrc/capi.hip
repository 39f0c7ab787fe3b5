// capi.hip -- the extern "C" surface of libslamem_hip.so (include/slamem_hip.h).
// Plain pointers and sizes only.  There is no CPU fallback anywhere in this library: without a
// usable gfx950 device every compute entry point fails with SLAMEM_ERR_NO_DEVICE / SLAMEM_ERR_HIP.
#include "common.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

namespace slamem {

static thread_local char g_err[512] = "";
static thread_local Timings g_tm;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    const char* base = strrchr(file, '/');
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, base ? base + 1 : file, line);
    if (e == hipErrorOutOfMemory) return SLAMEM_ERR_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return SLAMEM_ERR_NO_DEVICE;
    return SLAMEM_ERR_HIP;
}

Timings& thread_timings() { return g_tm; }
static thread_local bool g_want_stats = false;
static thread_local slamem_search_stats g_stats;
bool search_stats_wanted() { return g_want_stats; }
slamem_search_stats& last_search_stats() { return g_stats; }
static thread_local double g_clock[3];
double* last_search_clock() { return g_clock; }

int download_array(const slamem_index* idx, int which, void* host_dst, uint64_t count);
int sampled_lcp_stats(const slamem_index* idx, slamem_sslcp_stats* out);
int follow_letter_batch(const slamem_index*, const char*, uint32_t*, uint32_t*, uint32_t*, uint64_t, hipStream_t);
int enclosing_interval_batch(const slamem_index*, uint32_t*, uint32_t*, int32_t*, uint64_t, hipStream_t);
int position_in_text_batch(const slamem_index*, const uint32_t*, uint32_t*, uint64_t, hipStream_t);
int char_at_bwt_pos_batch(const slamem_index*, const uint32_t*, char*, uint64_t, hipStream_t);

static int check_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s): libslamem_hip has no CPU path", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return SLAMEM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (0..%d)", device, count - 1);
        return SLAMEM_ERR_ARG;
    }
    return SLAMEM_OK;
}

// Every section the kernels will index must lie inside the arena, in order, aligned for the paired 64-byte FM blocks:
// a truncated, stale or corrupted file / peer buffer is refused here instead of faulting on the GPU.
static int header_ok(const ArenaHeader& h, uint64_t bytes) {
    if (h.magic_lo != kArenaMagicLo || h.magic_hi != kArenaMagicHi || h.version != kArenaVersion) {
        set_error("not a slamem index arena (bad magic / version)");
        return SLAMEM_ERR_FORMAT;
    }
    if (h.total_bytes > bytes || h.total_bytes < kHeaderBytes) {
        set_error("index arena truncated: header says %llu bytes, got %llu", (unsigned long long)h.total_bytes,
                  (unsigned long long)bytes);
        return SLAMEM_ERR_FORMAT;
    }
    const uint64_t R = (uint64_t)h.n + 1;
    const char* bad = nullptr;
    uint64_t end = kHeaderBytes;  // end of the previous section
    auto section = [&](uint64_t off, uint64_t size, const char* name) {
        if (bad) return;
        if ((off & 127u) != 0 || off < end || off > h.total_bytes || size > h.total_bytes - off) bad = name;
        end = off + size;
    };
    if (h.n == 0 || h.n > 0xFFFFFFF0u) bad = "text length";
    if (!bad && h.nblocks != (uint32_t)((R + 1 + kFmRows - 1) >> kFmRowsLog2)) bad = "FM block count";
    if (!bad && h.off_fm != kHeaderBytes) bad = "FM block offset";
    section(h.off_fm, (uint64_t)h.nblocks * sizeof(FMBlock), "FM blocks");
    section(h.off_rec, (R + 1) * sizeof(RowRec), "row records");
    section(h.off_sa, R * 4, "suffix array");
    if (!bad && h.num_n > h.n) bad = "N row count";
    section(h.off_nrows, (uint64_t)(h.num_n ? h.num_n : 1) * 4, "N row list");
    if (h.off_kfilter) {
        if (!bad && (h.kfilter_log2 < 10 || h.kfilter_log2 > 40 || h.kfilter_k < 4 || h.kfilter_k > 26 ||
                     (h.kfilter_levels != 0u && h.kfilter_levels != 2u && h.kfilter_levels != 3u))) bad = "presence filter parameters";
        if (!bad) section(h.off_kfilter, 8ull << h.kfilter_log2, "presence filter");
    }
    if (h.off_tgrp || h.off_prec) {
        section(h.off_tgrp, ((R >> 4) + 2) * sizeof(TextGroup), "text groups");
        section(h.off_prec, R * sizeof(TextRec), "text-ordered records");
    }
    if (h.off_kjump) {
        if (!bad && (h.kjump_k < 1 || h.kjump_k > 12)) bad = "jump table K";
        if (!bad) section(h.off_kjump, 8ull << (2u * h.kjump_k), "jump table");
    }
    if (h.off_kbits) {
        if (!bad && (h.kbits_k < 8 || h.kbits_k > 16)) bad = "occurrence bitmap k";
        if (!bad) section(h.off_kbits, (1ull << (2u * h.kbits_k)) >> 3, "occurrence bitmap");
    }
    if (h.off_seed || h.off_tpl || h.off_spill) {
        const uint64_t units = text_units(h.n);
        if (!bad && (h.seed_k < 4 || h.seed_k > kSeedMaxK || h.seed_log2 < 10 || h.seed_log2 > 30 || 2u * h.seed_k < h.seed_log2 ||
                     2u * h.seed_k - h.seed_log2 > 7u)) bad = "seed table parameters";
        if (!bad) section(h.off_seed, sizeof(SeedBucket) << h.seed_log2, "seed table");
        section(h.off_tpl, units * sizeof(TextPlanes), "text units");
        if (!bad && (h.spill_cap != seed_spill_entries(h.n) || h.spill_used > h.spill_cap)) bad = "seed spill list size";
        section(h.off_spill, (uint64_t)h.spill_cap * 8, "seed spill list");
    }
    if (!bad && (h.off_tnm || h.off_tnb || h.off_tuq)) bad = "reserved section offsets";
    if (!bad && h.dollar_row > h.n) bad = "'$' row";
    if (bad) {
        set_error("index arena is corrupt or truncated: bad %s", bad);
        return SLAMEM_ERR_FORMAT;
    }
    return SLAMEM_OK;
}

}  // namespace slamem

using namespace slamem;

extern "C" {

int slamem_abi_version(void) { return SLAMEM_ABI_VERSION; }

const char* slamem_strerror(int code) {
    switch (code) {
    case SLAMEM_OK: return "ok";
    case SLAMEM_ERR_ARG: return "bad argument";
    case SLAMEM_ERR_HIP: return "HIP runtime error";
    case SLAMEM_ERR_NOMEM: return "out of memory";
    case SLAMEM_ERR_CAPACITY: return "output capacity too small";
    case SLAMEM_ERR_FORMAT: return "not a slamem index";
    case SLAMEM_ERR_IO: return "I/O error";
    case SLAMEM_ERR_NO_DEVICE: return "no usable GPU (there is no CPU fallback)";
    default: return "unknown error";
    }
}

const char* slamem_last_error_message(void) { return g_err; }

int slamem_device_count(int* count_out) {
    if (!count_out) return SLAMEM_ERR_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count_out = 0; return hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__); }
    *count_out = c;
    return SLAMEM_OK;
}

int slamem_device_warmup(int device) {
    SLAMEM_HIP(hipSetDevice(device));
    SLAMEM_HIP(hipFree(nullptr));
    return SLAMEM_OK;
}

int slamem_device_pci_bus_id(int device, char* out, int out_bytes) {
    if (!out || out_bytes < 16) return SLAMEM_ERR_ARG;
    int rc = check_device(device);
    if (rc) return rc;
    SLAMEM_HIP(hipDeviceGetPCIBusId(out, out_bytes, device));
    return SLAMEM_OK;
}

int slamem_get_timings(slamem_timings* out) {
    if (!out) return SLAMEM_ERR_ARG;
    *out = g_tm.t;
    return SLAMEM_OK;
}

int slamem_search_stats_enable(int on) {
    g_want_stats = on != 0;
    return SLAMEM_OK;
}

int slamem_get_search_stats(slamem_search_stats* out) {
    if (!out) return SLAMEM_ERR_ARG;
    *out = g_stats;
    return SLAMEM_OK;
}

int slamem_get_search_clock(double* us_to_empty_list, double* us_tail, double* us_wave_sum) {
    if (us_to_empty_list) *us_to_empty_list = g_clock[0];
    if (us_tail) *us_tail = g_clock[1];
    if (us_wave_sum) *us_wave_sum = g_clock[2];
    return SLAMEM_OK;
}

int slamem_reset_timings(void) {
    memset(&g_tm.t, 0, sizeof(g_tm.t));
    return SLAMEM_OK;
}

int slamem_device_mem_info(int device, uint64_t* free_out, uint64_t* total_out) {
    int rc = check_device(device);
    if (rc) return rc;
    SLAMEM_HIP(hipSetDevice(device));
    size_t f = 0, t = 0;
    SLAMEM_HIP(hipMemGetInfo(&f, &t));
    if (free_out) *free_out = f;
    if (total_out) *total_out = t;
    return SLAMEM_OK;
}

int slamem_index_build_bytes(uint32_t n, int layout, uint64_t* arena_bytes_out, uint64_t* peak_bytes_out) {
    return estimate_build_bytes(n, layout, arena_bytes_out, peak_bytes_out);
}

int slamem_index_build_device(const void* text_dev, uint32_t n, int device, void* stream, slamem_index** out) {
    return slamem_index_build_device_layout(text_dev, n, device, stream, SLAMEM_LAYOUT_AUTO, out);
}

int slamem_index_build_device_layout(const void* text_dev, uint32_t n, int device, void* stream, int layout,
                                     slamem_index** out) {
    int rc = check_device(device);
    if (rc) return rc;
    return build_index_device(text_dev, n, device, static_cast<hipStream_t>(stream), layout, out);
}

int slamem_index_build(const char* text_host, uint32_t n, int device, slamem_index** out) {
    return slamem_index_build_layout(text_host, n, device, SLAMEM_LAYOUT_AUTO, out);
}

int slamem_index_build_layout(const char* text_host, uint32_t n, int device, int layout, slamem_index** out) {
    if (!text_host || !out || n == 0) { set_error("slamem_index_build: empty text"); return SLAMEM_ERR_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    SLAMEM_HIP(hipSetDevice(device));
    void* d = nullptr;
    SLAMEM_HIP(hipMalloc(&d, (size_t)n + 16));
    hipError_t e = hipMemcpy(d, text_host, n, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e, "hipMemcpy(text)", __FILE__, __LINE__); }
    rc = build_index_device(d, n, device, nullptr, layout, out);
    (void)hipFree(d);
    return rc;
}

int slamem_index_free(slamem_index* idx) {
    if (!idx) return SLAMEM_OK;
    if (idx->owns_arena && idx->arena) {
        (void)hipSetDevice(idx->device);
        (void)hipFree(idx->arena);
    }
    free(idx);
    return SLAMEM_OK;
}

int slamem_index_get_info(const slamem_index* idx, slamem_index_info* out) {
    if (!idx || !out) return SLAMEM_ERR_ARG;
    out->text_length = idx->hdr.n;
    out->bwt_size = idx->hdr.n + 1;
    out->num_n_rows = idx->hdr.num_n;
    out->dollar_row = idx->hdr.dollar_row;
    out->max_lcp = idx->hdr.max_lcp;
    out->sort_rounds = idx->hdr.sort_rounds;
    out->arena_bytes = idx->arena_bytes;
    out->device = idx->device;
    out->owns_arena = idx->owns_arena;
    out->filter_k = idx->hdr.off_kfilter ? idx->hdr.kfilter_k : 0u;
    out->seed_k = idx->hdr.off_seed ? idx->hdr.seed_k : 0u;
    out->reserved1 = 0u;
    out->layout = idx->hdr.layout == 2u ? SLAMEM_LAYOUT_COMPACT : SLAMEM_LAYOUT_FULL;
    return SLAMEM_OK;
}

int slamem_index_arena(const slamem_index* idx, void** arena_dev_out, uint64_t* bytes_out) {
    if (!idx || !arena_dev_out || !bytes_out) return SLAMEM_ERR_ARG;
    *arena_dev_out = idx->arena;
    *bytes_out = idx->arena_bytes;
    return SLAMEM_OK;
}

int slamem_index_export(const slamem_index* idx, void* dst_dev, uint64_t dst_bytes, void* stream) {
    if (!idx || !dst_dev) return SLAMEM_ERR_ARG;
    if (dst_bytes < idx->arena_bytes) { set_error("slamem_index_export: destination too small"); return SLAMEM_ERR_ARG; }
    SLAMEM_HIP(hipSetDevice(idx->device));
    SLAMEM_HIP(hipMemcpyAsync(dst_dev, idx->arena, idx->arena_bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    SLAMEM_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return SLAMEM_OK;
}

int slamem_index_attach(void* arena_dev, uint64_t bytes, int device, slamem_index** out) {
    if (!arena_dev || !out || bytes < kHeaderBytes) { set_error("slamem_index_attach: bad arena"); return SLAMEM_ERR_ARG; }
    if (((uintptr_t)arena_dev & 127u) != 0) { set_error("slamem_index_attach: the arena must be 128-byte aligned"); return SLAMEM_ERR_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    SLAMEM_HIP(hipSetDevice(device));
    ArenaHeader h;
    SLAMEM_HIP(hipMemcpy(&h, arena_dev, sizeof(h), hipMemcpyDeviceToHost));
    rc = header_ok(h, bytes);
    if (rc) return rc;
    slamem_index* idx = static_cast<slamem_index*>(calloc(1, sizeof(slamem_index)));
    if (!idx) { set_error("out of host memory"); return SLAMEM_ERR_NOMEM; }
    idx->hdr = h;
    idx->arena = arena_dev;
    idx->arena_bytes = h.total_bytes;
    idx->device = device;
    idx->owns_arena = 0;
    make_view(idx);
    *out = idx;
    return SLAMEM_OK;
}

int slamem_index_adopt_arena(slamem_index* idx) {
    if (!idx) return SLAMEM_ERR_ARG;
    idx->owns_arena = 1;
    return SLAMEM_OK;
}

int slamem_index_save(const slamem_index* idx, const char* path) {
    if (!idx || !path) return SLAMEM_ERR_ARG;
    SLAMEM_HIP(hipSetDevice(idx->device));
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("cannot create <%s>", path); return SLAMEM_ERR_IO; }
    const uint64_t chunk = 64ull << 20;
    char* buf = static_cast<char*>(malloc(chunk));
    if (!buf) { fclose(f); set_error("out of host memory"); return SLAMEM_ERR_NOMEM; }
    int rc = SLAMEM_OK;
    for (uint64_t off = 0; off < idx->arena_bytes; off += chunk) {
        uint64_t len = idx->arena_bytes - off < chunk ? idx->arena_bytes - off : chunk;
        hipError_t e = hipMemcpy(buf, static_cast<char*>(idx->arena) + off, len, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpy(save)", __FILE__, __LINE__); break; }
        if (fwrite(buf, 1, len, f) != len) { set_error("short write to <%s>", path); rc = SLAMEM_ERR_IO; break; }
    }
    free(buf);
    if (fclose(f) != 0 && rc == SLAMEM_OK) { set_error("cannot close <%s>", path); rc = SLAMEM_ERR_IO; }
    return rc;
}

int slamem_index_validate_header(const void* header, uint64_t header_bytes, uint64_t available_bytes) {
    if (!header || header_bytes < sizeof(ArenaHeader)) { set_error("slamem_index_validate_header: need the first %zu bytes", sizeof(ArenaHeader)); return SLAMEM_ERR_ARG; }
    ArenaHeader h;
    memcpy(&h, header, sizeof(h));
    return header_ok(h, available_bytes);
}

int slamem_index_load(const char* path, int device, slamem_index** out) {
    if (!path || !out) return SLAMEM_ERR_ARG;
    int rc;
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("cannot open <%s>", path); return SLAMEM_ERR_IO; }
    ArenaHeader h;
    if (fread(&h, 1, sizeof(h), f) != sizeof(h)) { fclose(f); set_error("<%s> is too short", path); return SLAMEM_ERR_FORMAT; }
    uint64_t file_bytes = 0;
    if (fseeko(f, 0, SEEK_END) == 0) { off_t e2 = ftello(f); if (e2 > 0) file_bytes = (uint64_t)e2; }
    rc = header_ok(h, file_bytes);  // before any device call: a bad file is reported as such on any machine
    if (rc == SLAMEM_OK) rc = check_device(device);
    if (rc) { fclose(f); return rc; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { fclose(f); return hip_fail(e, "hipSetDevice", __FILE__, __LINE__); }
    void* arena = nullptr;
    e = hipMalloc(&arena, h.total_bytes);
    if (e != hipSuccess) { fclose(f); return hip_fail(e, "hipMalloc(load)", __FILE__, __LINE__); }
    const uint64_t chunk = 64ull << 20;
    char* buf = static_cast<char*>(malloc(chunk));
    rewind(f);
    rc = buf ? SLAMEM_OK : SLAMEM_ERR_NOMEM;
    for (uint64_t off = 0; rc == SLAMEM_OK && off < h.total_bytes; off += chunk) {
        uint64_t len = h.total_bytes - off < chunk ? h.total_bytes - off : chunk;
        if (fread(buf, 1, len, f) != len) { set_error("<%s> is truncated", path); rc = SLAMEM_ERR_FORMAT; break; }
        e = hipMemcpy(static_cast<char*>(arena) + off, buf, len, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(load)", __FILE__, __LINE__);
    }
    free(buf);
    fclose(f);
    if (rc) { (void)hipFree(arena); return rc; }
    slamem_index* idx = static_cast<slamem_index*>(calloc(1, sizeof(slamem_index)));
    if (!idx) { (void)hipFree(arena); set_error("out of host memory"); return SLAMEM_ERR_NOMEM; }
    idx->hdr = h;
    idx->arena = arena;
    idx->arena_bytes = h.total_bytes;
    idx->device = device;
    idx->owns_arena = 1;
    make_view(idx);
    *out = idx;
    return SLAMEM_OK;
}

int slamem_index_download(const slamem_index* idx, int which, void* host_dst, uint64_t count) {
    if (!idx || !host_dst) return SLAMEM_ERR_ARG;
    return download_array(idx, which, host_dst, count);
}

int slamem_index_sampled_lcp_stats(const slamem_index* idx, slamem_sslcp_stats* out) {
    if (!idx || !out) return SLAMEM_ERR_ARG;
    return sampled_lcp_stats(idx, out);
}

int slamem_follow_letter_batch(const slamem_index* idx, const char* letters_dev, uint32_t* top_dev, uint32_t* bottom_dev,
                               uint32_t* size_out_dev, uint64_t count, void* stream) {
    if (!idx || (count && (!letters_dev || !top_dev || !bottom_dev || !size_out_dev))) return SLAMEM_ERR_ARG;
    return follow_letter_batch(idx, letters_dev, top_dev, bottom_dev, size_out_dev, count, static_cast<hipStream_t>(stream));
}

int slamem_enclosing_interval_batch(const slamem_index* idx, uint32_t* top_dev, uint32_t* bottom_dev,
                                    int32_t* depth_out_dev, uint64_t count, void* stream) {
    if (!idx || (count && (!top_dev || !bottom_dev || !depth_out_dev))) return SLAMEM_ERR_ARG;
    return enclosing_interval_batch(idx, top_dev, bottom_dev, depth_out_dev, count, static_cast<hipStream_t>(stream));
}

int slamem_position_in_text_batch(const slamem_index* idx, const uint32_t* rows_dev, uint32_t* pos_out_dev,
                                  uint64_t count, void* stream) {
    if (!idx || (count && (!rows_dev || !pos_out_dev))) return SLAMEM_ERR_ARG;
    return position_in_text_batch(idx, rows_dev, pos_out_dev, count, static_cast<hipStream_t>(stream));
}

int slamem_char_at_bwt_pos_batch(const slamem_index* idx, const uint32_t* rows_dev, char* chars_out_dev, uint64_t count,
                                 void* stream) {
    if (!idx || (count && (!rows_dev || !chars_out_dev))) return SLAMEM_ERR_ARG;
    return char_at_bwt_pos_batch(idx, rows_dev, chars_out_dev, count, static_cast<hipStream_t>(stream));
}

int slamem_find_mems_workspace_bytes(uint32_t num_queries, int both_strands, uint64_t query_bytes, uint64_t mems_capacity,
                                     uint64_t* bytes_out) {
    if (!bytes_out) return SLAMEM_ERR_ARG;
    *bytes_out = find_mems_workspace_bytes(num_queries, both_strands, query_bytes, mems_capacity);
    return SLAMEM_OK;
}

int slamem_find_mems_device(const slamem_index* idx, const void* queries_dev, const uint64_t* offsets_dev,
                            uint32_t num_queries, uint64_t query_bytes, uint32_t min_len, int both_strands,
                            slamem_mem* mems_dev, uint64_t mems_capacity, uint64_t* block_offsets_dev, void* workspace_dev,
                            uint64_t workspace_bytes, void* stream, uint64_t* total_out) {
    return find_mems_device(idx, queries_dev, offsets_dev, num_queries, query_bytes, min_len, both_strands, 0, mems_dev,
                            mems_capacity, block_offsets_dev, workspace_dev, workspace_bytes, static_cast<hipStream_t>(stream),
                            total_out);
}

int slamem_find_mams_device(const slamem_index* idx, const void* queries_dev, const uint64_t* offsets_dev,
                            uint32_t num_queries, uint64_t query_bytes, uint32_t min_len, int both_strands,
                            slamem_mem* mems_dev, uint64_t mems_capacity, uint64_t* block_offsets_dev, void* workspace_dev,
                            uint64_t workspace_bytes, void* stream, uint64_t* total_out) {
    return find_mems_device(idx, queries_dev, offsets_dev, num_queries, query_bytes, min_len, both_strands, 1, mems_dev,
                            mems_capacity, block_offsets_dev, workspace_dev, workspace_bytes, static_cast<hipStream_t>(stream),
                            total_out);
}

void slamem_host_free(void* p) { free(p); }

static int find_matches_host(const slamem_index* idx, const char* queries, const uint64_t* offsets, uint32_t num_queries,
                             uint32_t min_len, int both_strands, int match_type, slamem_mem** mems_out,
                             uint64_t** block_offsets_out, uint64_t* total_out);

int slamem_find_mems_host(const slamem_index* idx, const char* queries, const uint64_t* offsets, uint32_t num_queries,
                          uint32_t min_len, int both_strands, slamem_mem** mems_out, uint64_t** block_offsets_out,
                          uint64_t* total_out) {
    return find_matches_host(idx, queries, offsets, num_queries, min_len, both_strands, 0, mems_out, block_offsets_out, total_out);
}

int slamem_find_mams_host(const slamem_index* idx, const char* queries, const uint64_t* offsets, uint32_t num_queries,
                          uint32_t min_len, int both_strands, slamem_mem** mems_out, uint64_t** block_offsets_out,
                          uint64_t* total_out) {
    return find_matches_host(idx, queries, offsets, num_queries, min_len, both_strands, 1, mems_out, block_offsets_out, total_out);
}

static int find_matches_host(const slamem_index* idx, const char* queries, const uint64_t* offsets, uint32_t num_queries,
                             uint32_t min_len, int both_strands, int match_type, slamem_mem** mems_out,
                             uint64_t** block_offsets_out, uint64_t* total_out) {
    if (!idx || !offsets || !mems_out || !block_offsets_out || !total_out || (num_queries && !queries)) {
        set_error("slamem_find_mems_host: null argument");
        return SLAMEM_ERR_ARG;
    }
    SLAMEM_HIP(hipSetDevice(idx->device));
    const uint64_t qbytes = offsets[num_queries];
    const uint64_t num_blocks = (uint64_t)num_queries * (both_strands ? 2 : 1);
    void *d_q = nullptr, *d_off = nullptr, *d_boff = nullptr, *d_mems = nullptr, *d_ws = nullptr;
    slamem_mem* h_mems = nullptr;
    uint64_t* h_boff = nullptr;
    int rc = SLAMEM_OK;
    uint64_t cap = qbytes / 16 + 4 * (uint64_t)num_blocks + 1024;  // first guess; grown on SLAMEM_ERR_CAPACITY
    hipError_t e;
#define HOST_TRY(call) if ((e = (call)) != hipSuccess) { rc = hip_fail(e, #call, __FILE__, __LINE__); goto done; }
    HOST_TRY(hipMalloc(&d_q, qbytes + 16));
    HOST_TRY(hipMalloc(&d_off, ((uint64_t)num_queries + 1) * 8));
    HOST_TRY(hipMalloc(&d_boff, (num_blocks + 1) * 8));
    HOST_TRY(hipMemcpy(d_q, queries, qbytes, hipMemcpyHostToDevice));
    HOST_TRY(hipMemcpy(d_off, offsets, ((uint64_t)num_queries + 1) * 8, hipMemcpyHostToDevice));
    for (int attempt = 0; attempt < 3; attempt++) {
        uint64_t ws_bytes = find_mems_workspace_bytes(num_queries, both_strands, qbytes, cap);
        HOST_TRY(hipMalloc(&d_mems, cap * sizeof(slamem_mem) + 16));
        HOST_TRY(hipMalloc(&d_ws, ws_bytes));
        rc = find_mems_device(idx, d_q, static_cast<const uint64_t*>(d_off), num_queries, qbytes, min_len, both_strands,
                              match_type, static_cast<slamem_mem*>(d_mems), cap, static_cast<uint64_t*>(d_boff), d_ws, ws_bytes,
                              nullptr, total_out);
        if (rc != SLAMEM_ERR_CAPACITY) break;
        (void)hipFree(d_mems); d_mems = nullptr;
        (void)hipFree(d_ws); d_ws = nullptr;
        cap = *total_out;
    }
    if (rc) goto done;
    h_mems = static_cast<slamem_mem*>(malloc((*total_out ? *total_out : 1) * sizeof(slamem_mem)));
    h_boff = static_cast<uint64_t*>(malloc((num_blocks + 1) * 8));
    if (!h_mems || !h_boff) { set_error("out of host memory"); rc = SLAMEM_ERR_NOMEM; goto done; }
    if (*total_out) HOST_TRY(hipMemcpy(h_mems, d_mems, *total_out * sizeof(slamem_mem), hipMemcpyDeviceToHost));
    HOST_TRY(hipMemcpy(h_boff, d_boff, (num_blocks + 1) * 8, hipMemcpyDeviceToHost));
    *mems_out = h_mems;
    *block_offsets_out = h_boff;
    h_mems = nullptr;
    h_boff = nullptr;
done:
#undef HOST_TRY
    free(h_mems);
    free(h_boff);
    if (d_q) (void)hipFree(d_q);
    if (d_off) (void)hipFree(d_off);
    if (d_boff) (void)hipFree(d_boff);
    if (d_mems) (void)hipFree(d_mems);
    if (d_ws) (void)hipFree(d_ws);
    return rc;
}

}  // extern "C"
