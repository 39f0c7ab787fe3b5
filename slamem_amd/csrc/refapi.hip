// refapi.hip -- include/slamem_refapi.h: the reference-named index functions over the C ABI of libslamem_hip.so.
// One element per call through the batched entry points (a launch and two small copies each): a compatibility layer, see the
// header.  Nothing here computes on the CPU: every answer comes from the index in HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/slamem_hip.h"
#include "../../include/slamem_refapi.h"

namespace {
slamem_index* g_idx = nullptr;  // one index per process, as in bwtindex.c:150-179
uint32_t g_rows = 0;            // BWT size = n + 1
struct Scratch { uint32_t top, bot, out; int32_t depth; char letter, ch; };
Scratch* g_dev = nullptr;       // the operands of one call, in device memory

[[noreturn]] void die(const char* what) {
    printf("> ERROR: %s (%s)\n", what, slamem_last_error_message());
    exit(-1);
}
void need(int rc, const char* what) { if (rc != SLAMEM_OK) die(what); }
void need_hip(hipError_t e, const char* what) { if (e != hipSuccess) { printf("> ERROR: %s (%s)\n", what, hipGetErrorString(e)); exit(-1); } }
void up(const Scratch& s) { need_hip(hipMemcpy(g_dev, &s, sizeof(s), hipMemcpyHostToDevice), "copy to the GPU"); }
Scratch down() { Scratch s; need_hip(hipMemcpy(&s, g_dev, sizeof(s), hipMemcpyDeviceToHost), "copy from the GPU"); return s; }
void have_index() { if (!g_idx) { printf("> ERROR: no index built\n"); exit(-1); } }
}  // namespace

extern "C" {

void FMI_BuildIndex(char** texts, unsigned int* sizes, unsigned int numTexts, unsigned char** lcpOut, char verbose) {
    (void)verbose;
    if (numTexts != 1 || !texts || !sizes) { printf("> ERROR: one text expected\n"); exit(-1); }
    const char* dv = getenv("SLAMEM_DEVICE");
    const int device = dv ? atoi(dv) : 0;
    if (g_idx) FMI_FreeIndex();
    need(slamem_index_build(texts[0], sizes[0], device, &g_idx), "index construction on the GPU");
    slamem_index_info info;
    need(slamem_index_get_info(g_idx, &info), "index information");
    g_rows = info.bwt_size;
    need_hip(hipSetDevice(device), "hipSetDevice");
    need_hip(hipMalloc(reinterpret_cast<void**>(&g_dev), sizeof(Scratch)), "hipMalloc");
    if (lcpOut) {  // the byte array of bwtindex.c:1094-1304: min(LCP, 255) for rows 0..n (row 0: 0)
        const uint64_t count = (uint64_t)g_rows + 1;  // rows 0 .. n+1, the sentinels included
        int32_t* l = static_cast<int32_t*>(malloc(count * sizeof(int32_t)));
        unsigned char* out = static_cast<unsigned char*>(malloc(g_rows));
        if (!l || !out) { printf("> ERROR: Not enough memory\n"); exit(-1); }
        need(slamem_index_download(g_idx, SLAMEM_ARRAY_LCP, l, count), "LCP download");
        for (uint32_t i = 0; i < g_rows; i++) out[i] = (unsigned char)(l[i] < 0 ? 0 : l[i] > 255 ? 255 : l[i]);
        free(l);
        *lcpOut = out;
    }
}

int BuildSampledLCPArray(char* text, unsigned int n, unsigned char* lcp, int minlcp, int verbose) {
    (void)text; (void)lcp; (void)minlcp; (void)verbose;
    have_index();
    if (g_rows != n + 1) { printf("> ERROR: index not built for this text\n"); exit(-1); }
    slamem_sslcp_stats st;
    need(slamem_index_sampled_lcp_stats(g_idx, &st), "LCP sampling statistics");
    return (int)st.num_samples;
}

unsigned int FMI_GetBWTSize(void) { have_index(); return g_rows; }
unsigned int FMI_GetTextSize(void) { have_index(); return g_rows - 1; }

unsigned int FMI_FollowLetter(char c, unsigned int* top, unsigned int* bottom) {
    have_index();
    Scratch s = {};
    s.top = *top;
    s.bot = *bottom >= g_rows ? g_rows - 1 : *bottom;  // the reference's initial bottom = n+1 (slamem.c:111)
    s.letter = c;
    up(s);
    need(slamem_follow_letter_batch(g_idx, &g_dev->letter, &g_dev->top, &g_dev->bot, &g_dev->out, 1, nullptr), "FMI_FollowLetter");
    need_hip(hipDeviceSynchronize(), "FMI_FollowLetter");
    s = down();
    if (s.out) { *top = s.top; *bottom = s.bot; }
    return s.out;
}

int GetEnclosingLCPInterval(unsigned int* top, unsigned int* bottom) {
    have_index();
    Scratch s = {};
    s.top = *top;
    s.bot = *bottom >= g_rows ? g_rows - 1 : *bottom;
    up(s);
    need(slamem_enclosing_interval_batch(g_idx, &g_dev->top, &g_dev->bot, &g_dev->depth, 1, nullptr), "GetEnclosingLCPInterval");
    need_hip(hipDeviceSynchronize(), "GetEnclosingLCPInterval");
    s = down();
    *top = s.top; *bottom = s.bot;
    return s.depth;
}

char FMI_GetCharAtBWTPos(unsigned int bwtpos) {
    have_index();
    Scratch s = {};
    s.top = bwtpos;
    up(s);
    need(slamem_char_at_bwt_pos_batch(g_idx, &g_dev->top, &g_dev->ch, 1, nullptr), "FMI_GetCharAtBWTPos");
    need_hip(hipDeviceSynchronize(), "FMI_GetCharAtBWTPos");
    return down().ch;
}

unsigned int FMI_PositionInText(unsigned int bwtpos) {
    have_index();
    Scratch s = {};
    s.top = bwtpos;
    up(s);
    need(slamem_position_in_text_batch(g_idx, &g_dev->top, &g_dev->out, 1, nullptr), "FMI_PositionInText");
    need_hip(hipDeviceSynchronize(), "FMI_PositionInText");
    return down().out;
}

void FMI_FreeIndex(void) {
    if (g_dev) { (void)hipFree(g_dev); g_dev = nullptr; }
    if (g_idx) { slamem_index_free(g_idx); g_idx = nullptr; }
    g_rows = 0;
}

void FreeSampledSuffixArray(void) {}  // the parent structure is part of the index (freed with it)

}  // extern "C"
