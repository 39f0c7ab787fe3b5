// replicate_rccl.hip -- libslamem_rccl.so: index replication across the GPUs of one node with RCCL (include/slamem_rccl.h).
// The index is ONE contiguous arena, so replication is one ncclBroadcast per peer inside one group call; xGMI is
// point-to-point (7 links per GPU), so the root feeds the peers over distinct links.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../include/slamem_rccl.h"

namespace {
thread_local char g_msg[256];
}

extern "C" const char* slamem_rccl_last_error(void) { return g_msg; }

#define RFAIL(code, ...) do { snprintf(g_msg, sizeof(g_msg), __VA_ARGS__); rc = (code); goto done; } while (0)

extern "C" int slamem_index_replicate(const slamem_index* src, const int* devices, int n, int force_copy, slamem_index** out) {
    if (!src || !devices || !out || n < 1) return SLAMEM_ERR_ARG;
    slamem_index_info info;
    void* src_arena = nullptr;
    uint64_t bytes = 0;
    int rc = slamem_index_get_info(src, &info);
    if (rc) return rc;
    rc = slamem_index_arena(src, &src_arena, &bytes);
    if (rc) return rc;
    if (devices[0] != info.device) { snprintf(g_msg, sizeof(g_msg), "devices[0] must be the source index's device"); return SLAMEM_ERR_ARG; }
    if (n == 1 && !force_copy) { out[0] = const_cast<slamem_index*>(src); return SLAMEM_OK; }

    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    std::vector<hipStream_t> streams((size_t)n, nullptr);
    std::vector<void*> arenas((size_t)n, nullptr);
    bool comms_ok = false;
    ncclResult_t nr;
    for (int i = 0; i < n; i++) out[i] = nullptr;

    for (int i = 0; i < n; i++) {
        if (hipSetDevice(devices[i]) != hipSuccess) RFAIL(SLAMEM_ERR_NO_DEVICE, "hipSetDevice(%d) failed", devices[i]);
        if (hipStreamCreate(&streams[(size_t)i]) != hipSuccess) RFAIL(SLAMEM_ERR_HIP, "hipStreamCreate failed on device %d", devices[i]);
        if (i > 0 || force_copy) {
            if (hipMalloc(&arenas[(size_t)i], bytes) != hipSuccess) RFAIL(SLAMEM_ERR_NOMEM, "cannot allocate %llu bytes on device %d", (unsigned long long)bytes, devices[i]);
        } else {
            arenas[(size_t)i] = src_arena;
        }
    }
    nr = ncclCommInitAll(comms.data(), n, devices);
    if (nr != ncclSuccess) RFAIL(SLAMEM_ERR_HIP, "ncclCommInitAll: %s", ncclGetErrorString(nr));
    comms_ok = true;
    nr = ncclGroupStart();
    if (nr != ncclSuccess) RFAIL(SLAMEM_ERR_HIP, "ncclGroupStart: %s", ncclGetErrorString(nr));
    for (int i = 0; i < n; i++) {
        const void* send = i == 0 ? src_arena : arenas[(size_t)i];
        nr = ncclBroadcast(send, arenas[(size_t)i], bytes, ncclChar, 0, comms[(size_t)i], streams[(size_t)i]);
        if (nr != ncclSuccess) { (void)ncclGroupEnd(); RFAIL(SLAMEM_ERR_HIP, "ncclBroadcast: %s", ncclGetErrorString(nr)); }
    }
    nr = ncclGroupEnd();
    if (nr != ncclSuccess) RFAIL(SLAMEM_ERR_HIP, "ncclGroupEnd: %s", ncclGetErrorString(nr));
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(devices[i]);
        if (hipStreamSynchronize(streams[(size_t)i]) != hipSuccess) RFAIL(SLAMEM_ERR_HIP, "broadcast failed on device %d", devices[i]);
    }
    for (int i = 0; i < n; i++) {
        if (i == 0 && !force_copy) { out[0] = const_cast<slamem_index*>(src); continue; }
        slamem_index* idx = nullptr;
        rc = slamem_index_attach(arenas[(size_t)i], bytes, devices[i], &idx);
        if (rc) { snprintf(g_msg, sizeof(g_msg), "attach on device %d: %s", devices[i], slamem_last_error_message()); goto done; }
        rc = slamem_index_adopt_arena(idx);  // the handle now owns (and frees) the arena
        if (rc) { slamem_index_free(idx); goto done; }
        arenas[(size_t)i] = nullptr;
        out[i] = idx;
    }
    rc = SLAMEM_OK;
done:
    if (comms_ok) for (int i = 0; i < n; i++) if (comms[(size_t)i]) (void)ncclCommDestroy(comms[(size_t)i]);
    for (int i = 0; i < n; i++) {
        if (streams[(size_t)i]) { (void)hipSetDevice(devices[i]); (void)hipStreamDestroy(streams[(size_t)i]); }
        if (rc != SLAMEM_OK) {
            if (out[i] && !(i == 0 && !force_copy)) { slamem_index_free(out[i]); }
            out[i] = nullptr;
            if (arenas[(size_t)i] && !(i == 0 && !force_copy)) { (void)hipSetDevice(devices[i]); (void)hipFree(arenas[(size_t)i]); }
        }
    }
    return rc;
}
