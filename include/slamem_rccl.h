/* slamem_rccl.h -- multi-GPU companion of slamem_hip.h (libslamem_rccl.so): replicate a built index to the other
 * GPUs of the node with ONE RCCL broadcast of its arena over xGMI.  Separate library so that programs which bring
 * their own RCCL (torch.distributed in bench.py) never load a second copy.
 *
 * The reference is single-process, single-thread, one index per process (file-static globals, bwtindex.c:150-179);
 * there is nothing to replace here -- this is the scale-out step the GPU engine adds: queries
 * shard by record across GPUs with no data-path collective, the index is built once and replicated.
 */
#ifndef SLAMEM_RCCL_H
#define SLAMEM_RCCL_H

#include "slamem_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* devices[0] must be src's device.  out[0] = (slamem_index*)src (not copied); out[i>0] = a new index on
 * devices[i] that owns its arena (free each with slamem_index_free; out[0] is freed by its owner).
 * One single-process communicator (ncclCommInitAll), one ncclBroadcast per device inside a group, root = devices[0].
 * force_copy != 0 with num_devices == 1 broadcasts into a second arena on the same device (self-test of the RCCL
 * path on a one-GPU box): out[0] is then a new index that owns its arena. */
int slamem_index_replicate(const slamem_index *src, const int *devices, int num_devices, int force_copy,
                           slamem_index **out);

#ifdef __cplusplus
}
#endif
#endif
