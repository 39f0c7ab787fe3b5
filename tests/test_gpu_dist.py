"""The N > 1 path with the REAL engine: two ranks (spawned processes, `gloo` rendezvous on 127.0.0.1) share the one GPU of
the test box.  Rank 0 builds the index and broadcasts its arena (slamem_amd/shard.py::broadcast_arena), rank 1 attaches
the broadcast copy (Index.attach), both match their contiguous read range (shard_bounds: records are independent,
slamem.c:90-95) with slamem_find_mems_device, counts are all-gathered and the MEM rows gathered on rank 0 in rank order.
Workload: BASELINE.json configs[1] (100 Mbp reference, 1 M x 150 bp reads, forward, -l 20) -- the concatenation must be
the REAL reference's answer (2,412,288 MEMs, sum of lengths 133,301,375, sha256 9d583ab9...).
On an 8-GPU node only the transport (RCCL instead of gloo + staging) is new."""
import hashlib
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, n, nreads, L, min_len, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from slamem_amd import engine, shard
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cpu = torch.device("cpu")
        ref = engine.synth_reference(n, 42, dev)  # every rank generates its own reads from the text
        index = None
        src = None
        if rank == 0:
            index = engine.Index.build(ref, dev)
            src = index.arena_view().cpu()  # gloo moves host memory; over RCCL the view itself is the send buffer
        arena = shard.broadcast_arena(src, cpu, src=0)
        if rank != 0:
            index = engine.Index.attach(arena.to(dev))
        assert index.n == n
        bounds = shard.shard_bounds(np.arange(nreads + 1, dtype=np.uint64) * np.uint64(L), world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        count = hi - lo
        reads = engine.synth_reads(ref, lo, count, L, 0.02, 42, 0)
        offsets = torch.arange(count + 1, dtype=torch.int64, device=dev) * L
        m = index.matcher(count, False, 4 * count + 1024, count * L)
        total = m.run(reads, offsets, min_len)
        counts = shard.gather_counts(total, cpu)
        assert int(counts[rank]) == total
        rows = m.mems[:total].cpu()
        allrows = shard.gather_variable(rows, counts, dst=0)
        if rank == 0:
            assert allrows.shape[0] == int(counts.sum())
            np.save(out_path, allrows.numpy())
        dist.barrier()
        index.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_reference_answer(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    import torch.multiprocessing as mp
    n, nreads, L, min_len = 100_000_000, 1_000_000, 150, 20
    out = str(tmp_path / "rows.npy")
    mp.spawn(_rank, args=(2, _free_port(), n, nreads, L, min_len, out), nprocs=2, join=True)
    mems = np.load(out).view(np.uint32).astype(np.int64)
    assert mems.shape[0] == 2_412_288
    r, q, ln = mems[:, 0], mems[:, 1], mems[:, 2]
    assert int(ln.sum()) == 133_301_375 and int(ln.max()) == 150
    order = np.lexsort((ln, q, r))
    lines = b"".join(b"%d\t%d\t%d\n" % (r[i] + 1, q[i] + 1, ln[i]) for i in order)
    assert hashlib.sha256(lines).hexdigest().startswith("9d583ab9312e1698")
