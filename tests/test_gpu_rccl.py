"""The torch.distributed calls of the multi-GPU path on the REAL backend (nccl = RCCL) in a one-rank group: the test box
has one GPU, so this rehearses the calls bench.py makes at N > 1 -- group creation with device_id, the arena
broadcast (uint8, GBs in production), all_gather_into_tensor of the counts, MAX all-reduce of float64 timings,
barrier -- not the transport.  The N > 1 logic itself is covered on CPU by tests/test_dist_gloo.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_rehearsal(tmp_path):
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from slamem_amd import engine, shard
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="file://" + str(tmp_path / "rdzv"), rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(5)
        text = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=200_001))
        idx = engine.Index.build(text)
        arena = idx.export_arena()
        got = shard.broadcast_arena(arena.clone(), dev, src=0, force=True)
        assert torch.equal(got, arena)
        # what bench.py does at N > 1: the broadcast reads straight from the index arena (zero-copy view), in pieces
        view = idx.arena_view()
        assert view.data_ptr() != arena.data_ptr() and view.numel() == arena.numel()
        old_piece, shard.BROADCAST_PIECE = shard.BROADCAST_PIECE, 1 << 20
        try:
            same = shard.broadcast_arena(view, dev, src=0, force=True)
        finally:
            shard.BROADCAST_PIECE = old_piece
        assert same.data_ptr() == view.data_ptr() and torch.equal(same, arena)
        idx2 = engine.Index.attach(got)
        q = np.frombuffer(text[1000:1150], dtype=np.uint8)
        off = np.array([0, 150], dtype=np.uint64)
        m1, _ = idx.find_mems(q, off, 20, True)
        m2, _ = idx2.find_mems(q, off, 20, True)
        assert len(m1) >= 1 and np.array_equal(m1, m2)
        counts = shard.gather_counts(len(m1), dev, force=True)
        assert counts.tolist() == [len(m1)]
        el = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        assert float(el.item()) == 1.25
        dist.barrier()
        torch.cuda.synchronize(dev)
        idx2.close()
        idx.close()
    finally:
        dist.destroy_process_group()
