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


def test_replicate_on_two_ranks_of_one_device_or_a_clean_refusal():
    """slamem_index_replicate (libslamem_rccl.so: ncclCommInitAll + one grouped ncclBroadcast per device, what SLAMEM_GPUS=N runs)
    with n = 2 and devices = {0, 0}: the only way to take the N > 1 branch of that function on a one-GPU box.  RCCL either
    accepts two ranks on one device -- then both copies must answer like the source -- or refuses the communicator (duplicate
    device): then the call must fail with its error code and message, hand back no index and leak no arena.  Run in a child
    process: the library brings /opt/rocm's RCCL, this process may hold torch's."""
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import ctypes as C, os, sys
import numpy as np
root = sys.argv[1]
hip = C.CDLL(os.path.join(root, "slamem_amd", "csrc", "libslamem_hip.so"), mode=C.RTLD_GLOBAL)
rc = C.CDLL(os.path.join(root, "slamem_amd", "csrc", "libslamem_rccl.so"))
rc.slamem_rccl_last_error.restype = C.c_char_p
hip.slamem_last_error_message.restype = C.c_char_p
rng = np.random.default_rng(9)
text = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300_001).tobytes()
idx = C.c_void_p()
assert hip.slamem_index_build(text, len(text), 0, C.byref(idx)) == 0, hip.slamem_last_error_message()
free0, total = C.c_uint64(), C.c_uint64()
hip.slamem_device_mem_info(0, C.byref(free0), C.byref(total))
devs = (C.c_int * 2)(0, 0)
out = (C.c_void_p * 2)()
code = rc.slamem_index_replicate(idx, devs, 2, 0, out)
if code == 0:
    assert out[0] == idx.value and out[1] and out[1] != idx.value
    info = (C.c_uint32 * 16)()
    print("accepted: two ranks on one device")
    assert hip.slamem_index_free(C.c_void_p(out[1])) == 0
else:
    msg = rc.slamem_rccl_last_error().decode()
    assert code in (2, 7) and ("nccl" in msg or "hip" in msg.lower()), (code, msg)
    assert not out[0] and not out[1]
    free1 = C.c_uint64()
    hip.slamem_device_mem_info(0, C.byref(free1), C.byref(total))
    assert free1.value + (64 << 20) >= free0.value, (free0.value, free1.value)  # the second arena was given back
    print("refused:", code, msg)
assert hip.slamem_index_free(idx) == 0
"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", child, root], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-1500:] + r.stderr.decode()[-3000:]
    print(r.stdout.decode().strip().splitlines()[-1])
