"""N>1 host logic on CPU: two gloo ranks shard the query records, replicate the index arena by broadcast,
match their shards and gather counts + MEM arrays on rank 0.  The compute stand-in is the oracle (this is
a test of slamem_amd/shard.py's plumbing, which bench.py runs unchanged over RCCL with the HIP engine)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slamem_amd import shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ref, reads, L, min_len, both, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as po
        dev = torch.device("cpu")
        # "arena": on the GPU this is the index arena; here the stand-in index is rebuilt from the broadcast text
        arena = torch.from_numpy(ref.copy()) if rank == 0 else None
        arena = shard.broadcast_arena(arena, dev, src=0)
        assert arena.numel() == ref.shape[0] and bytes(arena.numpy().tobytes()) == ref.tobytes()
        idx = po.OracleIndex(arena.numpy().tobytes())
        n = reads.shape[0] // L
        offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
        bounds = shard.shard_bounds(offsets, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        sub_off = offsets[lo:hi + 1] - offsets[lo]
        mems, bc = idx.match_batch(reads[lo * L:hi * L], sub_off, min_len, both)
        counts = shard.gather_counts(len(mems), dev)
        assert int(counts[rank]) == len(mems)
        rows = torch.from_numpy(np.stack([mems["ref_pos"], mems["query_pos"], mems["length"]], axis=1).astype(np.int32)
                                if len(mems) else np.zeros((0, 3), dtype=np.int32))
        allrows = shard.gather_variable(rows, counts, dst=0)
        if rank == 0:
            np.save(out_path, allrows.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_matching_equals_single_process(tmp_path, world):
    from oracle import pyoracle as po
    L, nreads, min_len, both = 60, 101, 12, True
    ref = synth.make_reference(20000, seed=5)
    reads = synth.make_reads(ref, 0, nreads, L, 0.03, seed=5, rc_percent=50).reshape(-1)
    out = str(tmp_path / "rows.npy")
    mp.spawn(_worker, args=(world, _free_port(), ref, reads, L, min_len, both, out), nprocs=world, join=True)
    got = np.load(out)
    idx = po.OracleIndex(ref.tobytes())
    offsets = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(L)
    mems, _ = idx.match_batch(reads, offsets, min_len, both)
    exp = np.stack([mems["ref_pos"], mems["query_pos"], mems["length"]], axis=1).astype(np.int32)
    assert np.array_equal(got, exp)  # contiguous shards concatenated in rank order restore the input order


def test_shard_bounds_balance_bases():
    offsets = np.concatenate([[0], np.cumsum([10] * 50 + [1000] * 5 + [10] * 45)]).astype(np.uint64)
    b = shard.shard_bounds(offsets, 4)
    assert b[0] == 0 and b[-1] == 100 and all(np.diff(b) >= 0)
    sizes = [int(offsets[b[i + 1]] - offsets[b[i]]) for i in range(4)]
    assert max(sizes) - min(sizes) <= 1000
    assert list(shard.shard_bounds(np.array([0, 5], dtype=np.uint64), 3)) == [0, 0, 0, 1] or \
        shard.shard_bounds(np.array([0, 5], dtype=np.uint64), 3)[-1] == 1


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus N` (no WORLD_SIZE) must start N ranks itself, as a child of a process that has not
    touched the GPU, and must refuse to report a smaller job (VERDICT r1 / ADVICE r1).  Checked here up to the device
    check: the command line it would run, and the refusal when fewer devices than ranks are visible."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--launch-dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    cmd = json.loads(r.stdout.decode().strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--launch-dry-run" not in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    # no GPU here: asking for 8 ranks over RCCL must fail loudly, never print an n_gpus:1 line
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"n_gpus" not in r.stdout and b"--gpus 8" in r.stderr
    # one rank of a two-rank job that was told --gpus 8: refused as well
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env2, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"WORLD_SIZE=2" in r.stderr


def _bcast_worker(rank, world, port, nbytes, piece):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard.BROADCAST_PIECE = piece
        g = torch.Generator().manual_seed(3)
        want = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, generator=g)
        got = shard.broadcast_arena(want.clone() if rank == 0 else None, torch.device("cpu"), src=0)
        assert got.numel() == nbytes and torch.equal(got, want)
    finally:
        dist.destroy_process_group()


def test_arena_broadcast_in_pieces():
    """The arena goes out as ONE logical broadcast issued in bounded pieces (3 GB - 98 GB arenas): ragged last piece."""
    mp.spawn(_bcast_worker, args=(2, _free_port(), 1_000_003, 65_536), nprocs=2, join=True)
