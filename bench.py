#!/usr/bin/env python3
"""bench.py -- MEMs/sec of the MI355X MEM engine on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1 without WORLD_SIZE in the environment: bench.py starts N ranks ITSELF (a child `python -m torch.distributed.run
--nproc-per-node N ... bench.py`, spawned before this process touches the GPU) and exits with the child's code; under
torch.distributed.run (the driver's launch line) it is one rank of N.  It refuses to run with fewer ranks or devices
than --gpus asks for.

Workload (config.workload): synthetic 100 Mbp reference (SURVEY.md Appendix C.2 generator, seed 42), 10 M x 150 bp
reads with 2 % substitutions, half of them reverse-complemented, `-b -l 20` (BASELINE.json configs[2], the
configuration the metric is quoted on).  One "step" = one pass of the hot path (slamem_find_mems_device: K8s
seed-and-compare for the reads, the presence filter + K8 index walk for the strands it leaves, K9 output placement) over
the rank's read batch, inputs already resident in HBM.  Reads shard
across ranks with no data-path collective (slamem.c:90-95: records are independent); the index is built once on rank
0 and broadcast over RCCL/xGMI straight from its arena; per-rank MEM counts are gathered every step.
  --scaling weak   (default; the contract's mode for sharded paths) every rank matches its own 10 M reads
  --scaling strong the SAME 10 M reads split into N contiguous ranges (slamem_amd/shard.py::shard_bounds)
With N > 1 the line also carries the other mode's rate, measured in the same process after the headline loop
(`strong_scaling` / `weak_scaling`), so one launch gives both.

Rank 0 prints ONE JSON line.
  roofline      the dominant kernel (K8s k_seed_mems since round 4; K8 k_find_mems_v3 when the batch does not take the seed
                path): `traffic` = bytes the kernel's lanes asked the memory system for, counted by the kernel's
                diagnostic instantiation on the same batch in this run (K8s: 64 B per seed-table line = one per window looked up,
                128 B per compare = four 32-byte units of the text (bit-planes, letter mask, occurs-once plane), the reads'
                own bytes, 12 B per MEM written, 5 B of flags / counts per strand); `achieved` = traffic / the kernel's
                mean duration (HIP events on its stream); `frac` = achieved / 8 TB/s.  `request_rate_frac` = 64-byte
                lines/s over a dependent-random-line ceiling measured in this process on this index arena.  The SURVEY
                8(d) reference-work formula is kept as `reference_work_GBps` (it charges the reference's work, not what
                this engine moves).
  cpu_baseline  the oracle (our CPU restatement of the reference algorithm) on a bounded sample of the same reads on
                this box's host cores: one thread (`cpu_baseline`) and all of this box's share (`cpu_baseline_all_cores`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FIXED_BYTES_PER_BASE = 155.0  # SURVEY.md 8(d) figure for these reads
# the oracle port against the real reference on the same reads, measured in the build container by
# tests/tools/calibrate_port_vs_reference.py (its output is committed: profiles/r03_port_vs_reference.json); the port is the
# faster of the two, i.e. a conservative CPU baseline.  1.22 = round 1's measurement (30.0 k vs 24.5 k MEMs/s, one Xeon core).
def _port_vs_reference_ratio() -> float:
    for name in ("r04_port_vs_reference.json", "r03_port_vs_reference.json"):  # the latest calibration that is there
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return float(json.load(f)["port_vs_reference_ratio"])
        except (OSError, KeyError, ValueError):
            pass
    return 1.22


PORT_VS_REFERENCE_RATIO = _port_vs_reference_ratio()


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    p.add_argument("--ref-len", type=int, default=100_000_000)
    p.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (weak) / in total (strong)")
    p.add_argument("--read-len", type=int, default=150)
    p.add_argument("--min-len", type=int, default=20)
    p.add_argument("--sub", type=float, default=0.02)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--rc-percent", type=int, default=50)
    p.add_argument("--forward-only", action="store_true")
    p.add_argument("--cpu-sample-reads", type=int, default=250_000)
    p.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0 = this box's share)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-stats", action="store_true", help="skip the diagnostic counter launch (profiling passes: keeps "
                                                          "the kernel list to the timed instantiations)")
    p.add_argument("--no-host-leg", action="store_true", help="skip the host-to-host (PCIe-inclusive) measurement")
    p.add_argument("--long-stream-reads", type=int, default=30_000_000,
                   help="also measure the host-to-host leg on a stream of this many reads in 2 M-read batches (0 = skip)")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                                                     "N>1 path on a box with fewer GPUs than ranks)")
    p.add_argument("--launch-dry-run", action="store_true", help="print the child command line and exit (launcher test)")
    return p.parse_args(argv)


# ---- launcher: `python bench.py --gpus N` starts its own ranks -------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(a, argv) -> list:
    """The child command that runs this file as N ranks (nothing here touches the GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def visible_devices() -> int:
    import torch
    return torch.cuda.device_count()  # counts devices without creating a HIP context in this process


def maybe_launch(a, argv) -> None:
    """--gpus N > 1 outside torch.distributed.run: spawn the ranks as a CHILD (never exec after a GPU call) and exit."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    cmd = launcher_command(a, [x for x in argv if x != "--launch-dry-run"])
    if a.launch_dry_run:
        print(json.dumps({"launch": cmd}))
        raise SystemExit(0)
    ndev = visible_devices()
    if a.backend == "nccl" and ndev < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) are visible; refusing to report a smaller job")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


# ---- CPU baseline ---------------------------------------------------------------------------------------------------
def cpu_baseline(ref_host, reads_host, read_len: int, min_len: int, both: bool, threads: int, reads_per_thread: int):
    """Oracle (port of the reference algorithm) on the host cores: bounded sample; one thread, then `threads` threads
    over disjoint read shards (the oracle's matching is re-entrant; ctypes releases the GIL)."""
    import numpy as np
    from oracle import pyoracle as po  # cpu_baseline leg: the only place bench.py touches oracle/
    t0 = time.time()
    idx = po.OracleIndex(ref_host.tobytes())
    build_s = time.time() - t0
    n = reads_host.shape[0] // read_len
    S = min(n, reads_per_thread)
    offsets = np.arange(S + 1, dtype=np.uint64) * np.uint64(read_len)
    counts = po.Counts()
    t0 = time.time()
    mems, bc = idx.match_batch(reads_host[: S * read_len], offsets, min_len, both, counts)
    match_s = time.time() - t0
    out = {"mems": mems, "block_counts": bc, "counts": counts, "build_s": build_s, "match_s": match_s, "reads": S}
    if threads > 1:
        T = min(threads, n // S) if S else 0
        res = [0] * T

        def work(t):
            m, _ = idx.match_batch(reads_host[t * S * read_len: (t + 1) * S * read_len], offsets, min_len, both)
            res[t] = len(m)
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.time()
        for x in th:
            x.start()
        for x in th:
            x.join()
        out["mt"] = {"threads": T, "reads": T * S, "mems": int(sum(res)), "match_s": time.time() - t0}
    return out


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    maybe_launch(a, argv)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; refusing to report a different job size")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPU(s) are visible")
    dev = torch.device("cuda", local_rank if a.backend == "nccl" else local_rank % ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
        assert dist.get_world_size() == a.gpus
    cdev = dev if a.backend == "nccl" else torch.device("cpu")  # where collective buffers live

    from slamem_amd import engine, shard

    both = not a.forward_only
    strands = 2 if both else 1
    n, M, L = a.ref_len, a.reads, a.read_len

    # ---- index: built once on rank 0, broadcast over RCCL/xGMI straight from its arena ---------------------------------
    ref = engine.synth_reference(n, a.seed, dev)  # every rank needs the text to generate its reads
    torch.cuda.synchronize(dev)
    build_s = bcast_s = 0.0
    build_t = {}
    index = None
    if rank == 0:
        t0 = time.time()
        index = engine.Index.build(ref, dev)
        torch.cuda.synchronize(dev)
        build_s = time.time() - t0
        build_t = {k: v for k, v in engine.timings().items() if k.startswith("build_")}
    if world > 1:
        src = index.arena_view() if rank == 0 else None  # zero-copy view of the arena: no second copy of the index
        if src is not None and cdev.type == "cpu":
            src = src.cpu()
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.time()
        arena = shard.broadcast_arena(src, cdev, src=0)  # ONE logical broadcast of the whole index (<= 1 GiB pieces)
        if cdev.type == "cpu":
            arena = arena.to(dev)
        torch.cuda.synchronize(dev)
        dist.barrier()
        bcast_s = time.time() - t0
        if rank != 0:
            index = engine.Index.attach(arena)
    arena_bytes = int(index.info.arena_bytes)

    # ---- the two shardings of the read set ------------------------------------------------------------------------------
    def make_batch(first, count):
        reads = engine.synth_reads(ref, first, count, L, a.sub, a.seed, a.rc_percent)
        offsets = torch.arange(count + 1, dtype=torch.int64, device=dev) * L
        return reads, offsets, count

    def weak_batch():
        return make_batch(rank * M, M)

    def strong_batch():
        bounds = shard.shard_bounds(np.arange(M + 1, dtype=np.uint64) * np.uint64(L), world)
        return make_batch(int(bounds[rank]), int(bounds[rank + 1] - bounds[rank]))

    def timed(batch, steps, warmup):
        """W untimed + K timed steps, barrier + synchronize on both sides, MAX over ranks."""
        reads, offsets, count = batch
        # (room for the MEMs: 2.4 a read on the default batch; more letters a read, more MEMs)
        matcher = index.matcher(count, both, mems_capacity=4 * count * max(1, (L + 149) // 150) + 1024, query_bytes=count * L)
        counts_all = torch.zeros(world, dtype=torch.int64, device=cdev)
        count_bufs = (torch.zeros(1, dtype=torch.int64, device=cdev), torch.zeros(world, dtype=torch.int64, device=cdev))

        def step():
            nonlocal counts_all
            total = matcher.run(reads, offsets, a.min_len)
            # the per-rank MEM counts, gathered at every step (tiny; no data-path collective; buffers made once)
            counts_all = shard.gather_counts(total, cdev, buffers=count_bufs)
        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        engine.reset_timings()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        tm = engine.timings()
        launches = max(1, tm["search_launches"])
        vals = torch.tensor([elapsed, tm["search_kernel_ms_sum"] / launches, tm["k8_ms_sum"] / launches,
                             tm["prefilter_ms_sum"] / launches, tm["seed_ms_sum"] / launches], dtype=torch.float64, device=cdev)
        lo = vals.clone()
        if world > 1:
            dist.all_reduce(vals, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        v, lo = vals.tolist(), lo.tolist()
        reads_total = torch.tensor([count], dtype=torch.int64, device=cdev)
        if world > 1:
            dist.all_reduce(reads_total)
        return {"elapsed": v[0], "kernel_ms": v[1], "k8_ms": v[2], "prefilter_ms": v[3], "seed_ms": v[4], "kernel_ms_min": lo[1],
                "mems": int(counts_all.sum().item()), "reads": int(reads_total.item()), "matcher": matcher,
                "batch": batch}

    main_batch = weak_batch() if a.scaling == "weak" else strong_batch()
    r = timed(main_batch, a.steps, a.warmup)
    other = None
    if world > 1:  # the other sharding, same process, fewer steps
        r["matcher"] = None if rank != 0 else r["matcher"]
        other = timed(strong_batch() if a.scaling == "weak" else weak_batch(), max(2, a.steps // 2), 1)
        other.pop("matcher")
        other.pop("batch")

    if rank == 0:
        reads, offsets, count = r["batch"]
        matcher = r["matcher"]
        per_step = r["elapsed"] / a.steps
        out = {
            "metric": "MEMs/sec (and queries/sec) on 100 Mbp ref x 10 M 150 bp queries, l=20",
            "value": r["mems"] / per_step,
            "unit": "MEMs/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": per_step * 1e3,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"synthetic {n} bp reference (splitmix64 seed {a.seed}) x {M} reads"
                                   f"{'/GPU' if a.scaling == 'weak' else ' in total'} of {L} bp, "
                                   f"{a.sub:.0%} substitutions, {a.rc_percent}% reverse-complemented, "
                                   f"{'-b ' if both else ''}-l {a.min_len}",
                       "ref_len": n, "reads_total": r["reads"], "read_len": L, "min_len": a.min_len, "both_strands": both,
                       "parallelism": f"query shards x{world}, " + ("one GPU, no replication" if world == 1 else
                                       f"index replicated by {'RCCL' if a.backend == 'nccl' else a.backend + ' (host-staged)'} broadcast")},
            "queries_per_sec": r["reads"] / per_step,
            "mems_per_step": r["mems"],
            "index_build_s": build_s,
            "index_build_ms": build_t,
            "index_bytes": arena_bytes,
            "index_broadcast_s": bcast_s,
            "kernel": "K8s k_seed_mems (+ presence filter and K8 k_find_mems_v3 for the strands it leaves; K8a k_prefilter + K8 "
                      "for batches that do not take the seed path); HIP events on their stream",
            "kernel_ms": r["kernel_ms"],
            "kernel_ms_min_over_ranks": r["kernel_ms_min"],
            "k8s_ms": r["seed_ms"],
            "k8_ms": r["k8_ms"],
            "k8a_ms": r["prefilter_ms"],
        }
        if other is not None:
            ps = other["elapsed"] / max(2, a.steps // 2)
            out["strong_scaling" if a.scaling == "weak" else "weak_scaling"] = {
                "value": other["mems"] / ps, "ms_per_step": ps * 1e3, "reads_total": other["reads"],
                "kernel_ms_max": other["kernel_ms"], "kernel_ms_min": other["kernel_ms_min"]}

        # ---- roofline of the dominant kernel, from counters of THIS run ---------------------------------------------
        st = engine.search_stats(matcher, reads, offsets, a.min_len) if not a.no_stats else None
        if st is None:
            st = {k: 0 for k in ("fm_lines_top", "fm_lines_bottom", "rec_lines_fail", "rec_lines_pend", "rec_lines_flush",
                                 "query_loads", "prefilter_probes", "prefilter_query_loads", "lane_trips", "wave_trips",
                                 "dir_sa_lines", "dir_group_loads", "dir_rec_lines")}
        k8_lines = (st["fm_lines_top"] + st["fm_lines_bottom"] + st["rec_lines_fail"] + st["rec_lines_pend"]
                    + st["rec_lines_flush"] + st.get("dir_sa_lines", 0) + st.get("dir_group_loads", 0)
                    + st.get("dir_rec_lines", 0) + st.get("jump_lines", 0) + st.get("skip_group_loads", 0)
                    + st.get("skip_probe_lines", 0) + st.get("skip_attempts", 0))
        # 64 B per random line; the packed query windows (16 B, 32 letters) and the 16 B of query beside a text group are
        # sequential within a strand's 80 bytes
        k8_bytes = 64 * k8_lines + 16 * (st["query_loads"] + st.get("dir_group_loads", 0) + st.get("skip_group_loads", 0))
        k8a_bytes = 64 * st["prefilter_probes"] + 16 * st["prefilter_query_loads"]
        k8_s = max(1e-9, r["k8_ms"] * 1e-3)
        ceiling = engine.random_line_ceiling(index) if hasattr(engine, "random_line_ceiling") else None
        bases = float(count) * L * strands
        walk = {"kernel": "k_find_mems_v3", "traffic": k8_bytes, "kernel_ms": r["k8_ms"], "lines_64B": k8_lines,
                "achieved": k8_bytes / k8_s / 1e9, "frac": k8_bytes / k8_s / 1e9 / HBM_PEAK_GBS, "lines_per_s": k8_lines / k8_s,
                "lane_use": st["lane_trips"] / max(1, 64 * st["wave_trips"]), "strands": st.get("survivors", 0)}
        pf = {"kernel": "k_prefilter", "traffic": k8a_bytes, "kernel_ms": r["prefilter_ms"],
              "achieved": k8a_bytes / max(1e-9, r["prefilter_ms"] * 1e-3) / 1e9,
              "frac": k8a_bytes / max(1e-9, r["prefilter_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if r["seed_ms"] > 0 and st.get("seed_reads", 0):
            # K8s: one seed-table line per window, four text units + two unit-mask words per compare (+ four letter-mask words for
            # the few whose units hold a letter that is not A,C,G,T), the reads' bytes once, 12 B per MEM and 5 B per strand out
            s_s = r["seed_ms"] * 1e-3
            s_bytes = (64 * st["seed_windows"] + 128 * st["seed_compares"] + st["seed_query_bytes"] + 12 * st["seed_mems"] + 5 * st["items"])
            s_lines = st["seed_windows"] + 2 * st["seed_compares"] + st["seed_query_bytes"] // 64
            achieved = s_bytes / s_s / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": "k_seed_mems", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": s_bytes,
                "traffic_source": "counters of the kernel's diagnostic instantiation on this batch (this run): 64 B per window looked "
                                  "up (seed-table line), 128 B per compare (four 32-byte text units: planes, letter mask, "
                                  "occurs-once plane), the reads' bytes, 12 B per MEM, 5 B per strand; cross-check against "
                                  "rocprofv3 FETCH_SIZE + WRITE_SIZE in profiles/",
                "bound_note": "the kernel is bound by vector-instruction issue and by the fabric's line-request rate at once "
                              "(profiles/: SQ_INSTS_VALU x 4 cycles / 1,024 SIMDs and TCC_EA0_RDREQ / time); frac is bytes over the "
                              "HBM peak as the contract defines it",
                "kernel_ms": r["seed_ms"], "lines_64B": s_lines, "lines_per_s": s_lines / s_s,
                "random_line_ceiling_per_s": ceiling, "request_rate_frac": (s_lines / s_s / ceiling) if ceiling else None,
                "request_rate_note": "lines the lanes ask for (two per compare: many of those are served by L2 / Infinity Cache -- "
                                     "the text units are 50 MB) over the probed ceiling of dependent random lines from HBM: at "
                                     "or above 1 the kernel is at that ceiling; the fabric's own count is TCC_EA0_RDREQ in profiles/",
                "lines_per_query_base": s_lines / bases, "windows_per_read": st["seed_windows"] / max(1, st["seed_reads"]),
                "compares_per_read": st["seed_compares"] / max(1, st["seed_reads"]),
                "strands_left_to_index_walk": st["seed_strands_left"], "index_walk": walk, "counters": st,
            }
        else:
            achieved = k8_bytes / k8_s / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": "k_find_mems_v3", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if not a.no_stats else None, "traffic": k8_bytes if not a.no_stats else None,
                "traffic_source": "load counters of the kernel's diagnostic instantiation on this batch (this run); "
                                  "cross-check against rocprofv3 FETCH_SIZE in profiles/",
                "kernel_ms": r["k8_ms"], "lines_64B": k8_lines, "lines_per_s": k8_lines / k8_s,
                "random_line_ceiling_per_s": ceiling, "request_rate_frac": (k8_lines / k8_s / ceiling) if ceiling else None,
                "lane_use": st["lane_trips"] / max(1, 64 * st["wave_trips"]),
                "lines_per_query_base": k8_lines / bases,
                "prefilter": pf,
                "counters": st,
            }

        # ---- CPU baseline + parity of the GPU result on the sample (the checker, not the thing measured) ----------------
        bytes_per_base = FIXED_BYTES_PER_BASE
        cpu = cpu_all = None
        if world == 1 and not a.no_cpu_baseline and a.cpu_sample_reads > 0:
            S = min(a.cpu_sample_reads, count)
            threads = a.cpu_threads or min(16, os.cpu_count() or 1)
            threads = max(1, min(threads, count // S))
            ref_h = ref.cpu().numpy()
            reads_h = reads[: threads * S * L].cpu().numpy()
            c = cpu_baseline(ref_h, reads_h, L, a.min_len, both, threads, S)
            cpu = {"value": len(c["mems"]) / c["match_s"], "unit": "MEMs/s", "cores": 1, "kind": "port",
                   "sample": f"first {S} reads of the same batch ({'both strands' if both else 'forward'}), "
                             f"oracle/liboracle.so single thread, matching only",
                   "queries_per_sec": S / c["match_s"], "match_s": c["match_s"], "index_build_s": c["build_s"],
                   "host_cpus": os.cpu_count(), "port_vs_reference_ratio": PORT_VS_REFERENCE_RATIO,
                   "port_vs_reference_note": "oracle port / real reference MEMs/s on the same reads, one core of the "
                                             "build container (the reference cannot travel to the GPU box)"}
            if "mt" in c:
                mt = c["mt"]
                cpu_all = {"value": mt["mems"] / mt["match_s"], "unit": "MEMs/s", "cores": mt["threads"], "kind": "port",
                           "sample": f"first {mt['reads']} reads in {mt['threads']} shards, one oracle thread per shard, "
                                     f"matching only", "queries_per_sec": mt["reads"] / mt["match_s"],
                           "match_s": mt["match_s"]}
            bytes_per_base = c["counts"].algorithmic_bytes() / max(1, c["counts"].n_querybase)
            nb = S * strands
            boff = matcher.block_offsets[: nb + 1].cpu().numpy()
            gm = matcher.mems[: int(boff[-1])].cpu().numpy().view(np.uint32)
            om = c["mems"]
            ok = (np.array_equal(np.diff(boff), c["block_counts"].astype(np.int64)) and len(om) == gm.shape[0]
                  and np.array_equal(gm[:, 0], om["ref_pos"]) and np.array_equal(gm[:, 1], om["query_pos"])
                  and np.array_equal(gm[:, 2], om["length"]))
            out["sample_parity_vs_oracle"] = bool(ok)
            out["op_counts_per_base"] = {k: v / max(1, c["counts"].n_querybase) for k, v in c["counts"].as_dict().items()}
        out["roofline"]["reference_work_GBps"] = bytes_per_base * bases / (r["kernel_ms"] * 1e-3) / 1e9
        out["roofline"]["reference_work_note"] = ("SURVEY 8(d) formula: op counts of the restated reference algorithm x "
                                                  "reference-layout bytes / (K8a+K8 time); the engine skips part of "
                                                  "that work, so this is NOT a bandwidth")
        out["cpu_baseline"] = cpu
        out["cpu_baseline_all_cores"] = cpu_all

        # ---- SURVEY 8(d)'s metric as defined: reads in host memory -> MEM triples in host memory ------------------------
        if world == 1 and not a.no_host_leg and hasattr(engine, "host_to_host_leg"):
            out.update(engine.host_to_host_leg(index, reads, count, L, a.min_len, both, steps=max(3, a.steps // 2) | 1))
            out["value_device_resident"] = out["value"]
            out["host_to_host_frac_of_device_resident"] = out["value_host_to_host"] / out["value"]
            # the link's share: the reads' letters and offsets one way at the 57 GB/s measured on these boxes
            link_ms = (count * L + 8 * count) / 57e9 * 1e3
            out["host_to_host_link_floor_ms"] = link_ms
            out["host_to_host_frac_of_link_floor"] = link_ms / out["host_to_host_ms"]
            # the same leg for a caller that holds its reads as bit-planes (slamem_stream_submit_packed: 48 B per 150 letters)
            pk = engine.host_to_host_leg(index, reads, count, L, a.min_len, both, steps=max(3, a.steps // 2) | 1, packed=True)
            out["host_to_host_packed"] = {"value": pk["value_host_to_host"], "ms": pk["host_to_host_ms"], "mems": pk["host_to_host_mems"],
                                          "frac_of_device_resident": pk["value_host_to_host"] / out["value"], **pk["host_to_host"]}
            if a.long_stream_reads > count:
                # the same leg on a longer stream of the same reads' generator: ramp and end amortised, batches large enough for
                # K8's per-launch cost; the reads are made in pieces (one generator thread per letter)
                ML = a.long_stream_reads
                long_reads = torch.empty(ML * L + 16, dtype=torch.uint8, device=dev)
                for first in range(0, ML, 10_000_000):
                    cnt = min(10_000_000, ML - first)
                    long_reads[first * L: (first + cnt) * L] = engine.synth_reads(ref, first, cnt, L, a.sub, a.seed, a.rc_percent)[: cnt * L]
                lr = engine.host_to_host_leg(index, long_reads, ML, L, a.min_len, both, steps=3, batch_reads=2_000_000)
                del long_reads
                out["host_to_host_long_stream"] = {
                    "reads": ML, "batch_reads": 2_000_000, "value": lr["value_host_to_host"], "ms": lr["host_to_host_ms"],
                    "mems": lr["host_to_host_mems"], "passes_ms": lr["host_to_host"]["passes_ms"],
                    "frac_of_device_resident": lr["value_host_to_host"] / out["value"],
                    "note": "same pipeline, three times the headline's reads, median of 3 passes"}
            out["value_note"] = ("`value` is the device-resident rate (inputs in HBM when the clock starts: the bench contract); "
                                 "`value_host_to_host` is SURVEY 8(d)'s metric as defined -- reads in host memory -> MEM triples in "
                                 "host memory through slamem_stream_*, median of the passes")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
