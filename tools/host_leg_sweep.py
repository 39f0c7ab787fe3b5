#!/usr/bin/env python3
"""Host-to-host (PCIe-inclusive) rate of slamem_stream_* on the bench workload for several batch sizes / slot counts,
beside the raw pinned-copy bandwidth of the box.  One JSON line per setting."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slamem_amd import engine

dev = torch.device("cuda:0")
n, M, L = 100_000_000, int(os.environ.get("READS", 10_000_000)), 150
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
# raw copies
h = torch.empty(1_500_000_000, dtype=torch.uint8).pin_memory()
d = torch.empty_like(h, device=dev)
for name, a, b in (("h2d", d, h), ("d2h", h, d)):
    a.copy_(b, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); a.copy_(b, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"copy": name, "GBps": 1.5 / dt}), flush=True)
del h, d
offsets = torch.arange(M + 1, dtype=torch.int64, device=dev) * L
m = idx.matcher(M, True, 4 * M + 1024, M * L)
m.run(reads, offsets, 20); engine.reset_timings()
t0 = time.perf_counter(); tot = m.run(reads, offsets, 20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"device_resident_ms": dt * 1e3, "mems": tot, "kernel_ms": engine.timings()["search_kernel_ms"]}), flush=True)
del m
K = 1000
for name, sched in (("current", None),
                    ("D", [250 * K, 500 * K, 1000 * K, 1500 * K, 2000 * K, 2000 * K, 1750 * K, 1000 * K]),
                    ("S1", [100 * K, 200 * K, 400 * K, 800 * K, 1500 * K, 2000 * K, 2000 * K, 1500 * K, 1000 * K, 500 * K]),
                    ("S2", [125 * K, 250 * K, 500 * K, 1000 * K, 1500 * K, 1500 * K, 1500 * K, 1500 * K, 1000 * K, 750 * K, 375 * K]),
                    ("S3", [250 * K, 500 * K, 1000 * K, 1500 * K, 1500 * K, 1500 * K, 1500 * K, 1250 * K, 750 * K, 250 * K]),
                    ("S4", [100 * K, 300 * K, 700 * K, 1400 * K, 2000 * K, 2000 * K, 2000 * K, 1000 * K, 500 * K])):
    r = engine.host_to_host_leg(idx, reads, M, L, 20, True, steps=2, batch_reads=1_000_000, slots=6, schedule=sched)
    print(json.dumps({"schedule": name, "ms": r["host_to_host_ms"], "MEMs_per_s": r["value_host_to_host"], "batches": r["host_to_host"]["batches"],
                      "kernel_ms_sum": r["host_to_host"]["kernel_ms_sum"]}), flush=True)
