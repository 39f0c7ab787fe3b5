#!/usr/bin/env python3
"""Timeline of one K8 launch (diagnostic instantiation): time until the work list is empty, tail after that, wave occupancy."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slamem_amd import engine
dev = torch.device("cuda:0")
n, L = 100_000_000, 150
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
for M in (500_000, 1_000_000, 2_000_000, 10_000_000):
    reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
    offsets = torch.arange(M + 1, dtype=torch.int64, device=dev) * L
    m = idx.matcher(M, True, 4 * M + 1024, M * L)
    m.run(reads, offsets, 20)
    engine.reset_timings(); m.run(reads, offsets, 20); k8 = engine.timings()["k8_ms"]
    st = engine.search_stats(m, reads, offsets, 20)
    k8s = engine.timings()["k8_ms"]
    print(json.dumps({"reads": M, "k8_ms": round(k8, 3), "k8_ms_stats_kernel": round(k8s, 3), "us_until_list_empty": round(st["k8_us_until_list_empty"], 1),
                      "us_tail": round(st["k8_us_tail"], 1), "mean_wave_us": round(st["k8_wave_us_sum"] / max(1, min(4096, (st["survivors"] + 63) // 64)), 1),
                      "lane_use": round(st["lane_trips"] / max(1, 64 * st["wave_trips"]), 3), "trips_per_item": round(st["lane_trips"] / st["survivors"], 1)}))
    del m
