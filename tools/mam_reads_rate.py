#!/usr/bin/env python3
"""-mam on the headline reads (1 M reads of the bench workload, both strands): device time against -mem."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slamem_amd import engine
dev = torch.device("cuda:0")
n, M, L = 100_000_000, int(os.environ.get("READS", 1_000_000)), 150
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
offsets = torch.arange(M + 1, dtype=torch.int64, device=dev) * L
out = {"reads": M}
for mam in (False, True):
    m = engine.Matcher(idx, M, True, 4 * M + 1024, M * L, mam=mam)
    m.run(reads, offsets, 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tot = m.run(reads, offsets, 20)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["mam" if mam else "mem"] = {"ms": round(dt * 1e3, 2), "found": int(tot), "per_s": round(tot / dt)}
    del m
print(json.dumps(out))
