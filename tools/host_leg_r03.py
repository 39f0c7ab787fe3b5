"""python tools/host_leg_r03.py [tag] [schedule in thousands of reads ...]: SURVEY 8(d)'s host-to-host leg on the headline workload, one JSON line
(median of 5 passes); the environment (SLAMEM_K8_WAVES, GPU_MAX_HW_QUEUES, ...) is whatever the caller set."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import engine
dev = torch.device("cuda:0")
n, M, L = 100_000_000, 10_000_000, 150
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
offsets = torch.arange(M + 1, dtype=torch.int64, device=dev) * L
m = idx.matcher(M, True, 4 * M + 1024, M * L)
for _ in range(2):
    m.run(reads, offsets, 20)
engine.reset_timings()
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    total = m.run(reads, offsets, 20)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t0) / 5 * 1e3
del m
tag = sys.argv[1] if len(sys.argv) > 1 else ""
sched = [int(a) * 1000 for a in sys.argv[2:]] or None
r = engine.host_to_host_leg(idx, reads, M, L, 20, True, steps=5, batch_reads=1_000_000, slots=6, schedule=sched)
print(json.dumps({"tag": tag, "schedule": sched, "device_resident_ms": round(dev_ms, 3), "host_to_host_ms": round(r["host_to_host_ms"], 3),
                  "frac": round(dev_ms / r["host_to_host_ms"], 4), "passes_ms": r["host_to_host"]["passes_ms"],
                  "steady": r["host_to_host"]["steady_state_MEMs_per_s"], "mems": r["host_to_host_mems"], "device_mems": total,
                  "env": {k: v for k, v in os.environ.items() if k.startswith("SLAMEM_") or k.startswith("GPU_MAX")}}))
