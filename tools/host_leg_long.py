"""python tools/host_leg_long.py [reads=30000000] [batch_reads=1000000]: the host-to-host leg on a LONGER stream of the headline reads (same 100 Mbp
reference), million-read batches, median of 3 passes; SLAMEM_STREAM_CARRY etc. from the environment."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import engine
dev = torch.device("cuda:0")
n, M, L = 100_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000, 150
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
# (the generator's grid covers one thread per letter: at most 2^32 threads per launch, so the reads are made in pieces)
reads = torch.empty(M * L + 16, dtype=torch.uint8, device=dev)
for first in range(0, M, 10_000_000):
    cnt = min(10_000_000, M - first)
    reads[first * L: (first + cnt) * L] = engine.synth_reads(ref, first, cnt, L, 0.02, 42, 50)[: cnt * L]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
r = engine.host_to_host_leg(idx, reads, M, L, 20, True, steps=3, batch_reads=B, slots=6)
print(json.dumps({"reads": M, "batch_reads": B, "host_to_host_ms": round(r["host_to_host_ms"], 2), "MEMs_per_s": round(r["value_host_to_host"] / 1e6, 1),
                  "passes_ms": r["host_to_host"]["passes_ms"], "steady": r["host_to_host"]["steady_state_MEMs_per_s"],
                  "env": {k: v for k, v in os.environ.items() if k.startswith("SLAMEM_")}}))
