"""Host-to-host leg (slamem_stream_*) on the headline batch: letters and packed reads.  python tools/host_leg_quick.py [reads]"""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import engine
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n, L, dev = 100_000_000, 150, torch.device("cuda:0")
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
for packed in (False, True):
    for br in (1_000_000, 2_000_000):
        r = engine.host_to_host_leg(idx, reads, M, L, 20, True, steps=3, batch_reads=br, packed=packed)
        print(json.dumps({"packed": packed, "batch_reads": br, "MEMs_per_s": r["value_host_to_host"], "ms": r["host_to_host_ms"], "mems": r["host_to_host_mems"],
                          "passes_ms": r["host_to_host"]["passes_ms"], "h2d_bytes": r["host_to_host"]["h2d_bytes"]}), flush=True)
