"""One traced pass of the host-to-host leg (SLAMEM_STREAM_TRACE=1 prints every stage of every batch): python tools/host_leg_trace.py [packed 0/1] [batch reads]"""
import json, os, sys
os.environ["SLAMEM_STREAM_TRACE"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import engine
packed = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
br = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
M, n, L, dev = 10_000_000, 100_000_000, 150, torch.device("cuda:0")
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
r = engine.host_to_host_leg(idx, reads, M, L, 20, True, steps=1, batch_reads=br, packed=packed)
print(json.dumps({"packed": packed, "ms": r["host_to_host_ms"]}))
