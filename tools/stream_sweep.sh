#!/bin/bash
# host-to-host leg against the number of search threads of the stream and the width of a K8 launch
for cfg in "2 4096" "3 2048" "4 1024" "4 2048" "3 1536" "2 2048"; do set -- $cfg
  SLAMEM_STREAM_SEARCH=$1 SLAMEM_K8_WAVES=$2 python - <<PY
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from slamem_amd import engine
dev = torch.device("cuda:0")
n, M, L = 100_000_000, 10_000_000, 150
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
for batch, slots in ((1_000_000, 8), (1_500_000, 8)):
    r = engine.host_to_host_leg(idx, reads, M, L, 20, True, steps=2, batch_reads=batch, slots=slots)
    print(json.dumps({"search_threads": $1, "k8_waves": $2, "batch_reads": batch, "slots": slots, "ms": round(r["host_to_host_ms"], 2)}), flush=True)
PY
done
