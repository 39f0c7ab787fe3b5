import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import engine
n = 3_100_000_000
ref = engine.synth_reference(n, 42, "cuda:0")
engine.synth_plant_repeats(ref, 42)
idx = engine.Index.build(ref, "cuda:0")
print(engine.timings()["build_total_ms"])
