import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import engine
dev = torch.device("cuda:0")
n = 3_100_000_000
ref = engine.synth_reference(n, 42, dev)
engine.synth_plant_repeats(ref, 42)
torch.cuda.synchronize()
t0 = time.time()
idx = engine.Index.build(ref, dev)
torch.cuda.synchronize()
print("build wall", time.time() - t0, engine.timings())
