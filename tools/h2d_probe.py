"""Pinned host -> device copy rate of this box for the batch sizes of the host-to-host leg (what slamem_stream's upload
stage can reach at best), and device -> pinned host beside it."""
import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
dev = torch.device("cuda:0")
out = {}
for mb in (37, 75, 150, 600, 1500):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d.copy_(h, non_blocking=True); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out["h2d_%dMB_GBps" % mb] = round((mb << 20) / best / 1e9, 1)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.copy_(d, non_blocking=True); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out["d2h_%dMB_GBps" % mb] = round((mb << 20) / best / 1e9, 1)
    del h, d
# both directions at once (two streams)
h1 = torch.empty(600 << 20, dtype=torch.uint8).pin_memory(); d1 = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
h2 = torch.empty(200 << 20, dtype=torch.uint8).pin_memory(); d2 = torch.empty(200 << 20, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["duplex_600MB_up_200MB_down_ms"] = round(dt * 1e3, 2)
print(json.dumps(out))
