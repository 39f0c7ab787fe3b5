"""One search of the genome-like 248 Mbp text (tools/repeat_load.py's model): python tools/genome_like_run.py [reads] [min_len] [repeats]"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import capi, engine, synth
n = 248_000_000
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L, dev = 150, "cuda:0"
ref = engine.synth_reference(n, 42, dev)
engine.synth_plant_repeats(ref, 42)
engine.synth_plant_genome_like(ref, 42)
c, s, sa, na, nl = synth.genome_like_layout(n)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, R, L, 0.02, 42, 50, avoid=(na, nl))
offsets = torch.arange(R + 1, dtype=torch.int64, device=dev) * L
cap = 8 * R
while True:
    m = idx.matcher(R, True, cap, R * L)
    try:
        total = m.run(reads, offsets, minlen)
        break
    except capi.SlamemError as e:
        if e.code != capi.SLAMEM_ERR_CAPACITY:
            raise
        cap = int(m.last_total * 1.05) + 1024
        del m
engine.reset_timings()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m.run(reads, offsets, minlen)
torch.cuda.synchronize()
tm = engine.timings()
print(json.dumps({"reads": R, "min_len": minlen, "mems": total, "step_ms": (time.perf_counter() - t0) / reps * 1e3, "k8s_ms": tm["seed_ms_sum"] / reps, "k8_ms": tm["k8_ms_sum"] / reps}))
