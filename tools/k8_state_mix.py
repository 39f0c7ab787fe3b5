"""python tools/k8_state_mix.py: where K8's loop trips go, by state of the lane's state machine, on the headline batch (diagnostic
instantiation): lane trips per state, and the wave trips in which at least one lane was in the state -- what a wave pays for
(every trip executes the code of the union of its lanes' states).  One JSON object."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import engine
dev = "cuda:0"
n, M, L = 100_000_000, 10_000_000, 150
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
offsets = torch.arange(M + 1, dtype=torch.int64, device=dev) * L
m = idx.matcher(M, True, 4 * M + 1024, M * L)
m.run(reads, offsets, 20)
st = engine.search_stats(m, reads, offsets, 20)
names = ["EXT", "REC", "FLUSH", "DSA", "DIR", "DEND", "JQ", "JT", "SKV", "SKQ", "SKP"]
lt, wt = st["state_lane_trips"], st["state_wave_trips"]
print(json.dumps({"lane_trips": st["lane_trips"], "wave_trips": st["wave_trips"], "lane_use": st["lane_trips"] / (64 * st["wave_trips"]),
                  "per_state": {names[i]: {"lane_trips": lt[i], "share_of_lane_trips": round(lt[i] / st["lane_trips"], 4),
                                           "wave_trips_with_state": wt[i], "share_of_wave_trips": round(wt[i] / st["wave_trips"], 4),
                                           "lanes_in_state_when_present": round(lt[i] / max(1, wt[i]), 1)} for i in range(11) if lt[i]},
                  "positions": st["positions"], "survivors": st["survivors"]}, indent=1))
