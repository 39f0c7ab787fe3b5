"""A batch of mixed read lengths (7 reads of 150 letters, then 3 of 250, repeated: average 180) against the 100 Mbp text, call
after call on one index: the first call takes the seed kernel's three-word form by the average and leaves the reads of 250
letters to the index walk; the kernel tells the index, and the calls after it take the four-word form.  Prints one JSON line per
call (time, strands left to the walk) and the digest of the MEMs of the first and the last call (equal).
    python tools/mixed_lengths_bench.py [groups_of_ten_reads]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamem_amd import engine  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dev = torch.device("cuda:0")
ref = engine.synth_reference(100_000_000, 42, dev)
idx = engine.Index.build(ref, dev)
a = engine.synth_reads(ref, 0, 7 * G, 150, 0.02, 42, 50)[: 7 * G * 150].view(G, 7 * 150)
b = engine.synth_reads(ref, 7 * G, 3 * G, 250, 0.02, 43, 50)[: 3 * G * 250].view(G, 3 * 250)
reads = torch.cat([torch.cat([a, b], dim=1).reshape(-1), torch.zeros(64, dtype=torch.uint8, device=dev)])
lens = torch.tensor([150] * 7 + [250] * 3, dtype=torch.int64, device=dev).repeat(G)
offsets = torch.zeros(10 * G + 1, dtype=torch.int64, device=dev)
offsets[1:] = torch.cumsum(lens, 0)
M = 10 * G
m = idx.matcher(M, True, 6 * M + 1024, int(offsets[-1].item()))


def digest():
    boff = m.block_offsets.cpu().numpy()
    mm = m.mems[: int(boff[-1])].cpu().numpy().view(np.uint32).astype(np.int64)
    return [int(mm.shape[0]), int(mm[:, 0].sum()), int(mm[:, 1].sum()), int(mm[:, 2].sum())]


first = None
for call in range(5):
    torch.cuda.synchronize()
    engine.reset_timings()
    t0 = time.perf_counter()
    total = m.run(reads, offsets, 20)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm = engine.timings()
    d = digest()
    first = first or d
    print(json.dumps({"call": call, "reads": M, "mems": total, "ms": el * 1e3, "seed_ms": tm["seed_ms_sum"], "k8_ms": tm["k8_ms_sum"],
                      "digest_equals_first_call": d == first}), flush=True)
st = engine.search_stats(m, reads, offsets, 20)
print(json.dumps({k: v for k, v in st.items() if k.startswith("seed_") or k in ("survivors", "items", "mems")}), flush=True)
