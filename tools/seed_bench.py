"""Headline batch through the search, timings and counters of K8s / K8 (no CPU legs): python tools/seed_bench.py [reads] [ref_len] [min_len]"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamem_amd import engine  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 20
L = int(sys.argv[4]) if len(sys.argv) > 4 else 150
dev = torch.device("cuda:0")
ref = engine.synth_reference(n, 42, dev)
t0 = time.time()
idx = engine.Index.build(ref, dev)
torch.cuda.synchronize()
print(json.dumps({"build_s": time.time() - t0, "arena_bytes": int(idx.info.arena_bytes),
                  **{k: v for k, v in engine.timings().items() if k.startswith("build_")}}), flush=True)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
offsets = torch.arange(M + 1, dtype=torch.int64, device=dev) * L
m = idx.matcher(M, True, 4 * M + 1024, M * L)
for rep in range(2):
    m.run(reads, offsets, minlen)
torch.cuda.synchronize()
engine.reset_timings()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    total = m.run(reads, offsets, minlen)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / K
tm = engine.timings()
ln = max(1, tm["search_launches"])
print(json.dumps({"mems": total, "ms_per_step": el * 1e3, "MEMs_per_s": total / el, "kernel_ms": tm["search_kernel_ms_sum"] / ln,
                  "seed_ms": tm["seed_ms_sum"] / ln, "k8_ms": tm["k8_ms_sum"] / ln, "k8a_ms": tm["prefilter_ms_sum"] / ln,
                  "search_total_ms": tm["search_total_ms"]}), flush=True)
st = engine.search_stats(m, reads, offsets, minlen)
print(json.dumps({k: v for k, v in st.items() if (k.startswith("seed_")) or k in ("survivors", "items", "mems", "overflow_records")}), flush=True)
# digest of the MEM set as tests/golden/known_answers.json records it: count, sum of lengths
boff = m.block_offsets.cpu().numpy()
mm = m.mems[: int(boff[-1])].cpu().numpy().view(np.uint32)
print(json.dumps({"count": int(mm.shape[0]), "sum_len": int(mm[:, 2].astype(np.int64).sum()), "max_len": int(mm[:, 2].max())}), flush=True)
