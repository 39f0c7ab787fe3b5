"""Genome-like text, reads at -l 20: K8 with its enumeration jobs in the waves (SLAMEM_ENUM_DEFER=0) against the queue (default):
which strands differ, and how."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamem_amd import capi, engine, synth
n = 248_000_000
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L, dev = 150, "cuda:0"
ref = engine.synth_reference(n, 42, dev)
engine.synth_plant_repeats(ref, 42)
engine.synth_plant_genome_like(ref, 42)
c, s, sa, na, nl = synth.genome_like_layout(n)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, R, L, 0.02, 42, 50, avoid=(na, nl))
offsets = torch.arange(R + 1, dtype=torch.int64, device=dev) * L
res = {}
for mode in ("0", "1"):
    os.environ["SLAMEM_ENUM_DEFER"] = mode
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
    res[mode] = (total, m.block_offsets.cpu().numpy().copy(), m.mems[:total].cpu().numpy().copy())
    print("mode", mode, "total", total, flush=True)
    del m
t0, b0, m0 = res["0"]; t1, b1, m1 = res["1"]
c0, c1 = np.diff(b0), np.diff(b1)
bad = np.nonzero(c0 != c1)[0]
print("strands with different counts:", len(bad), bad[:10], c0[bad[:10]], c1[bad[:10]])
if len(bad) == 0:
    d = np.nonzero((m0 != m1).any(axis=1))[0]
    print("rows that differ:", len(d), d[:5])
    if len(d):
        i = d[0]; g = np.searchsorted(b0, i, side="right") - 1
        print("first in strand", g, "at ordinal", i - b0[g], m0[i], m1[i])
else:
    g = int(bad[0])
    a = m0[int(b0[g]):int(b0[g + 1])]; b = m1[int(b1[g]):int(b1[g + 1])]
    sa_ = set(map(tuple, a.tolist())); sb_ = set(map(tuple, b.tolist()))
    miss = sorted(sa_ - sb_, key=lambda x: (-x[1], -x[2]))
    print("strand", g, "in-wave", len(a), "queue", len(b), "missing", len(miss), miss[:8], "extra", sorted(sb_ - sa_)[:5])
