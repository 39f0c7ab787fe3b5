#!/usr/bin/env python3
"""Soak of the sliced scans (records longer than 4096 letters): random texts with repeats and runs of N, queries pieced
together from text segments (exact, mutated, reverse-complemented, repeated), -mem and -mam, against the oracle in order.
    python tests/tools/stress_slices.py [cases] [first seed]      -> "cases N bad 0"
SLAMEM_SLICE_WARMUP / SLAMEM_MAM_WARMUP vary the warm-up (read once per process)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import rand_text, pack
from oracle import pyoracle as po
from slamem_amd import engine

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
LET = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b
bad = 0
for seed in range(first, first + ncases):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(30_000, 200_000))
    alpha = "ACGT" if seed % 3 else "ACGTN"
    t = np.frombuffer(rand_text(rng, n, alpha, int(rng.integers(0, 60)), max_rep=int(rng.integers(200, 6000)),
                                nrun=int(rng.integers(0, 3000)) if alpha == "ACGTN" else 0), dtype=np.uint8).copy()
    if seed % 4 == 0:  # a long duplicate
        L = int(rng.integers(5_000, min(20_000, n // 4))); a = int(rng.integers(0, n - 3 * L)); b = int(rng.integers(a + L, n - L))
        t[b:b + L] = t[a:a + L]
    qs = []
    for _ in range(int(rng.integers(1, 5))):
        parts = []
        for _ in range(int(rng.integers(1, 6))):
            L = int(rng.integers(500, 30_000)); a = int(rng.integers(0, max(1, n - L)))
            seg = t[a:a + L].copy()
            kind = int(rng.integers(0, 5))
            if kind == 1:
                m = rng.random(seg.shape[0]) < rng.choice([0.0005, 0.005, 0.03]); seg[m] = rng.choice(LET, size=int(m.sum()))
            elif kind == 2:
                seg = COMP[seg[::-1]]
            elif kind == 3:
                seg = rng.choice(LET, size=L)
            parts.append(seg)
        qs.append(np.concatenate(parts).tobytes())
    qs += [t[:int(rng.integers(1, 300))].tobytes(), b""]
    q, off = pack(qs)
    l = int(rng.choice([8, 12, 20, 33, 50])); both = bool(seed & 1)
    o = po.OracleIndex(t.tobytes())
    g = engine.Index.build(t.tobytes(), "cuda:0")
    for mam in (False, True):
        om, obc = o.match_batch(q, off, l, both, mam=mam)
        gm, goff = g.find_mems(q, off, l, both, mam=mam)
        ok = np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)) and len(gm) == len(om) and all(
            np.array_equal(gm[f], om[f]) for f in ("ref_pos", "query_pos", "length"))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "mam", mam, "l", l, "both", both, "n", n, len(gm), len(om), flush=True)
    g.close()
print("cases", ncases, "bad", bad)
