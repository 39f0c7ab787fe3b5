#!/usr/bin/env python3
"""400 more random cases of tests/test_gpu_stress.py (other seeds) against the oracle, in emission order: a longer soak
of the search path than the test suite affords.  Run on the GPU box:  python tests/tools/stress_more.py [first seed] [cases] [mam]  -> "cases 400 bad 0"."""
import os
import sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_stress as T
from oracle import pyoracle as po
from slamem_amd import engine
bad = 0
BASE = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
COUNT = int(sys.argv[2]) if len(sys.argv) > 2 else 400
MAM = len(sys.argv) > 3 and sys.argv[3] == "mam"
for seed in range(BASE, BASE + COUNT):
    rng = np.random.default_rng(seed)
    text, qs, l, both = T.random_case(rng)
    q = np.concatenate(qs) if qs else np.zeros(0, dtype=np.uint8)
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in qs])
    o = po.OracleIndex(text)
    om, obc = o.match_batch(q, off, l, both, mam=MAM)
    g = engine.Index.build(text)
    gm, goff = g.find_mems(q, off, l, both, mam=MAM)
    ok = np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)) and all(np.array_equal(gm[f], om[f]) for f in ("ref_pos", "query_pos", "length"))
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "l", l, "both", both, "n", len(text), len(gm), len(om))
    g.close()
print("cases", COUNT, "bad", bad)
