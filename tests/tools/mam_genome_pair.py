#!/usr/bin/env python3
"""Cost of -mam on long records: the 4.6 Mbp genome pair of configs[0] through the API, wall time of the search, checked
against the oracle.  (SLAMEM_MAM_WHOLE=1: one lane per whole strand, the round-1 form; default: verified slices.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from golden_cases import ecoli_like_pair
from slamem_amd import engine
from oracle import pyoracle as po
ref, qry = ecoli_like_pair()
off = np.array([0, len(qry)], dtype=np.uint64)
idx = engine.Index.build(ref, "cuda:0")
out = {}
for mam in (False, True):
    idx.find_mems(qry[:1000].copy(), np.array([0, 1000], dtype=np.uint64), 20, True, mam=mam)
    t0 = time.time()
    m, boff = idx.find_mems(qry, off, 20, True, mam=mam)
    out["mam_s" if mam else "mem_s"] = round(time.time() - t0, 3)
    out["mams" if mam else "mems"] = int(len(m))
    if mam:
        t0 = time.time()
        o = po.OracleIndex(ref.tobytes())
        om, obc = o.match_batch(qry, off, 20, True, mam=True)
        out["oracle_build_plus_mam_s"] = round(time.time() - t0, 2)
        out["mam_equal_to_oracle_in_order"] = bool(len(om) == len(m) and all(np.array_equal(m[f], om[f]) for f in ("ref_pos", "query_pos", "length")))
print(json.dumps(out))
