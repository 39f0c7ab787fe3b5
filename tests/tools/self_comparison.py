#!/usr/bin/env python3
"""Worst case of the sliced scans: a genome against ITSELF (one exact match as long as the record, so every slice's
warm-up grows to the record's end).  Wall time of -mem and -mam, checked against the oracle."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from golden_cases import ecoli_like_pair
from slamem_amd import engine
from oracle import pyoracle as po
ref, _ = ecoli_like_pair()
n = int(os.environ.get("SELF_BP", len(ref)))
ref = ref[:n].copy()
off = np.array([0, len(ref)], dtype=np.uint64)
idx = engine.Index.build(ref, "cuda:0")
o = po.OracleIndex(ref.tobytes())
out = {"bp": n}
for mam in (False, True):
    t0 = time.time()
    m, boff = idx.find_mems(ref, off, 20, False, mam=mam)
    out["mam_s" if mam else "mem_s"] = round(time.time() - t0, 3)
    tm = engine.timings()
    out["mam_device_ms" if mam else "mem_device_ms"] = {k: round(tm[k], 2) for k in ("search_kernel_ms", "prefilter_ms", "k8_ms", "search_total_ms") if k in tm}
    t0 = time.time()
    om, obc = o.match_batch(ref, off, 20, False, mam=mam)
    out["oracle_mam_s" if mam else "oracle_mem_s"] = round(time.time() - t0, 2)
    out["mam_equal" if mam else "mem_equal"] = bool(len(om) == len(m) and all(np.array_equal(m[f], om[f]) for f in ("ref_pos", "query_pos", "length")))
    out["mams" if mam else "mems"] = int(len(m))
print(json.dumps(out))
