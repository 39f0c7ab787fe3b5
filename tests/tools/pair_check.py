"""The genome pair of configs[0] through the device path (one long record, sliced) against the oracle -- prints what differs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden_cases import ecoli_like_pair
from slamem_amd import engine
from oracle import pyoracle as po
ref, qry = ecoli_like_pair()
idx = engine.Index.build(ref, "cuda:0")
offsets = np.array([0, len(qry)], dtype=np.uint64)
mems, boff = idx.find_mems(qry, offsets, 20, True)
o = po.OracleIndex(ref.tobytes())
om, obc = o.match_batch(qry, offsets, 20, True)
print("engine", len(mems), boff, "oracle", len(om), obc)
if len(mems) == len(om):
    for f in ("ref_pos", "query_pos", "length"):
        d = np.nonzero(mems[f] != om[f])[0]
        print(f, "differences:", len(d), d[:5])
else:
    se = set(map(tuple, mems.tolist())); so = set(map(tuple, om.tolist()))
    print("missing", sorted(so - se)[:10], "extra", sorted(se - so)[:10])
