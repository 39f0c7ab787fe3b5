"""Run by tests/test_gpu_parity.py::test_mam_slices_* / test_mem_slices_* in a child process (the SLAMEM_MAM_* /
SLAMEM_SLICE_* switches are read once per process; argv[1] = mam | mem): long strands in slices against the oracle's
whole-strand scan, in order -- -mam: k_find_mams_sliced (speculated start states, verified, wrong guesses scanned
again); -mem: k_slice_states (start states from a warm-up, completed by comparing with the text) + K8.  One JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from slamem_amd import engine  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_gpu_parity import rand_text, pack  # noqa: E402

MAM = (sys.argv[1] if len(sys.argv) > 1 else "mam") == "mam"
rng = np.random.default_rng(4242)
t = np.frombuffer(rand_text(rng, 400_000, "ACGT", 80, max_rep=3000), dtype=np.uint8).copy()
t[330_000:360_000] = t[120_000:150_000]  # a 30 kbp duplicate: warm-ups inside it meet no failed extension on SEVERAL rows
text = t.tobytes()
letters = np.frombuffer(b"ACGT", dtype=np.uint8)


def mutate(a, rate):
    a = a.copy()
    m = rng.random(a.shape[0]) < rate
    a[m] = rng.choice(letters, size=int(m.sum()))
    return a


withn = mutate(t[200_000:290_000], 0.01)
withn[rng.integers(0, withn.shape[0], 40)] = ord("N")
queries = [
    mutate(t[30_000:230_000], 0.01).tobytes(),          # 49 slices, substitutions every 100 letters
    t[100_000:220_000].tobytes(),                       # an exact copy: one 120 kbp match across 30 slices (the warm-up grows)
    mutate(t[5_000:105_000], 0.001).tobytes(),          # matches of about 1000 letters: around the default warm-up
    withn.tobytes(),                                    # the letter N resets the scan
    t[50_000:50_100].tobytes(), b"", b"ACGT",           # short records beside the long ones
    mutate(t[300_000:312_288], 0.02).tobytes(),         # exactly 3 slices
    t[::-1][10_000:60_000].copy().tobytes(),            # unrelated
    np.concatenate([t[10_000:70_000], t[200_000:260_000]]).tobytes(),  # two exact stretches on different diagonals: the chain
                                                                       # of open states breaks where they meet
    t[100_000:170_000].tobytes(),                       # across the duplicated stretch (two rows for 30 kbp)
    np.concatenate([t[340_000:352_000], mutate(t[40_000:52_000], 0.002)]).tobytes(),
]
qq, off = pack(queries)
o = po.OracleIndex(text)
g = engine.Index.build(text, "cuda:0")
out = {"cases": {}}
ok_all = True
for min_len, both in ((20, True), (12, False), (50, True)):
    om, obc = o.match_batch(qq, off, min_len, both, mam=MAM)
    gm, goff = g.find_mems(qq, off, min_len, both, mam=MAM)
    ok = np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)) and len(gm) == len(om) and all(
        np.array_equal(gm[f], om[f]) for f in ("ref_pos", "query_pos", "length"))
    out["cases"]["l%d_%s" % (min_len, "both" if both else "fwd")] = {"equal_in_order": bool(ok), "mams": int(len(om))}
    ok_all = ok_all and ok
out["all_equal"] = bool(ok_all)
print(json.dumps(out))
