import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_seed as T
from slamem_amd import engine
from oracle import pyoracle as po
from conftest import search_path
rng = np.random.default_rng(13)
n = 80_000
t = rng.choice(T.ACGT, size=n)
for _ in range(6):
    p = int(rng.integers(100, n - 400))
    t[p:p + int(rng.integers(1, 200))] = ord("N")
for _ in range(40):
    t[int(rng.integers(0, n))] = ord("N")
qs = T.reads_from(rng, t, 1500, 120, 0.02)
for i in range(0, len(qs), 9):
    qs[i] = qs[i].copy()
    qs[i][int(rng.integers(0, 120))] = ord("NRYKMnrw"[i % 8])
for i in range(1, len(qs), 50):
    qs[i] = np.frombuffer(qs[i].tobytes().lower(), dtype=np.uint8)
q, off = T.pack(qs)
o = po.OracleIndex(t.tobytes())
om, obc = o.match_batch(q, off, 20, True)
g = engine.Index.build(t.tobytes())
for path in ("seed", "walk"):
    with search_path(path):
        gm, goff = g.find_mems(q, off, 20, True)
    cnt = np.diff(goff.astype(np.int64))
    bad = np.nonzero(cnt != obc.astype(np.int64))[0]
    print(path, "blocks that differ:", bad[:20], len(bad))
    ob = np.concatenate([[0], np.cumsum(obc)]).astype(np.int64)
    for b in bad[:4]:
        r = b // 2
        print(" block", b, "read", r, qs[r].tobytes())
        print("   engine", gm[int(goff[b]):int(goff[b + 1])])
        print("   oracle", om[ob[b]:ob[b + 1]])
        for m in om[ob[b]:ob[b + 1]]:
            print("   text at match:", t[max(0, m[0] - 3):m[0] + m[2] + 3].tobytes())
