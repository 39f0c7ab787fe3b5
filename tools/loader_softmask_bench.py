"""Load seconds of a soft-masked reference (half of it lower case in stretches, 60-column lines, a few records, some N runs):
the whole-line path of the loader against its byte loop (SLAMEM_LOADER_BYTEWISE=1).  CPU only.
    python tools/loader_softmask_bench.py [letters = 1e9]"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
path = "/tmp/softmask_%d.fa" % n
if not os.path.exists(path):
    rng = np.random.default_rng(1)
    with open(path, "wb") as f:
        left, rec = n, 0
        while left > 0:
            m = min(left, 125_000_000)
            x = rng.integers(0, 4, size=m, dtype=np.uint8)
            a = np.frombuffer(b"ACGT", dtype=np.uint8)[x]
            # soft-masked stretches: lower case in alternating blocks of 100..2000 letters
            edges = np.cumsum(rng.integers(100, 2000, size=m // 1000 + 2))
            edges = edges[edges < m]
            mask = np.zeros(m, dtype=bool)
            for i in range(0, len(edges) - 1, 2):
                mask[edges[i]:edges[i + 1]] = True
            a = np.where(mask, a | 0x20, a).astype(np.uint8)
            a[m // 3: m // 3 + 50_000] = ord("N")
            lines = a[: m - m % 60].reshape(-1, 60)
            out = np.empty((lines.shape[0], 61), dtype=np.uint8)
            out[:, :60] = lines
            out[:, 60] = 10
            f.write(b">chr%d soft-masked\n" % rec)
            f.write(out.tobytes())
            left -= m
            rec += 1
child = r"""
import sys, time
sys.path.insert(0, sys.argv[1] + "/tests")
import hostlib, ctypes as C
s = hostlib.SeqSet()
t0 = time.time()
n = hostlib.lib().slh_load_file(sys.argv[2].encode(), int(sys.argv[3]), 0, 0, None, 1, 100, C.byref(s), None)
print(time.time() - t0, s.total, n)
"""
res = {"letters": n, "file_bytes": os.path.getsize(path)}
for merge in (1, 0):
    for mode in ("1", "0", "1", "0"):
        r = subprocess.run([sys.executable, "-c", child, ROOT, path, str(merge)], stdout=subprocess.PIPE, env=dict(os.environ, SLAMEM_LOADER_BYTEWISE=mode))
        sec, total, nrec = r.stdout.decode().split()
        res.setdefault(("reference" if merge else "reads") + (" byte loop" if mode == "1" else " whole lines"), []).append(round(float(sec), 3))
    res["records"] = int(nrec)
print(json.dumps(res))
