"""Randomised soak of the seed-and-compare path (K8s): for a stream of random cases -- text size, repeat structure (exact and
diverged copies of short and long segments, tandem arrays, reverse-complement copies, planted palindromes), letters that are not
A,C,G,T in the text or only in the reads, read lengths (fixed or mixed, up to 384), substitution rate, minimum length, one or
both strands -- the engine's answer on the default path (K8s, K8 for what it leaves) must equal, row for row in order, its
answer with SLAMEM_SEED_SEARCH=0 (the prefilter and the index walk, which the test suites pin to the oracle), and the oracle's
own answer on the smaller cases (checker only); in a third of the cases the reads also go through slamem_stream_submit in
windows of the one offsets array, batch by batch.  Prints one JSON line per case and a summary; exits 1 on the first difference.

    python tests/tools/seed_soak.py [seconds] [first_seed]        (SOAK_STRANDS=1|2: every case with one strand / both)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from slamem_amd import engine  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (checker)

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
    COMP[a] = b


def rc(a):
    return COMP[a[::-1]]


def mutate(rng, a, rate):
    a = a.copy()
    m = rng.random(len(a)) < rate
    a[m] = rng.choice(ACGT, size=int(m.sum()))
    return a


def make_text(rng, n):
    t = rng.choice(ACGT, size=n)
    kinds = []
    for _ in range(int(rng.integers(0, 7))):
        kind = int(rng.integers(0, 6))
        kinds.append(kind)
        if kind == 0:    # exact copies of a segment
            ln, copies = int(rng.integers(20, min(3000, n // 8))), int(rng.integers(2, 30))
            seg = t[int(rng.integers(0, n - ln)):][:ln].copy()
            for _ in range(copies):
                y = int(rng.integers(0, n - ln))
                t[y:y + ln] = seg
        elif kind == 1:  # diverged copies
            ln, copies, div = int(rng.integers(50, min(2000, n // 8))), int(rng.integers(2, 60)), float(rng.choice([0.003, 0.01, 0.05, 0.12]))
            seg = t[int(rng.integers(0, n - ln)):][:ln].copy()
            for _ in range(copies):
                y = int(rng.integers(0, n - ln))
                t[y:y + ln] = mutate(rng, seg, div)
        elif kind == 2:  # a tandem array
            unit, copies = int(rng.integers(2, 200)), int(rng.integers(3, 200))
            ln = min(unit * copies, n // 6)
            u = rng.choice(ACGT, size=unit)
            y = int(rng.integers(0, n - ln))
            t[y:y + ln] = np.tile(u, copies + 1)[:ln]
        elif kind == 3:  # reverse-complement copies
            ln = int(rng.integers(30, min(1500, n // 8)))
            seg = t[int(rng.integers(0, n - ln)):][:ln].copy()
            for _ in range(int(rng.integers(1, 6))):
                y = int(rng.integers(0, n - ln))
                t[y:y + ln] = rc(seg)
        elif kind == 4:  # palindromes (own reverse complement)
            for _ in range(int(rng.integers(1, 40))):
                h = rng.choice(ACGT, size=int(rng.integers(4, 30)))
                p = np.concatenate([h, rc(h)])
                y = int(rng.integers(0, n - len(p)))
                t[y:y + len(p)] = p
        else:            # words in a few copies (ties)
            for _ in range(int(rng.integers(1, 30))):
                w = rng.choice(ACGT, size=int(rng.integers(18, 40)))
                for _ in range(int(rng.integers(2, 16))):
                    y = int(rng.integers(0, n - len(w)))
                    t[y:y + len(w)] = w
    return t, kinds


def one(seed):
    rng = np.random.default_rng(seed)
    n = int(10 ** rng.uniform(3.5, 6.5))
    t, kinds = make_text(rng, n)
    text_n = rng.random() < 0.3
    if text_n:
        for _ in range(int(rng.integers(1, 20))):
            y = int(rng.integers(0, n - 1))
            t[y:y + int(rng.integers(1, 300))] = ord("N")
    long_reads = rng.random() < 0.2
    maxlen = 384 if long_reads else 192
    fixed = int(rng.integers(30, maxlen + 1)) if rng.random() < 0.6 else 0
    nreads = int(rng.integers(50, 6000))
    sub = float(rng.choice([0.0, 0.01, 0.02, 0.05, 0.1]))
    reads_n = rng.random() < 0.4
    qs = []
    for i in range(nreads):
        ln = fixed if fixed else int(rng.integers(12, maxlen + (40 if rng.random() < 0.02 else 1)))
        ln = min(ln, n)
        x = int(rng.integers(0, n - ln + 1))
        r = mutate(rng, t[x:x + ln], sub)
        if rng.random() < 0.5:
            r = rc(r)
        if reads_n and rng.random() < 0.15:
            for _ in range(int(rng.integers(1, 4))):
                y = int(rng.integers(0, ln))
                r[y:y + int(rng.integers(1, 3))] = ord("NnRyK"[int(rng.integers(0, 5))])
        if rng.random() < 0.01:
            r = np.frombuffer(r.tobytes().lower(), dtype=np.uint8)
        qs.append(r)
    both = rng.random() < 0.8
    if os.environ.get("SOAK_STRANDS") in ("1", "2"):  # (a soak of one setting)
        both = os.environ["SOAK_STRANDS"] == "2"
    q = np.concatenate(qs)
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in qs])
    g = engine.Index.build(t.tobytes())
    try:
        k = int(g.info.seed_k)
        lo = k + 2 if k else 12
        l = int(rng.integers(lo, lo + 12)) if rng.random() < 0.7 else int(rng.integers(lo, 70))
        step = os.environ.get("SLAMEM_SEED_STEP")
        if rng.random() < 0.25:
            os.environ["SLAMEM_SEED_STEP"] = str(int(rng.integers(1, 7)))
        else:
            os.environ.pop("SLAMEM_SEED_STEP", None)
        forced = os.environ.get("SLAMEM_SEED_STEP", "")
        os.environ.pop("SLAMEM_SEED_SEARCH", None)
        sm, so = g.find_mems(q, off, l, both)
        os.environ["SLAMEM_SEED_SEARCH"] = "0"
        wm, wo = g.find_mems(q, off, l, both)
        os.environ.pop("SLAMEM_SEED_SEARCH", None)
        if step is None:
            os.environ.pop("SLAMEM_SEED_STEP", None)
        else:
            os.environ["SLAMEM_SEED_STEP"] = step
        ok = np.array_equal(so, wo) and all(np.array_equal(sm[f], wm[f]) for f in ("ref_pos", "query_pos", "length"))
        streamed = False
        if ok and rng.random() < 0.3 and len(qs) >= 8:
            # the same reads through slamem_stream_submit in windows of the one offsets array (offsets that do not start at 0)
            streamed = True
            strands = 2 if both else 1
            cuts = sorted(set([0, len(qs)] + [int(x) for x in rng.integers(1, len(qs), size=int(rng.integers(1, 4)))]))
            maxq = max(int(off[cuts[i + 1]] - off[cuts[i]]) for i in range(len(cuts) - 1))
            stx = engine.Stream(g, 2, maxq + 64, max(cuts[i + 1] - cuts[i] for i in range(len(cuts) - 1)), both)
            try:
                wcnt = np.diff(wo.astype(np.int64))
                for i in range(len(cuts) - 1):
                    w = np.ascontiguousarray(off[cuts[i]: cuts[i + 1] + 1])
                    stx.submit(q, w, l)
                    m2, b2, _ = stx.next()
                    lo, hi = int(wo[cuts[i] * strands]), int(wo[cuts[i + 1] * strands])
                    ok = ok and np.array_equal(np.diff(b2.astype(np.int64)), wcnt[cuts[i] * strands: cuts[i + 1] * strands]) and all(
                        np.array_equal(m2[f], wm[f][lo:hi]) for f in ("ref_pos", "query_pos", "length"))
            finally:
                stx.close()
        checked_oracle = False
        if ok and n <= 300_000 and len(qs) <= 2500:
            u = np.frombuffer(q.tobytes().upper(), dtype=np.uint8).copy()
            u[~np.isin(u, ACGT)] = ord("N")
            om, obc = po.OracleIndex(t.tobytes()).match_batch(u, off, l, both)
            ok = np.array_equal(np.diff(so.astype(np.int64)), obc.astype(np.int64)) and all(
                np.array_equal(sm[f], om[f]) for f in ("ref_pos", "query_pos", "length"))
            checked_oracle = True
        rec = {"seed": seed, "n": n, "kinds": kinds, "text_n": bool(text_n), "reads": nreads, "fixed_len": fixed, "long": bool(long_reads),
               "sub": sub, "reads_n": bool(reads_n), "both": bool(both), "l": l, "seed_k": k, "step": forced, "mems": int(len(sm)),
               "oracle": checked_oracle, "stream": streamed, "ok": bool(ok)}
        print(json.dumps(rec), flush=True)
        return ok, len(sm)
    finally:
        g.close()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    cases = mems = 0
    while time.time() - t0 < seconds:
        ok, m = one(seed)
        cases += 1
        mems += m
        if not ok:
            print(json.dumps({"summary": "DIFFERENCE", "seed": seed, "cases": cases}), flush=True)
            sys.exit(1)
        seed += 1
    print(json.dumps({"summary": "all equal", "cases": cases, "mems": mems, "seconds": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
