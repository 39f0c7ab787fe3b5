#!/usr/bin/env python3
"""Adversarial reference for the index build: what real chromosomes contain and random texts do not -- a multi-Mbp
run of N (centromere gap), a tandem repeat, a homopolymer run, a long exact duplication.  Checks, without any CPU
oracle (sizes are too large): SA is a permutation, sampled neighbours in the suffix order are in order and their LCP
is exact (compared in the text), sampled MEMs of reads are real and maximal.

    tests/tools/robust_check.py [scale]      scale 1.0 = 40 Mbp text, 8 Mbp N run, 2 Mbp tandem, 1 Mbp poly-A, 3 Mbp copy
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from slamem_amd import capi, engine  # noqa: E402


def first_diff(x, y):
    """index of the first differing element of two equally long arrays, or len"""
    step = 1 << 16
    for s in range(0, len(x), step):
        d = np.nonzero(x[s:s + step] != y[s:s + step])[0]
        if len(d):
            return s + int(d[0])
    return len(x)


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    n = int(40_000_000 * scale) | 1
    dev = torch.device("cuda:0")
    ref = engine.synth_reference(n, 7, dev)
    t = ref.cpu().numpy().copy()
    u = lambda x: int(x * scale)
    feats = os.environ.get("ROBUST_FEATURES", "ntad")  # which of the four structures to plant
    if "n" in feats:
        t[u(5_000_000):u(13_000_000)] = ord("N")
    tand = np.frombuffer(b"ACGT" * (u(2_000_000) // 4), dtype=np.uint8)
    if "t" in feats:
        t[u(20_000_000):u(20_000_000) + len(tand)] = tand
    if "a" in feats:
        t[u(25_000_000):u(26_000_000)] = ord("A")
    if "d" in feats:
        t[u(30_000_000):u(33_000_000)] = t[u(15_000_000):u(18_000_000)]
    ref = torch.from_numpy(t).to(dev)
    torch.cuda.synchronize()
    t0 = time.time()
    idx = engine.Index.build(ref, dev)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    tm = {k: round(v, 1) for k, v in engine.timings().items() if k.startswith("build_")}
    sa = idx.download(capi.ARRAY_SA).astype(np.int64)
    lcp = idx.download(capi.ARRAY_LCP).astype(np.int64)
    assert sa[0] == n and np.array_equal(np.sort(sa), np.arange(n + 1))
    rng = np.random.default_rng(1)
    code = np.zeros(256, dtype=np.uint8)  # the index's letter order: $ < N < A < C < G < T
    for k, ch in enumerate(b"NACGT"):
        code[ch] = k + 1
    rows = np.concatenate([rng.integers(1, n, size=3000), np.argsort(lcp[: n + 1])[-200:]])
    bad = 0
    for i in rows:
        i = int(i)
        if i < 1 or i > n:
            continue
        a, b, l = int(sa[i - 1]), int(sa[i]), int(lcp[i])
        m = min(n - a, n - b)
        k = first_diff(t[a:a + min(m, l + 1)], t[b:b + min(m, l + 1)])
        ok = k == l and (l == m and a > b or l < m and code[t[a + l]] < code[t[b + l]]) if i > 1 else l == 0
        bad += not ok
    # reads from everywhere but the tandem / homopolymer regions (their MEM lists are astronomically long) and the N run
    nreads, L = 200_000, 150
    starts = rng.integers(0, n - L, size=nreads)
    keep = ~(((starts > u(4_999_000)) & (starts < u(13_001_000))) | ((starts > u(19_999_000)) & (starts < u(22_001_000))) | ((starts > u(24_999_000)) & (starts < u(26_001_000))))
    starts = starts[keep]
    nreads = len(starts)
    rd = t[starts[:, None] + np.arange(L)[None, :]].copy()
    mut = rng.random(rd.shape) < 0.02
    rd[mut] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(mut.sum()))
    off = np.arange(nreads + 1, dtype=np.uint64) * L
    t1 = time.time()
    mems, boff = idx.find_mems(rd.reshape(-1), off, 50, True)
    search_s = time.time() - t1
    blk = np.repeat(np.arange(2 * nreads), np.diff(boff.astype(np.int64)))
    comp = np.arange(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    badm = 0
    sel = rng.choice(len(mems), size=min(len(mems), 30_000), replace=False)
    for i in sel:
        a, b, c, g = int(mems["ref_pos"][i]), int(mems["query_pos"][i]), int(mems["length"][i]), int(blk[i])
        r = rd[g >> 1]
        if g & 1:
            r = comp[r[::-1]]
        ok = c >= 50 and (t[a:a + c] == r[b:b + c]).all()
        ok = ok and (a == 0 or b == 0 or t[a - 1] != r[b - 1]) and (a + c == n or b + c == L or t[a + c] != r[b + c])
        badm += not ok
    print(json.dumps({"n": n, "build_wall_s": round(build_s, 3), "build_ms": tm, "sort_rounds": int(idx.info.sort_rounds),
                      "max_lcp": int(idx.info.max_lcp), "rows_checked": int(len(rows)), "rows_bad": int(bad),
                      "reads": int(nreads), "mems": int(len(mems)), "mems_checked": int(len(sel)), "mems_bad": int(badm),
                      "search_wall_s": round(search_s, 3)}))
    assert bad == 0 and badm == 0


if __name__ == "__main__":
    main()
