"""Quick parity + counters of the seed-and-compare path (K8s, k_seed_mems) against the oracle (checker only) on small
inputs: random and repeat-rich texts, reads with substitutions on both strands, reads at the text's ends, N in reads and text,
several minimum lengths and read lengths.  Usage: python tools/seed_check.py [cases]"""
import os
import sys
import ctypes as C

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamem_amd import engine, synth, capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (checker)


def stats_for(idx, q, offsets, min_len, both):
    L = capi.lib()
    L.slamem_search_stats_enable(1)
    idx.find_mems(q, offsets, min_len, both)
    st = capi.SearchStats()
    L.slamem_get_search_stats(C.byref(st))
    L.slamem_search_stats_enable(0)
    d = st.as_dict()
    return {k: d[k] for k in ("seed_reads", "seed_windows", "seed_compares", "seed_letter_masks", "seed_mems",
                              "seed_strands_left", "survivors", "mems", "items")}


def one(case, n, nreads, rlen, min_len, both, sub=0.02, seed=1, with_n=False, repeats=False, ends=False):
    rng = np.random.default_rng(seed)
    ref = synth.make_reference(n, seed=seed)
    if repeats:
        ref = ref.copy()
        unit = ref[1000:1300].copy()
        for c in range(40):
            p = int(rng.integers(2000, n - 400))
            u = unit.copy()
            for _ in range(int(rng.integers(0, 6))):
                u[int(rng.integers(0, 300))] = ord("ACGT"[int(rng.integers(0, 4))])
            ref[p:p + 300] = u
    if with_n:
        ref = ref.copy()
        for _ in range(5):
            p = int(rng.integers(100, n - 400))
            ref[p:p + int(rng.integers(1, 200))] = ord("N")
    reads = synth.make_reads(ref, 0, nreads, rlen, sub, seed=seed + 7, rc_percent=50 if both else 0).copy()
    if ends:  # reads drawn from the first / last letters of the text
        for i in range(0, nreads, 7):
            reads[i] = ref[:rlen] if (i // 7) % 2 == 0 else ref[n - rlen:]
    if with_n:
        for i in range(0, nreads, 11):
            reads[i, int(rng.integers(0, rlen))] = ord("N")
    offsets = (np.arange(nreads + 1, dtype=np.uint64) * np.uint64(rlen))
    idx = engine.Index.build(ref, "cuda:0")
    mems, boff = idx.find_mems(reads.reshape(-1), offsets, min_len, both)
    o = po.OracleIndex(ref.tobytes())
    om, obc = o.match_batch(reads.reshape(-1), offsets, min_len, both)
    ok = np.array_equal(np.diff(boff.astype(np.int64)), obc.astype(np.int64))
    if ok:
        for f in ("ref_pos", "query_pos", "length"):
            ok = ok and np.array_equal(mems[f], om[f])
    st = stats_for(idx, reads.reshape(-1), offsets, min_len, both)
    print(f"{case:28s} n={n} reads={nreads}x{rlen} l={min_len} b={int(both)}: {'OK ' if ok else 'BAD'} mems={len(mems)} oracle={len(om)} {st}",
          flush=True)
    if not ok:
        cnt = np.diff(boff.astype(np.int64))
        badb = np.nonzero(cnt != obc.astype(np.int64))[0]
        print("   first blocks with different counts:", badb[:10], cnt[badb[:10]], obc[badb[:10]])
        if len(badb) == 0:
            for f in ("ref_pos", "query_pos", "length"):
                d = np.nonzero(mems[f] != om[f])[0]
                if len(d):
                    i = d[0]
                    print("   first difference at MEM", i, "engine", mems[i], "oracle", om[i])
                    break
        else:
            b = int(badb[0])
            ob = np.concatenate([[0], np.cumsum(obc)]).astype(np.int64)
            print("   engine:", mems[int(boff[b]):int(boff[b + 1])], "\n   oracle:", om[ob[b]:ob[b + 1]])
    idx.close()
    return ok


def main():
    ok = True
    ok &= one("smoke", 200_000, 2_000, 150, 20, True, seed=7)
    ok &= one("forward only", 200_000, 2_000, 150, 20, False, seed=8)
    ok &= one("l=50", 300_000, 3_000, 150, 50, True, seed=9)
    ok &= one("short reads", 100_000, 3_000, 36, 20, True, seed=10)
    ok &= one("len 192", 100_000, 1_000, 192, 25, True, seed=11)
    ok &= one("len 193 (left to K8)", 100_000, 500, 193, 25, True, seed=12)
    ok &= one("ends of the text", 50_000, 1_000, 100, 20, True, seed=13, ends=True)
    ok &= one("N in text and reads", 80_000, 2_000, 120, 20, True, seed=14, with_n=True)
    ok &= one("repeats", 120_000, 3_000, 150, 20, True, seed=15, repeats=True)
    ok &= one("repeats l=30", 120_000, 3_000, 150, 30, True, seed=16, repeats=True, sub=0.01)
    ok &= one("tiny text", 5_000, 500, 80, 20, True, seed=17)
    ok &= one("1 Mbp", 1_000_000, 20_000, 150, 20, True, seed=18)
    ok &= one("high divergence", 200_000, 3_000, 150, 18, True, seed=19, sub=0.08)
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
