"""Repeat-heavy inputs: intervals of thousands of rows at depth >= l (the case the wave-cooperative enumeration
of k_find_mems_v3 exists for).  Checked in order against the oracle; also exercises the overflow list
(more than kInlineMems MEMs per strand) and ancestor intervals that still qualify."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_repeat_case(rng, n, unit_len, copies, divergence, nreads, read_len):
    t = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=unit_len)
    starts = rng.choice(n - unit_len, size=copies, replace=False)
    for s in starts:
        u = unit.copy()
        mut = rng.random(unit_len) < divergence
        u[mut] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(mut.sum()))
        t[s:s + unit_len] = u
    reads = []
    for _ in range(nreads):
        a = int(rng.integers(0, unit_len - read_len + 1))
        r = unit[a:a + read_len].copy()
        mut = rng.random(read_len) < 0.01
        r[mut] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(mut.sum()))
        reads.append(r)
    return t.tobytes(), np.concatenate(reads), np.arange(nreads + 1, dtype=np.uint64) * np.uint64(read_len)


@pytest.mark.parametrize("copies,divergence,both", [(500, 0.0, True), (800, 0.02, False), (2000, 0.05, True)])
def test_repeat_family_matches_oracle(copies, divergence, both):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from oracle import pyoracle as po
    from slamem_amd import engine
    rng = np.random.default_rng(copies)
    text, q, off = make_repeat_case(rng, 400_000, 120, copies, divergence, 64, 80)
    o = po.OracleIndex(text)
    om, obc = o.match_batch(q, off, 20, both)
    g = engine.Index.build(text)
    from conftest import search_path
    for path in ("seed", "walk"):  # (seed: the family's windows overflow their buckets or tie -- K8s must leave those strands to K8)
        t0 = time.time()
        with search_path(path):
            gm, goff = g.find_mems(q, off, 20, both)
        dt = time.time() - t0
        assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)), path
        for f in ("ref_pos", "query_pos", "length"):
            assert np.array_equal(gm[f], om[f]), (f, path)
        print(f"repeat family x{copies} div {divergence} ({path}): {len(gm)} MEMs from 64 reads in {dt * 1e3:.1f} ms")
    g.close()


@pytest.mark.parametrize("both", [False, True])
def test_long_queries_sliced_across_lanes(both):
    """Records longer than one slice (4096) are cut into work items scanned by different lanes with a warm-up;
    exact matches longer than the warm-up (1024, 4096, ...) force the in-kernel restart with a longer one.
    Output must equal the oracle's whole-record scan, in order."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from oracle import pyoracle as po
    from slamem_amd import engine
    rng = np.random.default_rng(77)
    n = 300_000
    t = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)

    def mutated(a, b, p):
        s = t[a:b].copy()
        m = rng.random(b - a) < p
        s[m] = rng.choice(alpha, size=int(m.sum()))
        return s
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    qs = [mutated(1000, 21000, 0.01),                                   # 20 kbp, short exact runs
          np.concatenate([mutated(50000, 60000, 0.02), t[100000:130000], mutated(200000, 205000, 0.0)]),  # 30 kbp identical
          t[5:4101].copy(),                                             # exactly one slice + 0
          t[7000:7000 + 4097].copy(),                                   # slice + 1
          comp[t[150000:170000][::-1]],                                 # reverse-complement of 20 kbp, identical
          mutated(250000, 250150, 0.02), np.frombuffer(b"N" * 5000, dtype=np.uint8),
          rng.choice(alpha, size=9000)]
    q = np.concatenate(qs)
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in qs])
    o = po.OracleIndex(t.tobytes())
    om, obc = o.match_batch(q, off, 20, both)
    g = engine.Index.build(t.tobytes())
    gm, goff = g.find_mems(q, off, 20, both)
    assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
    for f in ("ref_pos", "query_pos", "length"):
        assert np.array_equal(gm[f], om[f]), f
    g.close()


def test_reference_with_long_diverged_repeats_builds_exactly():
    """SURVEY.md 8(d) repeat model: 0.5 % of the text copied as 1-10 kbp segments with 1 % divergence, plus one
    exact 40 kbp duplication (LCP 40,000 -> 12 prefix-doubling rounds).  SA / LCP / links must equal the oracle's."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from oracle import pyoracle as po
    from slamem_amd import capi, engine
    rng = np.random.default_rng(123)
    n = 3_000_000
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    t = rng.choice(alpha, size=n)
    copied = 0
    while copied < n // 200:
        L = int(rng.integers(1000, 10000))
        a, b = int(rng.integers(0, n - L)), int(rng.integers(0, n - L))
        seg = t[a:a + L].copy()
        m = rng.random(L) < 0.01
        seg[m] = rng.choice(alpha, size=int(m.sum()))
        t[b:b + L] = seg
        copied += L
    t[2_000_000:2_040_000] = t[500_000:540_000]
    text = t.tobytes()
    o = po.OracleIndex(text)
    g = engine.Index.build(text)
    assert int(g.info.max_lcp) >= 40_000 and int(g.info.sort_rounds) >= 12
    assert np.array_equal(g.download(capi.ARRAY_SA).astype(np.int64), o.sa)
    assert np.array_equal(g.download(capi.ARRAY_LCP).astype(np.int64), o.lcp)
    assert np.array_equal(g.download(capi.ARRAY_PSV).astype(np.int64)[1:n + 1], o.psv[1:n + 1])
    assert np.array_equal(g.download(capi.ARRAY_NSV).astype(np.int64)[1:n + 1], o.nsv[1:n + 1])
    print("build timings", {k: round(v, 1) for k, v in engine.timings().items() if k.startswith("build_")},
          "rounds", int(g.info.sort_rounds))
    g.close()


@pytest.mark.parametrize("chunks,defer,walk", [("0", "1", "0"), ("1", "1", "0"), ("1", "1", "1"), ("1", "0", "1")],
                         ids=["plain", "chunks+queue", "chunks+queue, index walk only", "chunks in the waves, index walk only"])
def test_enumeration_in_chunks_equals_oracle(chunks, defer, walk):
    """K8's kChunk instantiation (enumeration jobs take their places in the overflow list chunk-wise: chosen by itself on
    repeat-rich texts, ArenaHeader::lcp_ge; forced here both ways with SLAMEM_ENUM_CHUNKS in a child process) must give the
    oracle's MEMs in the oracle's order on a text that is mostly one repeat family (thousands of MEMs per strand, nearly all
    through the overflow list), -mem and -mam, reads and a long record in slices, with the capacity retry on the way.
    Round 4: the batch of reads alone takes K8's kDefer instantiation (the jobs go to a queue that k_enum_jobs runs with the
    whole chip, K9 puts the MEM numbers right: SLAMEM_ENUM_DEFER), with and without the seed path in front of it."""
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    child = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from oracle import pyoracle as po
from slamem_amd import engine
rng = np.random.default_rng(99)
unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300)
n = 400_000
text = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
for c in range(700):  # 700 copies of a 300 bp family, 3-12 % diverged, over half of the text
    cp = unit.copy()
    mut = rng.random(300) < rng.uniform(0.03, 0.12)
    cp[mut] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(mut.sum()))
    at = int(rng.integers(0, n - 300))
    text[at:at + 300] = cp
qs = []
for i in range(1500):
    L = int(rng.integers(40, 260))
    a = int(rng.integers(0, n - L))
    q = text[a:a + L].copy()
    mut = rng.random(L) < 0.02
    q[mut] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(mut.sum()))
    qs.append(q)
o = po.OracleIndex(text.tobytes())
g = engine.Index.build(text)
for with_long in (True, False):  # with a record cut into slices (K8's kSliced instantiation), and the reads alone (kDefer)
    qq = qs + [text[100_000:112_000].copy()] if with_long else qs
    off = np.zeros(len(qq) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(x) for x in qq])
    q = np.concatenate(qq)
    for mam in (False, True):
        for l in (12, 25):
            om, obc = o.match_batch(q, off, l, True, mam=mam)
            gm, goff = g.find_mems(q, off, l, True, mam=mam)
            assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)), (with_long, mam, l)
            for f in ("ref_pos", "query_pos", "length"):
                assert np.array_equal(gm[f], om[f]), (with_long, mam, l, f)
            print(mam, l, len(om), int(obc.max()))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", child, root], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, SLAMEM_ENUM_CHUNKS=chunks, SLAMEM_ENUM_DEFER=defer, **({"SLAMEM_SEED_SEARCH": "0"} if walk == "1" else {})),
                       timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = r.stdout.decode().split("\n")
    assert int(lines[0].split()[2]) > 200_000  # the case is enumeration-heavy
