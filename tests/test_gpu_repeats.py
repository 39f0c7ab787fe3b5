"""Repeat-heavy inputs: intervals of thousands of rows at depth >= l (the case the wave-cooperative enumeration
of k_find_mems_v3 exists for).  Checked in order against the oracle; also exercises the overflow list
(more than kInlineMems MEMs per strand) and ancestor intervals that still qualify."""
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
    t0 = time.time()
    gm, goff = g.find_mems(q, off, 20, both)
    dt = time.time() - t0
    assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
    for f in ("ref_pos", "query_pos", "length"):
        assert np.array_equal(gm[f], om[f]), f
    print(f"repeat family x{copies} div {divergence}: {len(gm)} MEMs from 64 reads in {dt * 1e3:.1f} ms")
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
