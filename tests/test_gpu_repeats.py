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
        pytest.skip("needs an MI355X")
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
