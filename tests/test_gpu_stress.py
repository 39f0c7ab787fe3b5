"""Randomised end-to-end stress of the search state machine (k_prefilter + k_find_mems_v3 + K9) against the oracle:
alphabets with and without N, planted repeats, N runs, reads from 1 to 9000 letters (slices, warm-up restarts),
minimum lengths on both sides of the presence filter's k, both strand modes.  Equality is in emission order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_case(rng):
    alpha = [b"AC", b"ACG", b"ACGT", b"ACGTN", b"A"][int(rng.integers(0, 5))]
    a = np.frombuffer(alpha, dtype=np.uint8)
    l = int(rng.choice([1, 2, 3, 5, 10, 12, 13, 16, 17, 18, 19, 20, 21, 30, 50]))
    small = l <= 5 or len(alpha) <= 2  # these produce MEMs by the million on long inputs: keep them short
    n = int(rng.integers(1, 3000 if small else 40000))
    t = rng.choice(a, size=n)
    for _ in range(int(rng.integers(0, 6))):
        if n < 40:
            break
        L = int(rng.integers(10, min(n // 2, 6000) + 1))
        x, y = int(rng.integers(0, n - L + 1)), int(rng.integers(0, n - L + 1))
        t[y:y + L] = t[x:x + L].copy()
    if len(alpha) == 5 and n > 400 and rng.random() < 0.5:
        L = int(rng.integers(1, 300))
        x = int(rng.integers(0, n - L))
        t[x:x + L] = ord("N")
    comp = np.arange(256, dtype=np.uint8)
    for u, v in zip(b"ACGT", b"TGCA"):
        comp[u] = v
    qs = []
    for k in range(int(rng.integers(1, 40))):
        mode = rng.random()
        qlen = int(rng.integers(1, 9001)) if (rng.random() < 0.15 and not small) else int(rng.integers(1, 150 if small else 400))
        if mode < 0.7 and n >= 2:
            qlen = min(qlen, n)
            x = int(rng.integers(0, n - qlen + 1))
            q = t[x:x + qlen].copy()
            m = rng.random(qlen) < rng.choice([0.0, 0.01, 0.05])
            q[m] = rng.choice(a, size=int(m.sum()))
            if rng.random() < 0.4:
                q = comp[q[::-1]]
        else:
            q = rng.choice(a, size=qlen)
        qs.append(q)
    if rng.random() < 0.3:
        qs.append(np.zeros(0, dtype=np.uint8))
    return t.tobytes(), qs, l, bool(rng.random() < 0.5)


@pytest.mark.parametrize("mam", [False, True], ids=["mem", "mam"])
@pytest.mark.parametrize("seed", range(40))
def test_random_case_matches_oracle_in_order(seed, mam):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from oracle import pyoracle as po
    from slamem_amd import engine
    rng = np.random.default_rng(1000 + seed)
    text, qs, l, both = random_case(rng)
    q = np.concatenate(qs) if qs else np.zeros(0, dtype=np.uint8)
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in qs])
    o = po.OracleIndex(text)
    om, obc = o.match_batch(q, off, l, both, mam=mam)
    assert len(om) <= 3_000_000  # all 80 cases run: the largest of these seeds has 590,034 MEMs (checked on the CPU)
    g = engine.Index.build(text)
    from conftest import search_path
    for path in (("seed", "walk") if not mam else ("seed",)):  # (-mam never takes the seed path)
        with search_path(path):
            gm, goff = g.find_mems(q, off, l, both, mam=mam)
        assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)), (len(text), l, both, path)
        for f in ("ref_pos", "query_pos", "length"):
            assert np.array_equal(gm[f], om[f]), (f, len(text), l, both, path)
    g.close()


@pytest.mark.parametrize("l", [12, 13, 14, 15, 16, 17, 18, 19, 20, 25])
def test_prefilter_never_drops_a_strand_with_a_barely_long_enough_mem(l):
    """The presence prefilter (one- and two-level, K8a) on its worst case: strands whose ONLY match is l .. l+3 letters
    long, placed at the strand's ends, in the middle and across the 4096-position slice borders of long records, on both
    strands.  k = 12 for this text, so l = 12..17 takes the two-level path, l >= 18 the one-level path."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from oracle import pyoracle as po
    from slamem_amd import engine
    rng = np.random.default_rng(500 + l)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = 50_001
    t = rng.choice(acgt, size=n)
    comp = np.arange(256, dtype=np.uint8)
    for u, v in zip(b"ACGT", b"TGCA"):
        comp[u] = v
    qs = []
    for qlen, places in [(150, [0, 1, 2, 3, 4, 5, 60, 61, 62, 63, 64, None]),
                         (9000, [0, 4090, 4092, 4094, 4096, 4097, 4100, 8180, 8190, None])]:
        for d in range(4):
            for at in places:
                m = l + d
                pos = qlen - m if at is None else at
                if pos + m > qlen:
                    continue
                q = rng.choice(acgt, size=qlen)  # junk: random 12-mers rarely occur in a 50 kbp text
                x = int(rng.integers(1, n - m - 1))
                q[pos:pos + m] = t[x:x + m]
                if pos > 0:  # make the copy maximal: the letters next to it differ from the text's
                    q[pos - 1] = acgt[(int(np.searchsorted(acgt, t[x - 1])) + 1) % 4]
                if pos + m < qlen:
                    q[pos + m] = acgt[(int(np.searchsorted(acgt, t[x + m])) + 1) % 4]
                qs.append(q if (len(qs) % 2 == 0) else comp[q[::-1]])
    q = np.concatenate(qs)
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in qs])
    o = po.OracleIndex(t.tobytes())
    om, obc = o.match_batch(q, off, l, True)
    g = engine.Index.build(t.tobytes())
    assert g.info.filter_k == 12
    gm, goff = g.find_mems(q, off, l, True)
    assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
    for f in ("ref_pos", "query_pos", "length"):
        assert np.array_equal(gm[f], om[f]), f
    assert len(om) >= len(qs)  # every record has its planted match (on one of the two strands)
    g.close()
