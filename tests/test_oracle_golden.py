"""Pins the oracle (and the front end's host logic) against the REAL reference's outputs.

For every golden case the inputs are parsed by the front end's own loader (slamem_host.c), matched by the
oracle (oracle/oracle.c) and formatted by the front end's own writer; the result must be byte-identical
to the file the reference wrote (tests/golden/<case>/expected-mems.txt)."""
import numpy as np
import pytest

import hostlib
from golden_cases import CASES, MANIFEST, case_paths, opt_value
from oracle import pyoracle as po


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference_output_file(case):
    opts = MANIFEST[case]["opts"]
    ref_fa, q_fa, exp_mems, _ = case_paths(case)
    both = "-b" in opts
    l = int(opt_value(opts, "-l", 20))
    acgt = 1 if "-n" in opts else 0
    m = int(opt_value(opts, "-m", 0))
    r = opt_value(opts, "-r")
    ref = hostlib.Loaded(ref_fa, 1, acgt, m, r)
    qs = hostlib.Loaded(q_fa, 0, acgt, m, None)
    assert ref.n > 0 and qs.n > 0
    idx = po.OracleIndex(ref.chars)
    off = np.array(qs.offsets, dtype=np.uint64)
    mam = "-mam" in MANIFEST[case].get("tail", [])
    mems, bc = idx.match_batch(np.frombuffer(qs.chars, dtype=np.uint8), off, l, both, mam=mam)
    tri = np.stack([mems["ref_pos"], mems["query_pos"], mems["length"]], axis=1).astype(np.uint32) if len(mems) else \
        np.zeros((0, 3), dtype=np.uint32)
    out = []
    pos = 0
    strands = 2 if both else 1
    for i in range(qs.n):
        for s in range(strands):
            cnt = int(bc[i * strands + s])
            out.append(hostlib.format_block(qs.names[i], s, tri[pos:pos + cnt], ref))
            pos += cnt
    assert b"".join(out) == open(exp_mems, "rb").read()


def test_oracle_matches_bruteforce_definition():
    """Outside the reference's validity domain (tiny texts, (n+1) % 32 == 0) the definition is the judge."""
    rng = np.random.default_rng(3)
    for n, alpha, l in [(31, "ACGT", 3), (63, "AC", 4), (127, "ACGTN", 2), (1023, "ACGT", 8), (5, "A", 1), (200, "AC", 6)]:
        t = bytes(rng.choice(np.frombuffer(alpha.encode(), dtype=np.uint8), size=n))
        idx = po.OracleIndex(t)
        for _ in range(5):
            m = int(rng.integers(1, 2 * n))
            q = bytes(rng.choice(np.frombuffer(alpha.encode(), dtype=np.uint8), size=m))
            a = po.sorted_triples(idx.get_matches(q, l))
            b = po.sorted_triples(po.brute_force_mems(t, q, l))
            assert np.array_equal(a, b)


def test_oracle_structures_match_definitions():
    """SA / LCP / BWT / PSV / NSV against their textbook definitions on small texts."""
    rng = np.random.default_rng(11)
    for n, alpha in [(50, "ACGT"), (200, "AC"), (300, "ACGTN"), (64, "A")]:
        t = bytes(rng.choice(np.frombuffer(alpha.encode(), dtype=np.uint8), size=n))
        idx = po.OracleIndex(t)
        order = {"$": 0, "N": 1, "A": 2, "C": 3, "G": 4, "T": 5}
        codes = [order[chr(c)] for c in t] + [0]
        suffixes = sorted(range(n + 1), key=lambda i: codes[i:])
        assert list(idx.sa) == suffixes
        lcp = [-1]
        for r in range(1, n + 1):
            a, b = codes[suffixes[r - 1]:], codes[suffixes[r]:]
            k = 0
            while k < len(a) and k < len(b) and a[k] == b[k]:
                k += 1
            lcp.append(k)
        lcp.append(-1)
        assert list(idx.lcp) == lcp
        assert list(idx.bwt) == [codes[s - 1] if s else 0 for s in suffixes]
        for top in range(0, n + 1, 7):
            for bot in range(top, min(n, top + 5) + 1):
                d = max(lcp[top], lcp[bot + 1])
                got = idx.enclosing_interval(top, bot)
                if d < 0:
                    assert got[0] == -1
                    continue
                tt, bb = top, bot
                while lcp[tt] >= d:
                    tt -= 1
                while lcp[bb + 1] >= d:
                    bb += 1
                assert got == (d, tt, bb)
        for row in range(0, n + 1, 3):
            assert idx.position_in_text(row) == suffixes[row]
