"""K8s (k_seed_mems: seed-and-compare for reads, slamem_amd/csrc/mem_search.hip) against the oracle, in emission order, on the
inputs that decide whether it is exact: windows that occur several times, ties that need the suffix order, palindromic
windows, letters that are not A,C,G,T in the reads and in the text, the ends of the text, matches that are barely long
enough, reads of every length in one batch.  Each case also says -- from the kernel's own counters -- that the seed path
really ran and how many strands it left to the index walk (K8)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.arange(256, dtype=np.uint8)
for _u, _v in zip(b"ACGTacgt", b"TGCAtgca"):
    COMP[_u] = _v


@pytest.fixture(scope="module")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from slamem_amd import engine
    return engine


def rc(a):
    return COMP[a[::-1]]


def pack(qs):
    q = np.concatenate(qs) if qs else np.zeros(0, dtype=np.uint8)
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in qs])
    return q, off


def mutate(rng, a, rate):
    a = a.copy()
    m = rng.random(len(a)) < rate
    a[m] = rng.choice(ACGT, size=int(m.sum()))
    return a


def seed_stats(eng, idx, q, off, l, both):
    import ctypes as C
    from slamem_amd import capi
    L = capi.lib()
    L.slamem_search_stats_enable(1)
    try:
        idx.find_mems(q, off, l, both)
        st = capi.SearchStats()
        capi.check(L.slamem_get_search_stats(C.byref(st)))
    finally:
        L.slamem_search_stats_enable(0)
    return st.as_dict()


def normalised(a):
    """What the reference's loader hands to GetMatches (sequence.c:61-81): upper case, every other letter N."""
    u = np.frombuffer(a.tobytes().upper(), dtype=np.uint8).copy()
    u[~np.isin(u, ACGT)] = ord("N")
    return u


def check(eng, text, qs, l, both, expect_seed=True, max_left_frac=None, min_left=0):
    """engine == oracle in order (counters instantiation too); returns the counters.  The engine gets the reads as they are
    (any case, any letter), the oracle the normalised ones."""
    from oracle import pyoracle as po
    text = text.tobytes() if isinstance(text, np.ndarray) else text
    q, off = pack(qs)
    o = po.OracleIndex(text)
    om, obc = o.match_batch(pack([normalised(x) for x in qs])[0], off, l, both)
    g = eng.Index.build(text)
    try:
        gm, goff = g.find_mems(q, off, l, both)
        assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
        for f in ("ref_pos", "query_pos", "length"):
            assert np.array_equal(gm[f], om[f]), f
        st = seed_stats(eng, g, q, off, l, both)
        assert st["mems"] == len(om)
        if expect_seed:
            assert g.info.seed_k >= 4 and l >= g.info.seed_k + 2
            assert st["seed_reads"] == len(qs), "the seed path did not take this batch"
            assert st["survivors"] <= st["seed_strands_left"]  # (what K8 scans: the strands left, less those the presence filter proves empty)
            if max_left_frac is not None:
                assert st["seed_strands_left"] <= max_left_frac * st["items"], st
            assert st["seed_strands_left"] >= min_left, st
        else:
            assert st["seed_reads"] == 0
        st["seed_k"] = int(g.info.seed_k)
        st["oracle_block_counts"] = obc
        return st, om
    finally:
        g.close()


def reads_from(rng, t, count, length, sub, rc_share=0.5):
    out = []
    for _ in range(count):
        x = int(rng.integers(0, len(t) - length + 1))
        r = mutate(rng, t[x:x + length], sub)
        out.append(rc(r) if rng.random() < rc_share else r)
    return out


@pytest.mark.parametrize("n,count,length,l,both,sub", [
    (200_000, 2000, 150, 20, True, 0.02), (200_000, 2000, 150, 20, False, 0.02), (300_000, 3000, 150, 50, True, 0.02),
    (100_000, 3000, 36, 20, True, 0.02), (100_000, 1000, 192, 25, True, 0.02), (5_000, 500, 80, 20, True, 0.02),
    (200_000, 3000, 150, 18, True, 0.08), (1_000_000, 20_000, 150, 20, True, 0.02), (70_000, 1000, 101, 33, True, 0.0)])
def test_reads_on_random_text(eng, n, count, length, l, both, sub):
    rng = np.random.default_rng(n + count + l)
    t = rng.choice(ACGT, size=n)
    st, om = check(eng, t, reads_from(rng, t, count, length, sub, 0.5 if both else 0.0), l, both, max_left_frac=0.02)
    assert st["seed_windows"] > 0 and st["seed_mems"] >= 0.95 * len(om)


def test_reads_longer_than_the_planes_go_to_the_index_walk(eng):
    """Three instantiations: three plane words a strand (reads of up to 192 letters), four (up to 256: a batch whose reads
    average 193 .. 256 letters takes it, 2 x 250 bp runs) and six (up to 384: a batch that averages more than 256).  In each, a
    longer read is left to the index walk strand by strand; a batch that averages more than 384 letters does not take the seed
    path at all."""
    rng = np.random.default_rng(193)
    t = rng.choice(ACGT, size=100_000)
    st, _ = check(eng, t, reads_from(rng, t, 300, 193, 0.02), 25, True, max_left_frac=0.05)   # four words: nothing is left for its length
    st, _ = check(eng, t, reads_from(rng, t, 300, 250, 0.02), 20, True, max_left_frac=0.05)
    st, _ = check(eng, t, reads_from(rng, t, 300, 256, 0.02), 22, True, max_left_frac=0.05)
    st, _ = check(eng, t, reads_from(rng, t, 300, 257, 0.02), 22, True, max_left_frac=0.05)   # six words
    st, _ = check(eng, t, reads_from(rng, t, 200, 384, 0.02), 30, True, max_left_frac=0.05)
    st, _ = check(eng, t, reads_from(rng, t, 100, 500, 0.02), 25, True, expect_seed=False)    # beyond all three
    qs = reads_from(rng, t, 300, 193, 0.02) + reads_from(rng, t, 900, 100, 0.02) + reads_from(rng, t, 20, 400, 0.02)
    order = rng.permutation(len(qs))
    # (average 128: three words -- the first search leaves the reads of 193 and 400 letters to the index walk and tells the index
    #  that a quarter of the reads were too long for the form; the search the counters come from is the second: four words, the
    #  reads of 400 letters are left.  test_the_form_follows_the_reads_of_the_last_batch looks at each step)
    st, _ = check(eng, t, [qs[i] for i in order], 25, True, min_left=2 * 20)
    assert st["seed_strands_left"] <= 2 * 20 + 40
    qs = reads_from(rng, t, 600, 250, 0.02) + reads_from(rng, t, 300, 150, 0.02) + reads_from(rng, t, 40, 385, 0.02) + reads_from(rng, t, 30, 257, 0.02)
    order = rng.permutation(len(qs))
    st, _ = check(eng, t, [qs[i] for i in order], 20, True, min_left=2 * 70)                  # four words (average 225): 257 and 385 are left
    assert st["seed_strands_left"] <= 2 * 70 + 60
    qs = reads_from(rng, t, 600, 300, 0.02) + reads_from(rng, t, 200, 150, 0.02) + reads_from(rng, t, 40, 385, 0.02) + reads_from(rng, t, 30, 340, 0.02)
    order = rng.permutation(len(qs))
    st, _ = check(eng, t, [qs[i] for i in order], 20, True, min_left=2 * 40)                  # six words (average 273): the reads of 385 letters are left (340: 33 windows at s = 10, stay)
    assert st["seed_strands_left"] <= 2 * 40 + 60


def test_the_form_follows_the_reads_of_the_last_batch(eng):
    """The seed kernel's form (three, four or six plane words a strand) is chosen by a batch's AVERAGE read length, which a batch
    of mixed lengths defeats: reads beyond the form are left to the index walk, each at the cost of a walk.  The kernel counts
    them; when they are more than an eighth of the batch, the next batch against the same index takes the form that holds
    them, and goes back when a batch comes that did not need it.  Every answer equals the oracle's whatever the form."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(257)
    t = rng.choice(ACGT, size=120_000)
    mixed = reads_from(rng, t, 700, 150, 0.02) + reads_from(rng, t, 300, 250, 0.02)      # average 180
    order = rng.permutation(len(mixed))
    mixed = [mixed[i] for i in order]
    longer = reads_from(rng, t, 700, 150, 0.02) + reads_from(rng, t, 300, 380, 0.02)     # average 219
    short = reads_from(rng, t, 1000, 150, 0.02)
    o = po.OracleIndex(t.tobytes())
    g = eng.Index.build(t.tobytes())
    try:
        def run(qs):
            q, off = pack(qs)
            om, obc = o.match_batch(q, off, 20, True)
            gm, goff = g.find_mems(q, off, 20, True)
            assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
            for f in ("ref_pos", "query_pos", "length"):
                assert np.array_equal(gm[f], om[f]), f
            return seed_stats(eng, g, q, off, 20, True)["seed_strands_left"]
        q, off = pack(mixed)
        assert seed_stats(eng, g, q, off, 20, True)["seed_strands_left"] >= 2 * 300   # three words: the reads of 250 letters are left
        assert run(mixed) <= 100                                                       # four words from the second batch on
        assert run(short) <= 100                                                       # (still four; no read needed them: back to three)
        q, off = pack(mixed)
        assert seed_stats(eng, g, q, off, 20, True)["seed_strands_left"] >= 2 * 300   # three words again
        assert run(mixed) <= 100
        q, off = pack(longer)
        assert seed_stats(eng, g, q, off, 20, True)["seed_strands_left"] >= 2 * 300   # four words (average and hint): 380 letters are left
        assert run(longer) <= 100                                                      # six words
    finally:
        g.close()


def test_every_read_length_in_one_batch(eng):
    """Lengths 0 .. 192 (and a few beyond), several reads of each, shuffled: windows per read from none to the most a wave
    holds, reads shorter than the seed and than the minimum length, empty records."""
    rng = np.random.default_rng(7)
    t = rng.choice(ACGT, size=150_000)
    qs = []
    for length in list(range(0, 193)) + [193, 200, 255, 256, 300]:
        for _ in range(3):
            qs += reads_from(rng, t, 1, length, 0.02) if length else [np.zeros(0, dtype=np.uint8)]
    order = rng.permutation(len(qs))
    for l in (14, 20, 31):
        st, _ = check(eng, t, [qs[i] for i in order], l, True)
    # the same for the six-word instantiation: every length from 150 to 400, twice
    qs = []
    for length in range(150, 401):
        qs += reads_from(rng, t, 2, length, 0.02)
    order = rng.permutation(len(qs))
    for l in (20, 27, 45):
        st, _ = check(eng, t, [qs[i] for i in order], l, True)


def test_ends_of_the_text_and_reads_hanging_over_them(eng):
    """Reads cut from the first / last letters of the text, and reads that run past either end (their diagonals start before
    position 0 or end behind n): maximality by the text boundary, compares with letters that face no text."""
    rng = np.random.default_rng(11)
    n = 50_000
    t = rng.choice(ACGT, size=n)
    qs = []
    for k in range(60):
        a = int(rng.integers(20, 120))
        junk = rng.choice(ACGT, size=int(rng.integers(1, 60)))
        qs += [t[:a].copy(), t[n - a:].copy(), rc(t[:a]), rc(t[n - a:]),
               np.concatenate([junk, t[:a]]), np.concatenate([t[n - a:], junk]), rc(np.concatenate([junk, t[:a]])),
               mutate(rng, np.concatenate([t[n - a:], junk]), 0.03)]
    check(eng, t, qs, 20, True, max_left_frac=0.05)


def test_letters_that_are_not_acgt(eng):
    """N in reads (the strand goes to the index walk: N equals N in the reference, A.1), N runs and single N in the text (a
    compare whose units hold one reads the letter masks), lower case and IUPAC letters on both sides."""
    rng = np.random.default_rng(13)
    n = 80_000
    t = rng.choice(ACGT, size=n)
    for _ in range(6):
        p = int(rng.integers(100, n - 400))
        t[p:p + int(rng.integers(1, 200))] = ord("N")
    for _ in range(40):
        t[int(rng.integers(0, n))] = ord("N")
    qs = reads_from(rng, t, 1500, 120, 0.02)
    for i in range(0, len(qs), 9):
        qs[i] = qs[i].copy()
        qs[i][int(rng.integers(0, 120))] = ord("NRYKMnrw"[i % 8])
    for i in range(1, len(qs), 50):
        qs[i] = np.frombuffer(qs[i].tobytes().lower(), dtype=np.uint8)  # lower case is A,C,G,T all the same
    st, _ = check(eng, t, qs, 20, True)
    assert st["seed_letter_masks"] > 0 and st["seed_strands_left"] >= 2 * (len(qs) // 9)


@pytest.mark.parametrize("length,l,both", [(150, 20, True), (150, 20, False), (250, 25, True), (300, 25, True), (36, 18, True)])
def test_letters_that_are_not_acgt_in_reads_on_a_text_without_them(eng, length, l, both):
    """A text of A,C,G,T only: N (and every other letter the loader turns into N) in a READ agrees with nothing, so no MEM holds
    it -- the seed path keeps the read (windows with such a letter are not looked up, its positions disagree in every compare, on
    both strands), where a text WITH such letters sends it to the index walk.  One to many such letters per read, at the ends, in
    runs, next to matches that are barely long enough; lower case beside them."""
    rng = np.random.default_rng(71 + length)
    n = 150_000
    t = rng.choice(ACGT, size=n)
    qs = reads_from(rng, t, 1200, length, 0.02)
    for i in range(0, len(qs), 3):
        q = qs[i].copy()
        for _ in range(int(rng.integers(1, 5))):
            x = int(rng.integers(0, length))
            q[x:x + int(rng.integers(1, 4))] = ord("NRYKMnrwSBDHV"[int(rng.integers(0, 13))])
        if i % 9 == 0:
            q[0] = ord("N")
        if i % 12 == 0:
            q[-1] = ord("n")
        qs[i] = q
    for i in range(1, len(qs), 40):
        qs[i] = np.frombuffer(qs[i].tobytes().lower(), dtype=np.uint8)
    st, om = check(eng, t, qs, l, both)
    assert st["seed_strands_left"] == 0 and st["seed_left_why"][0] == 0, st["seed_left_why"]
    assert len(om) > len(qs) // 2


def test_windows_that_occur_several_times_and_ties(eng):
    """Segments of the text in 2, 3, 5, 13 and 40 exact copies and reads from them: every copy is a hit of every window (up to
    12 per bucket + the spill list; beyond, the strand is left to the index walk), equal-length MEMs with equal starts tie and
    need the suffix order (from the text behind them, or the rows of the text-ordered records when the copies go on alike),
    nested repeats give MEMs with equal starts and different lengths."""
    rng = np.random.default_rng(17)
    n = 300_000
    t = rng.choice(ACGT, size=n)
    segs = []
    for copies, length in [(2, 400), (3, 300), (5, 200), (13, 150), (40, 120)]:
        x = int(rng.integers(0, n - length))
        seg = t[x:x + length].copy()
        segs.append(seg)
        for _ in range(copies - 1):
            y = int(rng.integers(0, n - length))
            t[y:y + length] = seg
    # nested: a copy of the first segment's middle part only
    y = int(rng.integers(0, n - 100))
    t[y:y + 100] = segs[0][150:250]
    qs = reads_from(rng, t, 1500, 150, 0.02)
    for seg in segs:
        for _ in range(60):
            a = int(rng.integers(0, len(seg) - 100))
            r = mutate(rng, seg[a:a + 100 + int(rng.integers(0, min(50, len(seg) - a - 100) + 1))], 0.01)
            qs.append(rc(r) if rng.random() < 0.5 else r)
    st, _ = check(eng, t, qs, 20, True)
    # (copies that go on alike behind the match for more letters than a compare keeps: ordered by their rows, TextRec::row)
    assert st["seed_mems"] > 1500 and st["seed_left_why"][6] == 0, st["seed_left_why"]


def test_ties_are_ordered_as_the_rows_of_the_suffix_array(eng):
    """MEMs of one strand with the same start and length (a word of 22-30 letters planted 2-9 times in a random text, so the
    copies go on differently behind it): the reference lists them in row order around the interval the walk came up from
    (slamem.c:140,165) -- above it ascending, below it descending, all ascending when there is none.  Reads that hold a word
    alone (no deeper interval), reads cut from around one copy (that copy's longer match is the deeper interval, the other
    copies lie on both sides of it), with and without errors, both strands.  The seed path decides all of them itself."""
    rng = np.random.default_rng(23)
    n = 400_000
    t = rng.choice(ACGT, size=n)
    words = []
    for copies in (2, 2, 3, 3, 4, 5, 7, 9) * 6:
        wl = int(rng.integers(22, 31))
        w = rng.choice(ACGT, size=wl)
        spots = []
        for _ in range(copies):
            x = int(rng.integers(200, n - 200))
            t[x:x + wl] = w
            spots.append(x)
        words.append((w, spots))
    qs = reads_from(rng, t, 300, 150, 0.02)
    for w, spots in words:
        # the word between random letters: the interval of the word is the deepest one at its start
        r = np.concatenate([rng.choice(ACGT, size=40), w, rng.choice(ACGT, size=50)])
        qs += [r, rc(r)]
        for x in spots[:3]:  # around one copy: a longer match there, the other copies tie below it
            for lo, hi in ((60, 60), (0, 100), (100, 0), (3, 5)):
                r = t[x - lo: x + len(w) + hi].copy()
                qs += [r, rc(mutate(rng, r, 0.01))]
    st, om = check(eng, t, qs, 20, True)
    # (a word in nine copies shares its bucket with the bucket's other k-mers: more than twelve, read from the spill list)
    # (what is left: a few trips whose compares did not fit the wave's list)
    assert st["seed_left_why"][6] == 0 and st["seed_left_why"][1] == 0 and st["seed_strands_left"] <= 64, st["seed_left_why"]
    # the batch does hold ties: same strand, start and length
    obc = st["oracle_block_counts"].astype(np.int64)
    key = np.stack([np.repeat(np.arange(len(obc)), obc), om["query_pos"].astype(np.int64), om["length"].astype(np.int64)], axis=1)
    assert len(om) > len(np.unique(key, axis=0)) + 200


def test_palindromic_windows(eng):
    """A window that equals its own reverse complement (even seed length) hits BOTH strands at one text position: a compare
    for each.  Planted so that a window of the read starts exactly on the palindrome."""
    rng = np.random.default_rng(19)
    n = 300_000  # 2^18 < n <= 2^19: seed_k = 12
    t = rng.choice(ACGT, size=n)
    spots = [int(x) for x in rng.integers(1000, n - 1000, size=30)]
    for i, x in enumerate(spots):  # ten palindromes, three copies of each (a bucket holds twelve k-mers)
        half = np.random.default_rng(100 + i // 3).choice(ACGT, size=6)
        pal = np.concatenate([half, rc(half)])
        assert np.array_equal(rc(pal), pal)
        t[x:x + 12] = pal
    l = 21  # s = 10
    qs = reads_from(rng, t, 500, 150, 0.02)
    for x in spots:
        for w in (0, 1, 5):  # the palindrome at a window start (offset 10 * w) of the read
            r = t[x - 10 * w: x - 10 * w + 150].copy()
            qs += [r, rc(r)]
    st, _ = check(eng, t, qs, l, True)
    assert st["seed_k"] == 12 and st["seed_left_why"][2] >= 2 * 30 and st["seed_left_why"][6] == 0, st
    assert st["seed_strands_left"] <= 8, st


@pytest.mark.parametrize("l", [13, 14, 15, 16, 17, 18, 19, 20, 21, 25, 40])
def test_matches_that_are_barely_long_enough(eng, l):
    """Strands whose ONLY match is l .. l+3 letters long, at the strand's ends, in the middle, on both strands: the windows at
    multiples of s = l - k + 1 must catch each of them (the counterpart of test_prefilter_never_drops_...)."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(900 + l)
    n = 50_001
    t = rng.choice(ACGT, size=n)
    qs = []
    for qlen in (150, 64, 192):
        for d in range(4):
            for at in [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 30, 31, 32, 33, None]:
                m = l + d
                pos = qlen - m if at is None else at
                if pos + m > qlen:
                    continue
                q = rng.choice(ACGT, size=qlen)
                x = int(rng.integers(1, n - m - 1))
                q[pos:pos + m] = t[x:x + m]
                if pos > 0:
                    q[pos - 1] = ACGT[(int(np.searchsorted(ACGT, t[x - 1])) + 1) % 4]
                if pos + m < qlen:
                    q[pos + m] = ACGT[(int(np.searchsorted(ACGT, t[x + m])) + 1) % 4]
                qs.append(q if (len(qs) % 2 == 0) else rc(q))
    st, om = check(eng, t, qs, l, True)
    assert st["seed_k"] == 10 and len(om) >= len(qs)


@pytest.mark.parametrize("copies,left", [(10, False), (14, False), (18, False), (40, True)])
def test_buckets_with_more_than_twelve_kmers(eng, copies, left):
    """A 60-letter word in 10 .. 40 exact copies: every k-mer of it has that many text positions, all in one bucket with the same
    tag.  Up to 28 k-mers of a bucket are kept (twelve in its line, the rest in the spill list) and each is a compare; beyond,
    the strand is left to the index walk.  Reads that hold the word (with letters around it from one of the copies, from none)
    on both strands; the equal matches of the copies tie and are ordered from the text behind them."""
    rng = np.random.default_rng(31 + copies)
    n = 200_000
    t = rng.choice(ACGT, size=n)
    w = rng.choice(ACGT, size=60)
    spots = [int(x) for x in rng.choice(np.arange(300, n - 300, 200), size=copies, replace=False)]
    for x in spots:
        t[x:x + 60] = w
    qs = []
    for i in range(40):  # (among other reads: a wave's sixteen reads share one list of compares)
        qs += reads_from(rng, t, 15, 150, 0.02)
        r = np.concatenate([rng.choice(ACGT, size=int(rng.integers(0, 60))), w, rng.choice(ACGT, size=int(rng.integers(0, 30)))])
        qs.append(r if i % 2 else rc(r))
        qs += reads_from(rng, t, 15, 150, 0.02)
        x = spots[i % copies]
        r = mutate(rng, t[x - 45: x + 105], 0.01)
        qs.append(r if i % 2 else rc(r))
    st, om = check(eng, t, qs, 20, True)
    if left:
        assert st["seed_left_why"][1] > 0 and st["seed_strands_left"] >= 2 * 80, st["seed_left_why"]
    else:
        assert st["seed_left_why"][1] == 0 and st["seed_strands_left"] <= 64, st["seed_left_why"]
    assert len(om) > 40 * copies


@pytest.mark.parametrize("l", [19, 20, 23, 28])
def test_one_round_and_two_rounds_of_lookups_agree(eng, l, monkeypatch):
    """The windows of a read are looked up in two rounds when they lie close (every third / second window first; a window inside
    a run that a first-round compare measured, whose k-mer occurs once in the text, not at all): the same MEMs in the same order
    as with every window looked up in one round (SLAMEM_SEED_STEP=1), on a text with exact and diverged copies (k-mers that
    occur once next to k-mers that do not), reads with 0-6 % substitutions, every read length from l to 192."""
    rng = np.random.default_rng(41 + l)
    n = 500_000  # seed_k = 12
    t = rng.choice(ACGT, size=n)
    for copies, length, div in [(2, 500, 0.0), (3, 300, 0.03), (6, 150, 0.01), (2, 2000, 0.002), (10, 60, 0.0)]:
        x = int(rng.integers(0, n - length))
        seg = t[x:x + length].copy()
        for _ in range(copies - 1):
            y = int(rng.integers(0, n - length))
            t[y:y + length] = mutate(rng, seg, div) if div else seg
    qs = []
    for sub in (0.0, 0.02, 0.06):
        qs += reads_from(rng, t, 700, 150, sub)
    for length in range(l, 193, 3):
        qs += reads_from(rng, t, 4, length, 0.02)
    from oracle import pyoracle as po
    q, off = pack(qs)
    om, obc = po.OracleIndex(t.tobytes()).match_batch(q, off, l, True)
    g = eng.Index.build(t.tobytes())
    try:
        assert l >= g.info.seed_k + 2
        lookups = {}
        for step in ("1", "2", "3", "4", "6", ""):
            if step:
                monkeypatch.setenv("SLAMEM_SEED_STEP", step)
            else:
                monkeypatch.delenv("SLAMEM_SEED_STEP")
            gm, goff = g.find_mems(q, off, l, True)
            assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)), step
            for f in ("ref_pos", "query_pos", "length"):
                assert np.array_equal(gm[f], om[f]), (step, f)
            st = seed_stats(eng, g, q, off, l, True)
            assert st["seed_reads"] == len(qs) and st["mems"] == len(om)
            lookups[step] = st["seed_windows"]
        assert max(lookups[x] for x in ("2", "3", "4", "6")) < lookups["1"]  # (fewer lines to fetch is the point)
        assert lookups[""] <= lookups["1"]
    finally:
        g.close()


@pytest.mark.parametrize("k,l", [(17, 19), (17, 24), (18, 20), (18, 31)])
def test_seeds_of_17_and_18_letters(eng, k, l, monkeypatch):
    """Texts of 2^29 letters and more take seeds of 17 / 18 letters: keys of 34 / 36 bits, hashed with 64-bit arithmetic (a
    second form of seed_place), text positions that use all 32 bits.  SLAMEM_SEED_K forces such seeds on a small text (the
    table then has as many buckets as the 7-bit tag needs: 8.6 / 34 GB of mostly empty lines); minimum lengths from k + 2
    (three letters between two windows); repeats, palindromes and ties as in the other cases."""
    monkeypatch.setenv("SLAMEM_SEED_K", str(k))
    rng = np.random.default_rng(300 + k + l)
    n = 400_000
    t = rng.choice(ACGT, size=n)
    for copies, length, div in [(3, 400, 0.0), (5, 200, 0.02), (12, 80, 0.0)]:
        x = int(rng.integers(0, n - length))
        seg = t[x:x + length].copy()
        for _ in range(copies - 1):
            y = int(rng.integers(0, n - length))
            t[y:y + length] = mutate(rng, seg, div) if div else seg
    for i in range(20):
        half = rng.choice(ACGT, size=k // 2 + 2)
        pal = np.concatenate([half, rc(half)])
        y = int(rng.integers(0, n - 60))
        t[y:y + len(pal)] = pal
    qs = reads_from(rng, t, 1500, 150, 0.02) + reads_from(rng, t, 300, 60, 0.0) + reads_from(rng, t, 100, 192, 0.05)
    st, om = check(eng, t, qs, l, True)
    assert st["seed_k"] == k and st["seed_strands_left"] <= 64, st["seed_left_why"]
    assert len(om) > 1500


def test_windows_of_one_offset_array_and_waves_that_do_not_fit_the_staging_buffer(eng):
    """slamem_stream_submit hands the search windows of ONE offsets array (a batch's offsets start anywhere, its letters lie
    behind the same base pointer), and a wave whose 21 reads are longer together than the staging buffer reads its letters from
    global memory: the last word it may touch is the batch's last, wherever the batch starts.  Reads of 30-384 letters mixed so
    that many waves are of that kind, four batches, both the letters and the bit-planes form of the upload."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(83)
    n = 200_000
    t = rng.choice(ACGT, size=n)
    qs = []
    for i in range(1200):
        qs += reads_from(rng, t, 1, int(rng.integers(30, 385)) if i % 3 else 150, 0.02)
    q, off = pack(qs)
    chars = np.frombuffer(q, dtype=np.uint8) if not isinstance(q, np.ndarray) else q
    o = po.OracleIndex(t.tobytes())
    idx = eng.Index.build(t.tobytes())
    per = 300
    st = eng.Stream(idx, 3, 1 << 18, per, True)
    keep = []
    try:
        for packed in (False, True):
            for b in range(len(qs) // per):
                w = off[b * per: (b + 1) * per + 1]
                if packed:
                    units = int(((np.diff(w) + 63) // 64).sum())
                    pl = eng.PinnedBuffer(units * 16 + 64)
                    ot = np.zeros(units + 1, dtype=np.uint64)
                    assert eng.pack_reads(chars, w, pl.array, ot, threads=2) == units
                    keep.append(pl)
                    st.submit_packed(pl.array, None, w, 20, units=units)
                else:
                    st.submit(chars, w, 20)
                m, boff, _ = st.next()
                om, obc = o.match_batch(chars[int(w[0]): int(w[-1])], w - w[0], 20, True)
                assert np.array_equal(np.diff(boff.astype(np.int64)), obc.astype(np.int64)), (packed, b)
                for f in ("ref_pos", "query_pos", "length"):
                    assert np.array_equal(m[f], om[f]), (packed, b, f)
    finally:
        st.close()
        for pl in keep:
            pl.close()
        idx.close()


def test_a_record_longer_than_a_slice_among_the_reads(eng):
    """slamem_find_mems_device takes the seed path without asking the device for the longest record first; K8s notices a record
    that needs slices (more than 4096 letters) and the call starts again with the item tables: the answer is the oracle's, for the
    long record (cut into slices, scanned by the index walk) and for the reads beside it."""
    rng = np.random.default_rng(97)
    n = 150_000
    t = rng.choice(ACGT, size=n)
    qs = reads_from(rng, t, 900, 150, 0.02)
    qs.insert(400, mutate(rng, t[20_000:29_000], 0.01))
    qs.insert(700, rc(mutate(rng, t[90_000:94_097], 0.02)))  # 4097 letters: one more than a slice
    st, om = check(eng, t, qs, 20, True, expect_seed=False)
    assert len(om) > 1500
    st, om = check(eng, t, [q for q in qs if len(q) <= 4096], 20, True)  # (without them: the seed path, no second start)
    assert st["seed_strands_left"] == 0


def test_spill_list_runs_full(eng):
    """The spill list holds n/16 + 64 entries: 40 words of 60 letters in 14 copies each ask for more (some 1,900 buckets of 14+
    k-mers, four entries each).  The buckets that still fit are read from the list, the others count as "more than fit" and
    their strands go to the index walk: same MEMs either way."""
    rng = np.random.default_rng(59)
    n = 100_000
    t = rng.choice(ACGT, size=n)
    words = [rng.choice(ACGT, size=60) for _ in range(40)]
    spots = rng.choice(np.arange(100, n - 100, 70), size=40 * 14, replace=False)
    for i, x in enumerate(spots):
        t[int(x):int(x) + 60] = words[i // 14]
    qs = reads_from(rng, t, 600, 150, 0.02)
    for i, w in enumerate(words):
        r = np.concatenate([rng.choice(ACGT, size=30), w, rng.choice(ACGT, size=40)])
        qs.insert(15 * i, r if i % 2 else rc(r))
    st, om = check(eng, t, qs, 20, True)
    assert st["seed_left_why"][1] > 0 and 0 < st["seed_strands_left"] < st["items"], st["seed_left_why"]
    assert len(om) > 40 * 14


def test_satellite_fills_its_buckets(eng):
    """A tandem array (one unit 2,000 times) puts thousands of positions into the buckets of its k-mers (count 13 = more than
    fit): reads from it, and reads that merely share one window with it, are left to the index walk; the others are not."""
    rng = np.random.default_rng(23)
    n = 400_000
    t = rng.choice(ACGT, size=n)
    unit = rng.choice(ACGT, size=37)
    t[50_000:50_000 + 37 * 2000] = np.tile(unit, 2000)
    qs = reads_from(rng, t, 1200, 150, 0.02)
    chim = []
    for _ in range(50):  # 20 letters of the array inside an ordinary read
        x = int(rng.integers(200_000, n - 200))
        r = t[x:x + 150].copy()
        a = int(rng.integers(0, 130))
        r[a:a + 20] = np.tile(unit, 2)[3:23]
        chim.append(r)
    st, _ = check(eng, t, qs + chim, 30, True)
    assert 0 < st["seed_strands_left"] < st["items"]


def test_forward_only_ignores_the_other_strand(eng, monkeypatch):
    """Without -b a hit in the other orientation is never a MEM.  With one round of lookups such hits are dropped before any
    compare; with two, the first round still compares them -- not to report anything, but to learn which windows of the second
    round have their only occurrence on that strand and need no lookup (half the reads of a sequencing run come from it):
    fewer windows are looked up than in one round, and the answer is the same (nothing, here: every read comes from the
    other strand)."""
    rng = np.random.default_rng(29)
    t = rng.choice(ACGT, size=120_000)
    qs = reads_from(rng, t, 1000, 150, 0.02, rc_share=1.0)  # every read comes from the other strand
    monkeypatch.setenv("SLAMEM_SEED_STEP", "1")
    one, om = check(eng, t, qs, 20, False)
    assert len(om) < 20 and one["seed_compares"] < 3000  # (chance forward hits only; the true ones are all on the other strand)
    monkeypatch.delenv("SLAMEM_SEED_STEP")
    two, om2 = check(eng, t, qs, 20, False)
    assert len(om2) == len(om) and two["seed_mems"] == one["seed_mems"]
    assert two["seed_windows"] < 0.7 * one["seed_windows"], (one["seed_windows"], two["seed_windows"])


def test_full_size_headline_both_paths_agree_in_order(eng):
    """BASELINE.json configs[2] at full size through both paths: the seed path's 24,216,704 MEMs are the index walk's, row for
    row in the same order (the walk's order is the oracle's on every sample checked; its set is the REAL reference's digest in
    test_config3_known_answer_full_size), and nearly every strand is answered by K8s itself."""
    import torch
    from conftest import search_path
    n, nreads, L = 100_000_000, 10_000_000, 150
    ref = eng.synth_reference(n, 42, "cuda:0")
    reads = eng.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)
    offsets = torch.arange(nreads + 1, dtype=torch.int64, device="cuda:0") * L
    idx = eng.Index.build(ref, "cuda:0")
    m = idx.matcher(nreads, True, 4 * nreads, nreads * L)
    with search_path("seed"):
        total = m.run(reads, offsets, 20)
    assert total == 24_216_704
    seed_mems = m.mems[:total].clone()
    seed_off = m.block_offsets.clone()
    with search_path("walk"):
        assert m.run(reads, offsets, 20) == total
    assert torch.equal(seed_off, m.block_offsets) and torch.equal(seed_mems, m.mems[:total])
    with search_path("seed"):
        st = eng.search_stats(m, reads, offsets, 20)
    assert st["seed_reads"] == nreads and 8 * nreads <= st["seed_windows"] <= 27 * nreads and st["seed_strands_left"] < 1000, st["seed_left_why"]
    assert st["survivors"] <= st["seed_strands_left"]
    idx.close()
