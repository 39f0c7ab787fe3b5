"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from slamem_amd import engine
    return engine


def rand_text(rng, n, alpha, repeats=0, max_rep=300, nrun=0):
    t = rng.choice(np.frombuffer(alpha.encode(), dtype=np.uint8), size=n)
    for _ in range(repeats):
        L = int(rng.integers(10, max_rep))
        a, b = int(rng.integers(0, n - L)), int(rng.integers(0, n - L))
        t[b:b + L] = t[a:a + L].copy()
    if nrun:
        a = int(rng.integers(0, n - nrun))
        t[a:a + nrun] = ord("N")
    return t.tobytes()


CASES = [("ACGT", 1000, 0, 0), ("ACGT", 5000, 4, 0), ("AC", 2000, 3, 0), ("ACGTN", 3000, 2, 150), ("A", 300, 0, 0),
         ("ACGT", 1, 0, 0), ("ACGT", 2, 0, 0), ("ACGT", 127, 0, 0), ("ACGT", 128, 0, 0), ("ACGT", 4095, 2, 0),
         ("ACGT", 100000, 20, 0)]


@pytest.mark.parametrize("alpha,n,repeats,nrun", CASES)
def test_index_arrays_match_oracle(eng, alpha, n, repeats, nrun):
    """SA / BWT / LCP / PSV / NSV are uniquely defined by the text: bit-exact equality."""
    from oracle import pyoracle as po
    from slamem_amd import capi
    rng = np.random.default_rng(n * 7 + repeats)
    text = rand_text(rng, n, alpha, repeats if n > 700 else 0, nrun=nrun if n > 700 else 0)
    o = po.OracleIndex(text)
    g = eng.Index.build(text)
    assert g.bwt_size() == n + 1
    assert np.array_equal(g.download(capi.ARRAY_SA).astype(np.int64), o.sa)
    assert np.array_equal(g.download(capi.ARRAY_BWT), o.bwt)
    lcp = g.download(capi.ARRAY_LCP).astype(np.int64)
    assert np.array_equal(lcp, o.lcp)
    psv, nsv = g.download(capi.ARRAY_PSV).astype(np.int64), g.download(capi.ARRAY_NSV).astype(np.int64)
    assert np.array_equal(psv[1:n + 1], o.psv[1:n + 1])
    assert np.array_equal(nsv[1:n + 1], o.nsv[1:n + 1])
    g.close()


def make_queries(rng, text, nq, alpha, maxlen=300):
    t = np.frombuffer(text, dtype=np.uint8)
    n = len(t)
    qs = []
    for k in range(nq):
        L = int(rng.integers(1, min(maxlen, n) + 1))
        if k % 5 != 4:
            a = int(rng.integers(0, n - L + 1))
            q = t[a:a + L].copy()
            mut = rng.random(L) < 0.03
            q[mut] = rng.choice(np.frombuffer(alpha.encode(), dtype=np.uint8), size=int(mut.sum()))
        else:
            q = rng.choice(np.frombuffer(alpha.encode(), dtype=np.uint8), size=L)
        qs.append(q.tobytes())
    return qs


def pack(qs):
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(q) for q in qs])
    return np.frombuffer(b"".join(qs), dtype=np.uint8), off


@pytest.mark.parametrize("alpha,n,repeats,nrun,l,both", [
    ("ACGT", 3000, 3, 0, 20, False), ("ACGT", 3000, 3, 0, 20, True), ("AC", 1500, 2, 0, 10, True),
    ("ACGTN", 2500, 2, 120, 8, True), ("ACGT", 800, 1, 0, 1, False), ("ACGT", 900, 1, 0, 3, True),
    ("ACGT", 20000, 10, 0, 15, True), ("A", 200, 0, 0, 5, True), ("ACGTN", 4000, 3, 400, 2, False)])
def test_find_mems_matches_oracle_in_order(eng, alpha, n, repeats, nrun, l, both):
    """Same MEMs, same blocks, same emission order as the restated GetMatches (slamem.c:114-199)."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(n + l)
    text = rand_text(rng, n, alpha, repeats, nrun=nrun)
    qs = make_queries(rng, text, 40, alpha) + [b"", b"N" * 25, text[:30], text[-30:]]
    q, off = pack(qs)
    o = po.OracleIndex(text)
    om, obc = o.match_batch(q, off, l, both)
    g = eng.Index.build(text)
    from conftest import search_path
    for path in ("seed", "walk"):  # K8s + K8 for what it leaves (where the batch qualifies), and K8a + K8 for everything
        with search_path(path):
            gm, goff = g.find_mems(q, off, l, both)
        assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)), path
        assert len(gm) == len(om), path
        for f in ("ref_pos", "query_pos", "length"):
            assert np.array_equal(gm[f], om[f]), (f, path)
    g.close()


@pytest.mark.parametrize("alpha,n,repeats,l,both", [
    ("ACGT", 3000, 8, 8, True), ("AC", 1500, 4, 12, False), ("ACG", 900, 3, 5, True), ("ACGT", 20000, 30, 15, True),
    ("ACGTN", 2500, 4, 6, True), ("A", 300, 0, 4, True)])
def test_find_mams_matches_oracle_in_order(eng, alpha, n, repeats, l, both):
    """-mam (slamem_find_mams_device): same triples in the same order as the restated scan with matchType 1
    (slamem.c:131), stale fall-back interval included; the restatement itself is pinned by the mam_* golden files."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(7 * n + l)
    text = rand_text(rng, n, alpha, repeats, nrun=60 if "N" in alpha else 0)
    qs = make_queries(rng, text, 40, alpha) + [b"", b"N" * 25, text[:30], text[-30:], text[: min(n, 6000)]]
    q, off = pack(qs)
    o = po.OracleIndex(text)
    om, obc = o.match_batch(q, off, l, both, mam=True)
    mm, _ = o.match_batch(q, off, l, both)
    g = eng.Index.build(text)
    gm, goff = g.find_mems(q, off, l, both, mam=True)
    assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
    assert len(gm) == len(om)
    for f in ("ref_pos", "query_pos", "length"):
        assert np.array_equal(gm[f], om[f]), f
    if repeats:
        assert len(om) != len(mm)  # the case does exercise the mode
    g.close()


def test_fine_grained_ops_match_oracle(eng):
    """FMI_FollowLetter / GetEnclosingLCPInterval / FMI_PositionInText / FMI_GetCharAtBWTPos, batched."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(5)
    text = rand_text(rng, 5000, "ACGTN", 4, nrun=100)
    n = len(text)
    o = po.OracleIndex(text)
    g = eng.Index.build(text)
    tops = rng.integers(0, n + 1, size=4000)
    bots = np.minimum(n, tops + (rng.integers(0, 50, size=4000) * (rng.random(4000) < 0.7)))
    tops[:10], bots[:10] = 0, n
    letters = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=4000))
    s, t, b = g.follow_letter(letters, tops, bots)
    d, pt, pb = g.enclosing_interval(tops, bots)
    for i in range(4000):
        es, et, eb = o.follow_letter(chr(letters[i]), int(tops[i]), int(bots[i]))
        assert (int(s[i]), int(t[i]), int(b[i])) == (es, et, eb), i
        ed, ept, epb = o.enclosing_interval(int(tops[i]), int(bots[i]))
        assert (int(d[i]), int(pt[i]), int(pb[i])) == (ed, ept, epb), i
    rows = np.arange(0, n + 1)
    assert np.array_equal(g.position_in_text(rows).astype(np.int64), o.sa)
    assert g.char_at_bwt_pos(rows) == bytes(b"$NACGT"[c] for c in o.bwt)
    g.close()


def test_arena_export_attach_save_load(eng, tmp_path):
    import torch
    rng = np.random.default_rng(9)
    text = rand_text(rng, 3000, "ACGT", 2)
    qs = make_queries(rng, text, 20, "ACGT")
    q, off = pack(qs)
    g = eng.Index.build(text)
    ref, refoff = g.find_mems(q, off, 12, True)
    arena = g.export_arena()
    peer = torch.empty_like(arena)
    peer.copy_(arena)  # stands in for torch.distributed.broadcast
    g2 = eng.Index.attach(peer)
    a, aoff = g2.find_mems(q, off, 12, True)
    assert np.array_equal(a, ref) and np.array_equal(aoff, refoff)
    p = str(tmp_path / "idx.slamem")
    g.save(p)
    g3 = eng.Index.load(p)
    a, aoff = g3.find_mems(q, off, 12, True)
    assert np.array_equal(a, ref) and np.array_equal(aoff, refoff)
    with pytest.raises(Exception):
        eng.Index.attach(torch.zeros(8192, dtype=torch.uint8, device="cuda:0"))


def test_config2_known_answer_full_size(eng):
    """BASELINE.json configs[1] at full size: 100 Mbp reference, 1 M x 150 bp reads, forward, -l 20 (and the index through a
    file: an arena of 8.2 GB saved, loaded, searched again).
    Known answers recorded from the REAL reference in SURVEY.md Appendix C.3 (2,412,288 MEMs, sum of lengths
    133,301,375, sha256 prefix 9d583ab9312e1698 of the numerically sorted 'ref<TAB>query<TAB>len' lines)."""
    import hashlib
    import torch
    n, nreads, L = 100_000_000, 1_000_000, 150
    ref = eng.synth_reference(n, 42, "cuda:0")
    reads = eng.synth_reads(ref, 0, nreads, L, 0.02, 42, 0)
    offsets = torch.arange(nreads + 1, dtype=torch.int64, device="cuda:0") * L
    idx = eng.Index.build(ref, "cuda:0")
    # structure statistics the reference printed for this text
    st = idx.sampled_lcp_stats()
    assert st["num_samples"] == 85_214_690 and st["num_oversized_lcp"] == 1
    assert st["num_oversized_links"] == 1_573_293 and st["max_lcp"] == 26
    assert st["sum_lcp"] // (n + 1) == 12 and round(st["sum_link_distance"] / st["num_samples"], 2) == 43.93
    m = idx.matcher(nreads, False, 4 * nreads, nreads * L)
    total = m.run(reads, offsets, 20)
    assert total == 2_412_288
    mems = m.mems[:total].cpu().numpy().view(np.uint32).astype(np.int64)
    assert int(mems[:, 2].sum()) == 133_301_375
    assert int(mems[:, 2].max()) == 150
    # size-independent properties: every MEM is a real match inside the read, maximal on both sides
    ref_h = ref.cpu().numpy()
    reads_h = reads[: nreads * L].cpu().numpy().reshape(nreads, L)
    boff = m.block_offsets[: nreads + 1].cpu().numpy()
    rid = np.repeat(np.arange(nreads), np.diff(boff))
    r, q, ln = mems[:, 0], mems[:, 1], mems[:, 2]
    assert (q + ln <= L).all() and (r + ln <= n).all() and (ln >= 20).all()
    sel = np.random.default_rng(0).choice(total, size=20000, replace=False)
    for i in sel:
        a, b, c, d = int(r[i]), int(q[i]), int(ln[i]), int(rid[i])
        assert (ref_h[a:a + c] == reads_h[d, b:b + c]).all()
        assert a == 0 or b == 0 or ref_h[a - 1] != reads_h[d, b - 1]
        assert a + c == n or b + c == L or ref_h[a + c] != reads_h[d, b + c]
    order = np.lexsort((ln, q, r))
    lines = b"".join(b"%d\t%d\t%d\n" % (r[i] + 1, q[i] + 1, ln[i]) for i in order)
    assert hashlib.sha256(lines).hexdigest().startswith("9d583ab9312e1698")
    # save -> load of an arena beyond 4 GiB (8.2 GB: 64-bit offsets, sections in every size class), then the same search on the
    # loaded index -- both paths, since the file must carry the seed sections and the index walk's alike
    import os
    import tempfile
    from conftest import search_path
    assert idx.info.arena_bytes > (4 << 30)
    first = m.mems[:total].clone()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        path = os.path.join(td, "index.slamem")
        idx.save(path)
        assert os.path.getsize(path) == idx.info.arena_bytes
        idx.close()
        idx2 = eng.Index.load(path, "cuda:0")
    assert idx2.info.arena_bytes > (4 << 30) and idx2.info.seed_k == 16
    m2 = idx2.matcher(nreads, False, 4 * nreads, nreads * L)
    for path_name in ("seed", "walk"):
        with search_path(path_name):
            assert m2.run(reads, offsets, 20) == total
        assert torch.equal(m2.mems[:total], first) and torch.equal(m2.block_offsets, m.block_offsets), path_name
    idx2.close()


def test_config3_known_answer_full_size(eng):
    """BASELINE.json configs[2] -- the workload bench.py times -- at full size: 100 Mbp reference, 10 M x 150 bp reads,
    half of them reverse-complemented, -b -l 20.  The known answer (tests/golden/config3_known_answer.json) was
    recorded from a run of the REAL reference on the same FASTA files (tools/gen_synth.py ... 42 50; ~35 minutes of
    one core in the build container) and digested with tools/mems_digest.py: MEM count, sum / max of the lengths and
    sha256 over the sorted (strand block, ref, query, length) rows."""
    import json
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from mems_digest import digest_rows
    known = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_known_answer.json")))
    n, nreads, L = 100_000_000, 10_000_000, 150
    ref = eng.synth_reference(n, 42, "cuda:0")
    reads = eng.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)
    offsets = torch.arange(nreads + 1, dtype=torch.int64, device="cuda:0") * L
    idx = eng.Index.build(ref, "cuda:0")
    m = idx.matcher(nreads, True, 4 * nreads, nreads * L)
    total = m.run(reads, offsets, 20)
    assert total == known["mems"]
    mems = m.mems[:total].cpu().numpy().view(np.uint32)
    boff = m.block_offsets[: 2 * nreads + 1].cpu().numpy()
    rows = np.empty((total, 4), dtype=np.uint32)
    rows[:, 0] = np.repeat(np.arange(2 * nreads, dtype=np.uint32), np.diff(boff))
    rows[:, 1] = mems[:, 0] + 1
    rows[:, 2] = mems[:, 1] + 1
    rows[:, 3] = mems[:, 2]
    got = digest_rows(rows)
    assert got == {k: known[k] for k in ("mems", "sum_len", "max_len", "sha256")}
    idx.close()


def test_capacity_error_reports_the_need_and_retry_succeeds(eng):
    """SLAMEM_ERR_CAPACITY contract: the call fails loudly, total_out holds the room to ask for, a retry works."""
    import torch
    from slamem_amd import capi
    rng = np.random.default_rng(21)
    text = rand_text(rng, 20000, "ACGT", 30, max_rep=400)
    qs = make_queries(rng, text, 200, "ACGT")
    q, off = pack(qs)
    idx = eng.Index.build(text)
    qd = torch.zeros((len(q) + 15) // 8 * 8, dtype=torch.uint8, device="cuda:0")
    qd[: len(q)] = torch.from_numpy(q.copy()).to("cuda:0")
    od = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    small = idx.matcher(len(qs), True, 3, int(off[-1]))
    with pytest.raises(capi.SlamemError) as ei:
        small.run(qd, od, 5)
    assert ei.value.code == capi.SLAMEM_ERR_CAPACITY and small.last_total > 3
    big = idx.matcher(len(qs), True, small.last_total, int(off[-1]))
    total = big.run(qd, od, 5)
    ref, _ = idx.find_mems(q, off, 5, True)
    assert total == len(ref) <= small.last_total
    got = big.mems[:total].cpu().numpy().view(np.uint32)
    assert np.array_equal(got[:, 0], ref["ref_pos"]) and np.array_equal(got[:, 2], ref["length"])
    # degenerate batches
    e, eo = idx.find_mems(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64), 20, True)
    assert len(e) == 0 and list(eo) == [0]
    idx.close()


def test_stream_host_to_host_matches_device_path(eng):
    """slamem_stream_* (pinned host buffers, pipelined slots; replaces the query loop at slamem.c:90-207): every batch
    equals slamem_find_mems_device on the same records and the oracle, in order; offsets windows with a non-zero base;
    ragged last batch; the busy / empty errors; MAM mode through the same entry point."""
    from oracle import pyoracle as po
    from slamem_amd import capi
    rng = np.random.default_rng(33)
    text = rand_text(rng, 60000, "ACGT", 40, max_rep=300)
    qs = make_queries(rng, text, 1000, "ACGTN")
    q, off = pack(qs)
    idx = eng.Index.build(text)
    o = po.OracleIndex(bytes(text))
    nq = len(qs)
    per = 137
    nbatches = (nq + per - 1) // per
    buf = eng.PinnedBuffer(len(q) + 64)
    buf.array[: len(q)] = q
    for mam in (False, True):
        st = eng.Stream(idx, 3, int(max(np.diff(off[::1]).max() * per, 1 << 16)), per, True, mam=mam)
        with pytest.raises(capi.SlamemError):
            st.next()  # nothing pending
        wins = [off[b * per: min(nq, (b + 1) * per) + 1] for b in range(nbatches)]
        st.submit(buf.array, wins[0], 11)
        st.submit(buf.array, wins[1], 11)
        for b in range(nbatches):
            m, boff, _ = st.next()
            if b + 2 < nbatches:
                st.submit(buf.array, wins[b + 2], 11)
                if b + 3 < nbatches:
                    with pytest.raises(capi.SlamemError):  # slots - 1 in flight beside the lent result: full
                        st.submit(buf.array, wins[b + 3], 11)
            w = wins[b]
            sub = q[int(w[0]): int(w[-1])]
            om, obc = o.match_batch(sub, w - w[0], 11, True, mam=mam)
            assert np.array_equal(np.diff(boff.astype(np.int64)), obc.astype(np.int64))
            for f in ("ref_pos", "query_pos", "length"):
                assert np.array_equal(m[f], om[f]), (mam, b, f)
        st.close()
    buf.close()
    idx.close()


def test_stream_packed_reads_match_the_letters(eng):
    """slamem_stream_submit_packed: the same batches handed in as bit-planes (slamem_pack_reads; letters that are not A,C,G,T in
    the third plane) give what the letters give -- the oracle's MEMs in order -- for reads of every length incl. empty ones,
    long records (cut into slices), N, windows of the offsets with a non-zero base, and with the `other` plane left out when
    the batch has no such letter."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(44)
    text = rand_text(rng, 80000, "ACGT", 20, max_rep=300)
    qs = make_queries(rng, text, 700, "ACGTN") + [np.frombuffer(bytes(text[5000:14000]), dtype=np.uint8), np.zeros(0, dtype=np.uint8)]
    qs += [np.frombuffer(bytes(text[a:a + 150]), dtype=np.uint8) for a in rng.integers(0, 70000, 600)]
    q, off = pack(qs)
    idx = eng.Index.build(text)
    o = po.OracleIndex(bytes(text))
    nq = len(qs)
    per = 211
    wins = [off[b * per: min(nq, (b + 1) * per) + 1] for b in range((nq + per - 1) // per)]
    chars = np.frombuffer(q, dtype=np.uint8) if not isinstance(q, np.ndarray) else q
    st = eng.Stream(idx, 3, 1 << 17, per, True)
    keep = []
    for b, w in enumerate(wins):
        units = int(((np.diff(w) + 63) // 64).sum())
        pl = eng.PinnedBuffer(units * 16 + 64)
        ot = np.zeros(units + 1, dtype=np.uint64)
        assert eng.pack_reads(chars, w, pl.array, ot, threads=3) == units
        has_other = bool(ot.any())
        keep.append(pl)
        st.submit_packed(pl.array, ot if has_other or b % 2 == 0 else None, w, 13, units=units if b % 3 else 0)
        m, boff, _ = st.next()
        om, obc = o.match_batch(chars[int(w[0]): int(w[-1])], w - w[0], 13, True)
        assert np.array_equal(np.diff(boff.astype(np.int64)), obc.astype(np.int64)), b
        for f in ("ref_pos", "query_pos", "length"):
            assert np.array_equal(m[f], om[f]), (b, f)
    st.close()
    for pl in keep:
        pl.close()
    idx.close()


def test_a_slice_with_2_to_the_28_mems_is_an_error_not_wrong_output(eng):
    """The overflow records carry the MEM's ordinal within its work item (since round 4: within its job, when K8 queues them) in 28 bits.  A text of 1.4 M copies of one 20-mer,
    each behind a different letter than the query's, and a 4095-letter query of 195 copies: 273 M MEMs from ONE slice.
    The call must fail with SLAMEM_ERR_ARG and a message that names the limit (not SLAMEM_ERR_CAPACITY, which callers
    answer by asking again)."""
    from slamem_amd import capi
    rng = np.random.default_rng(28)
    motif = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=20)
    copies = 1_400_000
    text = np.empty((copies, 21), dtype=np.uint8)
    text[:, 0] = rng.choice(np.frombuffer(b"ACT", dtype=np.uint8), size=copies)
    text[:, 1:] = motif
    query = np.tile(np.concatenate([np.frombuffer(b"G", dtype=np.uint8), motif]), 195)
    assert query.shape[0] == 4095
    idx = eng.Index.build(text.reshape(-1))
    off = np.array([0, query.shape[0]], dtype=np.uint64)
    import os
    os.environ["SLAMEM_ENUM_DEFER"] = "0"  # the jobs in the waves: one MEM numbering per strand, 28 bits of it in a record
    try:
        with pytest.raises(capi.SlamemError) as e:
            idx.find_mems(query, off, 20, False)
    finally:
        os.environ.pop("SLAMEM_ENUM_DEFER", None)
    assert e.value.code == capi.SLAMEM_ERR_ARG and "2^28" in str(e.value)
    # Round 4: with the jobs in the queue (the default on such a text) a record's number counts within its job and K9 adds what
    # comes before in 64 bits: the same call answers -- every one of the 195 copies in the query against every copy in the text
    import torch
    m = idx.matcher(1, False, 280_000_000, query.shape[0])
    qd = torch.zeros(4096 + 16, dtype=torch.uint8, device="cuda:0")
    qd[:4095] = torch.from_numpy(query).to("cuda:0")
    total = m.run(qd, torch.from_numpy(off.view(np.int64)).to("cuda:0"), 20)
    assert total == 195 * copies
    got = m.mems[:total]
    assert int(got[:, 2].min()) == 20 and int(got[:, 2].max()) == 20 and torch.equal(got[::copies, 1].cpu(), torch.arange(194, -1, -1, dtype=torch.int32) * 21 + 1)
    del m, got
    # one letter fewer per copy of the query's motif: nothing reaches min_len, and the same index answers normally
    m, boff = idx.find_mems(np.tile(np.concatenate([np.frombuffer(b"G", dtype=np.uint8), motif[:19]]), 100), np.array([0, 2000], dtype=np.uint64), 20, False)
    assert len(m) == 0 and list(boff) == [0, 0]
    idx.close()


@pytest.mark.parametrize("min_len,mam", [(8, False), (9, False), (13, False), (15, False), (16, False), (17, False), (21, False),
                                         (40, False), (151, False), (9, True), (13, True), (15, True), (21, True), (151, True)])
def test_minimum_length_thresholds_of_the_search_path(eng, min_len, mam):
    """One 3 Mbp text (presence filter k = 15, jump table K = 9) and 20,000 (1,000) reads of 150 letters, both strands, for the
    minimum lengths at which the path changes shape: below / at K (no jump), below / at the filter's k (no prefilter,
    one level, two, three levels), a common value, a large one, and one above the read length (nothing can match).
    Equal to the oracle in order every time; -mam (its own kernel, behind the same filter) at five of them."""
    from oracle import pyoracle as po
    from slamem_amd import synth
    n, nreads, L = 3_000_000, (20_000 if min_len >= 13 else 1_000), 150  # (short matches are many: fewer reads for them)
    ref = synth.make_reference(n, seed=31)
    synth.plant_repeats(ref, 31)
    reads = synth.make_reads(ref, 0, nreads, L, 0.03, seed=31, rc_percent=50).reshape(-1)
    off = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(L)
    idx = eng.Index.build(ref)
    o = po.OracleIndex(ref.tobytes())
    om, obc = o.match_batch(reads, off, min_len, True, mam=mam)
    gm, goff = idx.find_mems(reads, off, min_len, True, mam=mam)
    assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
    for f in ("ref_pos", "query_pos", "length"):
        assert np.array_equal(gm[f], om[f]), f
    if min_len > L:
        assert len(gm) == 0
    idx.close()


@pytest.mark.parametrize("read_len,min_len", [(319, 17), (320, 17), (321, 17), (500, 17), (640, 16), (641, 21), (1000, 17),
                                              (1280, 17), (1300, 17), (2500, 40)])
def test_prefilter_parts_for_reads_longer_than_one_span(eng, read_len, min_len):
    """K8a packs the strands of a block into LDS in one, two or four parts (128 reads of up to 320 letters, 64 of up to
    640, 32 of up to 1280) and falls back to its letter loop beyond; every shape equals the oracle in order.  3 Mbp text
    (filter k = 15): min_len 17 is the three-level cascade, 16 two levels, 21 and 40 the single-level path."""
    from oracle import pyoracle as po
    from slamem_amd import synth
    n, nreads = 3_000_000, 700
    ref = synth.make_reference(n, seed=77)
    reads = synth.make_reads(ref, 0, nreads, read_len, 0.03, seed=read_len, rc_percent=50).reshape(-1)
    # a few records of other lengths in between, so that spans are ragged and one wave's span can exceed the others'
    rng = np.random.default_rng(read_len)
    lens = np.full(nreads, read_len, dtype=np.int64)
    lens[rng.integers(0, nreads, size=40)] = rng.integers(1, read_len + 1, size=40)
    lens[rng.integers(0, nreads, size=5)] = 0
    pieces, pos = [], 0
    for ln in lens:  # cut the generated letters into records of these lengths (the tail of a shortened read is dropped)
        pieces.append(reads[pos:pos + ln])
        pos += read_len
    q = np.concatenate(pieces)
    q[rng.integers(0, len(q), size=30)] = ord("N")
    off = np.zeros(nreads + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    idx = eng.Index.build(ref)
    o = po.OracleIndex(ref.tobytes())
    om, obc = o.match_batch(q, off, min_len, True)
    gm, goff = idx.find_mems(q, off, min_len, True)
    assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
    for f in ("ref_pos", "query_pos", "length"):
        assert np.array_equal(gm[f], om[f]), f
    assert len(gm) > nreads  # (the reads do match)
    idx.close()


def test_stream_large_batches_from_ordinary_and_page_locked_memory(eng):
    """Batches of 40 MB and 72 MB whose characters live in ordinary (pageable) memory, the same batches from page-locked
    memory, and slamem_find_mems_device on the same reads must agree MEM for MEM.  (Staging pageable batches through
    page-locked buffers inside the stream was measured against the runtime's own path: slower, not adopted.)"""
    from slamem_amd import synth
    n, nreads, L = 2_000_000, 750_000, 150
    ref = synth.make_reference(n, seed=9)
    reads = synth.make_reads(ref, 0, nreads, L, 0.02, seed=9, rc_percent=50).reshape(-1)
    off = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(L)
    idx = eng.Index.build(ref)
    want, want_off = idx.find_mems(reads, off, 20, True)
    cuts = [0, 270_000, 750_000]  # 40.5 MB and 72 MB
    pinned = eng.PinnedBuffer(len(reads) + 64)
    pinned.array[: len(reads)] = reads
    for src in (np.array(reads), pinned.array):
        st = eng.Stream(idx, 3, 1 << 20, 1024, True)
        got, got_counts = [], []
        for b in range(len(cuts) - 1):
            st.submit(src, off[cuts[b]: cuts[b + 1] + 1], 20)
        for b in range(len(cuts) - 1):
            m, boff, _ = st.next()
            got.append(m.copy())
            got_counts.append(np.diff(boff.astype(np.int64)))
        st.close()
        gm = np.concatenate(got)
        assert np.array_equal(np.concatenate(got_counts), np.diff(want_off.astype(np.int64)))
        for f in ("ref_pos", "query_pos", "length"):
            assert np.array_equal(gm[f], want[f]), f
    pinned.close()
    idx.close()


def test_stream_slots_grow_with_the_batches(eng):
    """slamem_stream_create's sizes are a reservation, not a limit: a stream set up for 4 records of 1 kB takes batches that
    grow from 3 to 600 records (and a 30 kB record among 100-letter reads), from pageable memory, with empty records and an
    empty batch in between; every batch equals the oracle in order."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(808)
    text = rand_text(rng, 80000, "ACGT", 30, max_rep=400)
    t = np.frombuffer(text, dtype=np.uint8)
    qs = make_queries(rng, text, 1200, "ACGT")
    qs[700] = t[20000:50000].tobytes()  # a record that is cut into slices
    qs[5] = b""
    qs[650] = b""
    q, off = pack(qs)
    idx = eng.Index.build(text)
    o = po.OracleIndex(bytes(text))
    st = eng.Stream(idx, 3, 1024, 4, True)
    bounds = [0, 3, 3, 10, 60, 300, 900, 1200]  # (an empty batch: 3..3)
    qa = np.array(q)  # pageable
    for b in range(len(bounds) - 1):
        w = off[bounds[b]: bounds[b + 1] + 1]
        st.submit(qa, w, 12)
        m, boff, _ = st.next()
        sub = q[int(w[0]): int(w[-1])]
        om, obc = o.match_batch(sub, w - w[0], 12, True)
        assert np.array_equal(np.diff(boff.astype(np.int64)), obc.astype(np.int64)), b
        for f in ("ref_pos", "query_pos", "length"):
            assert np.array_equal(m[f], om[f]), (b, f)
    st.close()
    idx.close()


def test_mam_long_record_matches_oracle_and_overflows_inline_slots(eng):
    """-mam on a long record (160 kbp against a 250 kbp text with repeat families): 40 verified slices per strand (the
    mode's stale fall-back interval, slamem.c:122-123,131,197-198, makes a slice depend on its past: k_find_mams_sliced),
    its MAMs leave through the v3 output path (4 inline slots, then the overflow list; rows resolved by K9).  In order
    against the oracle, both strands."""
    import time
    from oracle import pyoracle as po
    rng = np.random.default_rng(77)
    text = rand_text(rng, 250_000, "ACGT", 60, max_rep=2000)
    t = np.frombuffer(text, dtype=np.uint8).copy()
    q = t[40_000:200_000].copy()
    mut = rng.random(q.shape[0]) < 0.01
    q[mut] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(mut.sum()))
    qs = [q.tobytes(), q[::-1].tobytes()[:5000]]
    qq, off = pack(qs)
    o = po.OracleIndex(text)
    om, obc = o.match_batch(qq, off, 20, True, mam=True)
    g = eng.Index.build(text)
    t0 = time.time()
    gm, goff = g.find_mems(qq, off, 20, True, mam=True)
    dt = time.time() - t0
    assert len(om) > 100  # far more than the inline slots of a strand
    assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
    for f in ("ref_pos", "query_pos", "length"):
        assert np.array_equal(gm[f], om[f]), f
    assert dt < 20.0
    g.close()


@pytest.mark.parametrize("env", [{}, {"SLAMEM_MAM_WARMUP": "8"}, {"SLAMEM_MAM_WARMUP": "100000000"}, {"SLAMEM_MAM_WHOLE": "1"}],
                         ids=["default", "warmup8", "warmup_whole_strand", "whole_strands_kernel"])
def test_mam_slices_equal_the_whole_strand_scan(eng, env):
    """-mam over 4096-position slices of long strands (tests/mam_slices_check.py in a child process): the start state of a
    slice is guessed from a warm-up and verified against the state its right neighbour ends with; wrong guesses are
    scanned again.  Whatever the warm-up (8 positions: nearly every guess wrong; unbounded: every guess right; the
    default), and with the one-lane-per-strand kernel, the MAMs are the oracle's whole-strand scan's, in order."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "mam_slices_check.py"), "mam"], env=dict(os.environ, SLAMEM_MAM_TRACE="1", **env),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert out["all_equal"], out
    assert all(c["mams"] > 100 for c in out["cases"].values())
    trace = r.stderr.decode()
    if env.get("SLAMEM_MAM_WARMUP") == "8":
        assert "scanned again" in trace and " 0 scanned again" not in trace  # the verification did its work
    if env.get("SLAMEM_MAM_WARMUP") == "100000000":
        assert " 0 scanned again in 0 passes" in trace  # a warm-up from the strand's end is the whole scan


@pytest.mark.parametrize("env", [{}, {"SLAMEM_SLICE_WARMUP": "8"}, {"SLAMEM_SLICE_WARMUP": "100000000"}, {"SLAMEM_TEXT_SECTIONS": "0"}],
                         ids=["default", "warmup8", "warmup_whole_strand", "index_without_text_sections"])
def test_mem_slices_equal_the_whole_strand_scan(eng, env):
    """-mem over 4096-position slices of long strands (tests/mam_slices_check.py mem, in a child process): k_slice_states
    gives every slice the state the full scan has at its right end -- from a warm-up when an extension failed in it, by
    comparing the query with the text when the warm-up's match is one row (an exact 120 kbp copy among the queries), by a
    longer warm-up otherwise (and always, on an index built without the text sections).  Whatever the warm-up, the MEMs
    are the oracle's whole-strand scan's, in order."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "mam_slices_check.py"), "mem"], env=dict(os.environ, **env),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert out["all_equal"], out
    assert all(c["mams"] > 100 for c in out["cases"].values())


def test_skip_variant_is_exact(eng):
    """The K8 instantiation with the skipping states (SLAMEM_SKIP=1; measured and not the default) must give
    the same MEMs in the same order: tests/skip_variant_check.py in a child process (the switch is read once per process),
    four read sets against the oracle and the full-size digest of the REAL reference; the skips must actually happen."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SLAMEM_SKIP="1", SLAMEM_SEED_SEARCH="0")  # (the variant is K8's: the index walk answers the reads)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "skip_variant_check.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert out["config3_digest_equal"]
    assert all(out[c]["equal_in_order"] for c in ("random", "repeats", "dense_subs", "l30"))
    assert out["random"]["skips"] > 1000 and out["l30"]["skips"] > 100


@pytest.mark.parametrize("mam", [False, True], ids=["mem", "mam"])
@pytest.mark.parametrize("alpha,n,repeats,nrun,l,both", [
    ("ACGT", 3000, 3, 0, 20, True), ("ACGTN", 2500, 2, 120, 8, True), ("ACGT", 200000, 40, 0, 15, True),
    ("AC", 1500, 2, 0, 10, False)])
def test_compact_layout_matches_oracle_in_order(eng, alpha, n, repeats, nrun, l, both, mam):
    """SLAMEM_LAYOUT_COMPACT (no text-ordered sections, half-size presence filter): the same MEMs / MAMs in the same order
    as the oracle (slamem.c:114-199), in a smaller arena than the full layout's; long records (slices) included."""
    from oracle import pyoracle as po
    from slamem_amd import capi
    rng = np.random.default_rng(3 * n + l)
    text = rand_text(rng, n, alpha, repeats, nrun=nrun)
    qs = make_queries(rng, text, 60, alpha) + [b"", b"N" * 25, text[:30], text[-30:], text[: min(n, 9000)]]
    q, off = pack(qs)
    o = po.OracleIndex(text)
    om, obc = o.match_batch(q, off, l, both, mam=mam)
    full = eng.Index.build(text, layout=capi.LAYOUT_FULL)
    g = eng.Index.build(text, layout=capi.LAYOUT_COMPACT)
    assert g.info.layout == capi.LAYOUT_COMPACT and full.info.layout == capi.LAYOUT_FULL
    assert g.info.arena_bytes < full.info.arena_bytes
    assert eng.build_bytes(n, capi.LAYOUT_COMPACT)[0] == g.info.arena_bytes or "N" in alpha  # the estimate (no N rows counted)
    gm, goff = g.find_mems(q, off, l, both, mam=mam)
    assert np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64))
    for f in ("ref_pos", "query_pos", "length"):
        assert np.array_equal(gm[f], om[f]), f
    # an exported compact arena attaches like any other
    g2 = eng.Index.attach(g.export_arena())
    assert g2.info.layout == capi.LAYOUT_COMPACT
    gm2, _ = g2.find_mems(q, off, l, both, mam=mam)
    assert np.array_equal(gm2, gm)
    g2.close()
    g.close()
    full.close()


def test_auto_layout_follows_the_free_hbm(eng, monkeypatch):
    """SLAMEM_LAYOUT_AUTO: full when its build peak fits what is free (SLAMEM_HBM_BUDGET_GB caps that), compact when only
    that fits, SLAMEM_ERR_NOMEM with the numbers otherwise -- never a failed hipMalloc halfway through a build."""
    from slamem_amd import capi
    n = 40_000_000
    ref = eng.synth_reference(n, 9, "cuda:0")
    a_full, p_full = eng.build_bytes(n, capi.LAYOUT_FULL)
    a_comp, p_comp = eng.build_bytes(n, capi.LAYOUT_COMPACT)
    assert a_comp < 0.62 * a_full and p_comp < p_full
    monkeypatch.delenv("SLAMEM_INDEX_LAYOUT", raising=False)
    monkeypatch.delenv("SLAMEM_HBM_BUDGET_GB", raising=False)
    g = eng.Index.build(ref)
    assert g.info.layout == capi.LAYOUT_FULL and g.info.arena_bytes == a_full
    g.close()
    monkeypatch.setenv("SLAMEM_HBM_BUDGET_GB", f"{(p_comp + p_full) / 2 / 2**30:.3f}")
    g = eng.Index.build(ref)
    assert g.info.layout == capi.LAYOUT_COMPACT and g.info.arena_bytes == a_comp
    reads = eng.synth_reads(ref, 0, 20_000, 150, 0.02, 9, 50)
    import torch
    offsets = torch.arange(20_001, dtype=torch.int64, device="cuda:0") * 150
    m = g.matcher(20_000, True, 200_000, 3_000_000)
    t_compact = m.run(reads, offsets, 20)
    g.close()
    monkeypatch.setenv("SLAMEM_HBM_BUDGET_GB", f"{p_comp / 2 / 2**30:.3f}")
    with pytest.raises(capi.SlamemError) as ei:
        eng.Index.build(ref)
    assert ei.value.code == capi.SLAMEM_ERR_NOMEM and "compact layout" in str(ei.value)
    monkeypatch.setenv("SLAMEM_INDEX_LAYOUT", "full")  # the environment decides for AUTO callers
    g = eng.Index.build(ref)
    assert g.info.layout == capi.LAYOUT_FULL
    m = g.matcher(20_000, True, 200_000, 3_000_000)
    assert m.run(reads, offsets, 20) == t_compact > 20_000
    g.close()


@pytest.mark.parametrize("mode", ["carry", "two_streams"])
@pytest.mark.parametrize("mam", [False, True], ids=["mem", "mam"])
def test_stream_k8_without_tail_passes_lanes_to_the_next_batch(eng, mam, mode, monkeypatch):
    """The two opt-in organisations of the search stage of slamem_stream_*, each against the oracle batch by batch.
    two_streams: SLAMEM_STREAM_SEARCH_STREAMS=2 -- even and odd batches on two search streams, a K8 that has another batch behind
    it on a part of the chip only (here 16 waves, so that every launch runs into the cap), the preparation at high priority.
    carry: SLAMEM_STREAM_CARRY=1 and five batches in flight: every K8 but the last ends when its work list is empty and passes its
    unfinished lanes -- mid-strand, in any state of the scan -- to the next batch's K8 (kCarry); their MEMs
    land in the batch they belong to.  Reads with N, repeats (multi-row intervals: enumeration jobs of carried lanes),
    min_len 9, both strands; batches of very different sizes, one of them empty: every batch equals the oracle in order
    (slamem.c:114-199)."""
    from oracle import pyoracle as po
    if mode == "carry":
        monkeypatch.setenv("SLAMEM_STREAM_CARRY", "1")
    else:
        monkeypatch.setenv("SLAMEM_STREAM_SEARCH_STREAMS", "2")
        monkeypatch.setenv("SLAMEM_STREAM_K8_WAVES", "16")
        monkeypatch.setenv("SLAMEM_STREAM_PREP_PRIORITY", "1")
    rng = np.random.default_rng(4242)
    text = rand_text(rng, 300_000, "ACGT", 120, max_rep=600)
    qs = make_queries(rng, text, 24_000, "ACGTN", maxlen=260)
    q, off = pack(qs)
    idx = eng.Index.build(text)
    o = po.OracleIndex(bytes(text))
    buf = eng.PinnedBuffer(len(q) + 64)
    buf.array[: len(q)] = q
    bounds = [0, 3000, 3040, 9000, 9000, 15000, 15500, 21000, 23900, 24000]
    wins = [off[bounds[b]: bounds[b + 1] + 1] for b in range(len(bounds) - 1)]
    st = eng.Stream(idx, 6, 1 << 20, 6000, True, mam=mam)
    nb = len(wins)
    for b in range(min(5, nb)):
        st.submit(buf.array, wins[b], 9)
    for b in range(nb):
        m, boff, _ = st.next()
        if b + 5 < nb:
            st.submit(buf.array, wins[b + 5], 9)
        w = wins[b]
        om, obc = o.match_batch(q[int(w[0]): int(w[-1])], w - w[0], 9, True, mam=mam)
        assert np.array_equal(np.diff(boff.astype(np.int64)), obc.astype(np.int64)), b
        for f in ("ref_pos", "query_pos", "length"):
            assert np.array_equal(m[f], om[f]), (b, f)
    st.close()
    buf.close()
    idx.close()
