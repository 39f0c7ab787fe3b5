"""BASELINE.json configs[0], [3] and [4] under `pytest -m gpu`, on seeded inputs, with pinned answers
(tests/golden/known_answers.json; recorded by tests/golden/make_known_answers.py from the REAL reference where a CPU
run fits the build container, otherwise the digest of this engine's own deterministic output plus full-size
size-independent properties).  Generators: slamem_amd/synth.py == csrc/synth.hip (SURVEY.md Appendix C.2 + the repeat
model of 8(d)); the reference semantics matched are slamem.c:114-199 and, for -n, sequence.c:61-81."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
KNOWN = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))
DIGEST_KEYS = ("mems", "sum_len", "max_len", "sha256")

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


@pytest.fixture(scope="module")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from slamem_amd import engine
    return engine


def note(name, digest):
    """Keep what this run computed (scratch; how the engine-pinned digests were first recorded)."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "engine_digests.jsonl"), "a") as f:
            f.write(json.dumps({"case": name, **digest}) + "\n")
    except OSError:
        pass


def digest_of(matcher, total, nblocks):
    from mems_digest import digest_rows
    mems = matcher.mems[:total].cpu().numpy().view(np.uint32)
    boff = matcher.block_offsets[: nblocks + 1].cpu().numpy()
    rows = np.empty((total, 4), dtype=np.uint32)
    rows[:, 0] = np.repeat(np.arange(nblocks, dtype=np.uint32), np.diff(boff))
    rows[:, 1] = mems[:, 0] + 1
    rows[:, 2] = mems[:, 1] + 1
    rows[:, 3] = mems[:, 2]
    return digest_rows(rows), rows


def check_sampled_mems(rows, ref_h, reads_h, L, min_len, sample, seed=0):
    """Every sampled MEM is a real match inside its read and maximal on both sides."""
    n = ref_h.shape[0]
    total = rows.shape[0]
    assert (rows[:, 3] >= min_len).all()
    sel = np.random.default_rng(seed).choice(total, size=min(total, sample), replace=False)
    bad = 0
    for i in sel:
        g, a, b, c = int(rows[i, 0]), int(rows[i, 1]) - 1, int(rows[i, 2]) - 1, int(rows[i, 3])
        rd = reads_h[g >> 1]
        if g & 1:
            rd = _COMP[rd[::-1]]
        ok = a + c <= n and b + c <= L and bool((ref_h[a:a + c] == rd[b:b + c]).all())
        ok = ok and (a == 0 or b == 0 or ref_h[a - 1] != rd[b - 1])
        ok = ok and (a + c == n or b + c == L or ref_h[a + c] != rd[b + c])
        bad += not ok
    return bad, len(sel)


def read_starts(n, L, count, seed=42):
    """Text position every synthetic read was drawn from (slamem_amd/synth.py draw layout)."""
    from slamem_amd import synth
    base = np.uint64(n) + np.arange(count, dtype=np.uint64) * np.uint64(L + 2)
    return (synth.splitmix64_at(seed, base) % np.uint64(n - L + 1)).astype(np.int64)


def verifier_sample(n, L, count, seed, n_random, n_high, n_repeat):
    """Reads for the definitional verifier: random ones, ones drawn from beyond text position 2^31 (when the text is that
    long) and ones drawn from inside a planted repeat (source or copy), all seeded."""
    from slamem_amd import synth
    p = read_starts(n, L, count, seed)
    rng = np.random.default_rng(12345)
    picks = [rng.choice(count, size=min(count, n_random), replace=False)]
    high = np.nonzero(p >= (1 << 31))[0]
    if len(high):
        picks.append(rng.choice(high, size=min(len(high), n_high), replace=False))
    segs = np.array([(ln, src, dst) for _, ln, src, dst in synth.repeat_segments(n, seed)], dtype=np.int64)
    inside = np.zeros(count, dtype=bool)
    for col in (1, 2):
        o = np.argsort(segs[:, col])
        st, ln = segs[o, col], segs[o, 0]
        i = np.searchsorted(st, p, "right") - 1
        ok = i >= 0
        inside[ok] |= p[ok] + L <= st[i[ok]] + ln[i[ok]]
    rep = np.nonzero(inside)[0]
    picks.append(rng.choice(rep, size=min(len(rep), n_repeat), replace=False))
    return np.unique(np.concatenate(picks)), int((p[picks[-1]] >= 0).sum()), (int(len(picks[1])) if len(high) else 0)


def check_complete(ref_h, ref_dev, reads_h, rows, sample, min_len, label):
    """Set equality, on the sampled reads, of the engine's MEMs with the index-independent definitional set
    (tests/mem_verifier.py; semantics of slamem.c:139-193)."""
    import mem_verifier as mv
    res = mv.verify_sample(ref_h, ref_dev, reads_h, sample, rows, min_len, True)
    note(label, {k: v for k, v in res.items() if k not in ("missing", "extra")})
    assert res["missing_count"] == 0 and res["extra_count"] == 0 and res["engine_duplicates"] == 0, res
    assert res["definitional_mems"] > 0
    return res


def check_sampled_rows(idx, ref_h, rows_to_check):
    """Suffixes of neighbouring BWT rows are in the index's letter order ($ < N < A < C < G < T)."""
    n = ref_h.shape[0]
    r = np.asarray(rows_to_check, dtype=np.int64)
    sa_prev = idx.position_in_text((r - 1).astype(np.uint32)).astype(np.int64)
    sa_cur = idx.position_in_text(r.astype(np.uint32)).astype(np.int64)
    code = np.zeros(256, dtype=np.uint8)
    for k, ch in enumerate(b"NACGT"):
        code[ch] = k + 1
    bad = 0
    for a, b in zip(sa_prev, sa_cur):
        a, b = int(a), int(b)
        if a == n:  # row 0 is the '$' suffix: smaller than everything
            continue
        m = min(n - a, n - b, 1 << 16)
        x, y = ref_h[a:a + m], ref_h[b:b + m]
        d = np.nonzero(x != y)[0]
        if len(d):
            bad += not (code[x[d[0]]] < code[y[d[0]]])
        else:
            bad += not (a > b)  # one is a prefix of the other: the shorter suffix sorts first
    return bad


def test_config1_genome_pair_cli_byte_identical_to_reference(eng, tmp_path):
    """configs[0] stand-in (tests/golden_cases.py::ecoli_like_pair): a 4.64 Mbp genome against a 4.56 Mbp strain, -b -l 20,
    through the slaMEM-hip command line.  The output FILE must have the sha256 of the file the REAL reference wrote for
    the same FASTAs (known_answers.json: config1_pair; 50,338 MEMs, longest 886)."""
    from golden_cases import ecoli_like_pair
    from slamem_amd import synth
    known = KNOWN["config1_pair"]
    assert known["reference_valid"]
    ref, qry = ecoli_like_pair()
    synth.write_fasta_reference(str(tmp_path / "ref.fa"), ref, "ecoli_like_ref")
    synth.write_fasta_reference(str(tmp_path / "qry.fa"), qry, "ecoli_like_strain")
    exe = os.path.join(ROOT, "slamem_amd", "host", "slaMEM-hip")
    r = subprocess.run([exe, "-b", "-l", "20", "-o", "out.txt", "ref.fa", "qry.fa"], cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-2000:]
    data = (tmp_path / "out.txt").read_bytes()
    assert len(data) == known["file_bytes"]
    assert hashlib.sha256(data).hexdigest() == known["file_sha256"]


def test_config1_genome_pair_mam_cli_byte_identical_to_reference(eng, tmp_path):
    """The same genome pair with -mam: 2226 slices whose start states are guessed and verified (k_find_mams_sliced); the
    output FILE must have the sha256 of the file the REAL reference wrote (known_answers.json: config1_pair_mam)."""
    from golden_cases import ecoli_like_pair
    from slamem_amd import synth
    known = KNOWN["config1_pair_mam"]
    ref, qry = ecoli_like_pair()
    synth.write_fasta_reference(str(tmp_path / "ref.fa"), ref, "ecoli_like_ref")
    synth.write_fasta_reference(str(tmp_path / "qry.fa"), qry, "ecoli_like_strain")
    exe = os.path.join(ROOT, "slamem_amd", "host", "slaMEM-hip")
    r = subprocess.run([exe, "-b", "-l", "20", "-o", "out.txt", "ref.fa", "qry.fa", "-mam"], cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-2000:]
    data = (tmp_path / "out.txt").read_bytes()
    assert len(data) == known["file_bytes"]
    assert hashlib.sha256(data).hexdigest() == known["file_sha256"]


def test_config2_mam_known_answer(eng):
    """-mam at the headline workload's scale: the 100 Mbp reference of configs[1]/[2] and the first 200,000 of its reads,
    -b -l 20 -mam, on K8's kMam instantiation -- against the digest of what the REAL reference printed for the same FASTA
    files (known_answers.json: config2_mam_first200k; every line of it checked against the texts)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from mems_digest import digest_rows
    known = KNOWN["config2_mam_first200k"]
    assert known["reference_valid"]
    n, nreads, L = 100_000_000, 200_000, 150
    ref = eng.synth_reference(n, 42, "cuda:0")
    reads = eng.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)
    offsets = torch.arange(nreads + 1, dtype=torch.int64, device="cuda:0") * L
    idx = eng.Index.build(ref, "cuda:0")
    m = eng.Matcher(idx, nreads, True, 8 * nreads, nreads * L, mam=True)
    total = m.run(reads, offsets, 20)
    assert total == known["mems"]
    mems = m.mems[:total].cpu().numpy().view(np.uint32)
    boff = m.block_offsets[: 2 * nreads + 1].cpu().numpy()
    rows = np.empty((total, 4), dtype=np.uint32)
    rows[:, 0] = np.repeat(np.arange(2 * nreads, dtype=np.uint32), np.diff(boff))
    rows[:, 1] = mems[:, 0] + 1
    rows[:, 2] = mems[:, 1] + 1
    rows[:, 3] = mems[:, 2]
    assert digest_rows(rows) == {k: known[k] for k in ("mems", "sum_len", "max_len", "sha256")}
    idx.close()


def test_config4_chr1_sized_known_answer(eng):
    """configs[3] (human chr1 stand-in): 248 Mbp text WITH the repeat model, 150 bp reads, -b -l 50.
    (1) the first 1 M reads against the answer of the REAL reference (known_answers.json: config4_first1M);
    (2) one GPU's share of the 50 M reads (6.25 M): every sampled MEM real and two-sided maximal, neighbouring rows in
        suffix order, and the digest of the engine's deterministic output pinned (config4_share_engine)."""
    import torch
    known = KNOWN["config4_first1M"]
    n, L, min_len = 248_000_000, 150, 50
    dev = "cuda:0"
    ref = eng.synth_reference(n, 42, dev)
    planted = eng.synth_plant_repeats(ref, 42)
    assert planted >= n // 200
    idx = eng.Index.build(ref, dev)
    assert idx.info.max_lcp >= 255  # the repeat model must exercise the long-LCP machinery
    nreads = 1_000_000
    reads = eng.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)
    offsets = torch.arange(nreads + 1, dtype=torch.int64, device=dev) * L
    m = idx.matcher(nreads, True, 8 * nreads, nreads * L)
    total = m.run(reads, offsets, min_len)
    got, rows = digest_of(m, total, 2 * nreads)
    note("config4_first1M", got)
    assert got == {k: known[k] for k in DIGEST_KEYS}, (got, known.get("reference_valid"))
    del m
    share = 6_250_000
    reads = eng.synth_reads(ref, 0, share, L, 0.02, 42, 50)
    offsets = torch.arange(share + 1, dtype=torch.int64, device=dev) * L
    m = idx.matcher(share, True, 8 * share, share * L)
    total = m.run(reads, offsets, min_len)
    got, rows = digest_of(m, total, 2 * share)
    ref_h = ref.cpu().numpy()
    reads_h = reads[: share * L].cpu().numpy().reshape(share, L)
    bad, checked = check_sampled_mems(rows, ref_h, reads_h, L, min_len, 30_000)
    assert bad == 0 and checked == 30_000
    rr = np.random.default_rng(1).integers(1, n + 1, size=1500)
    assert check_sampled_rows(idx, ref_h, rr) == 0
    # completeness: the definitional MEM set of 2,500 sampled reads (500 of them drawn from planted repeats) == the engine's
    sample, in_repeats, _ = verifier_sample(n, L, share, 42, 2000, 0, 500)
    assert in_repeats >= 400
    check_complete(ref_h, ref, reads_h, rows, sample, min_len, "config4_share_verifier")
    note("config4_share_engine", got)
    assert got == {k: KNOWN["config4_share_engine"][k] for k in DIGEST_KEYS}, got
    idx.close()


def test_config5_grch38_sized_full_size_properties(eng):
    """configs[4] (GRCh38 stand-in, > 2^31 BWT rows): 3.1 Gbp text with the repeat model, one GPU's share of the
    100 M reads (12.5 M x 150 bp), -b -l 20 (-n strips non-ACGT letters on the host, sequence.c:61-81: a no-op on this
    ACGT text, exercised on small inputs by the golden `normalise_*` cases).  The first 100,000 reads are pinned by the REAL
    reference (round 3); for the rest no CPU run fits, so:
    the suffix array is a permutation (sum and sum of squares over ALL rows, on the device), neighbouring sampled rows
    are in suffix order, every sampled MEM is real and two-sided maximal, and the digest of the engine's deterministic
    output is pinned (config5_share_engine) so that regressions show."""
    import torch
    n, L, min_len, share = 3_100_000_000, 150, 20, 12_500_000
    dev = "cuda:0"
    ref = eng.synth_reference(n, 42, dev)
    eng.synth_plant_repeats(ref, 42)
    idx = eng.Index.build(ref, dev)
    assert idx.n == n and idx.info.max_lcp >= 255
    # SA is a permutation of 0..n: closed-form sums modulo 2^64 over every row, computed on the device
    arena = idx.arena_view()
    off_sa = int(np.frombuffer(arena[:64].cpu().numpy().tobytes(), dtype=np.uint64)[5])  # ArenaHeader.off_sa
    sa32 = arena[off_sa: off_sa + 4 * (n + 1)].view(torch.int32)
    s1 = s2 = 0
    step = 1 << 28
    for a in range(0, n + 1, step):
        v = sa32[a:a + step].to(torch.int64) & 0xFFFFFFFF
        s1 += int(v.sum().item())
        s2 = (s2 + (int((v * v).sum().item()) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF  # int64 wraps = mod 2^64
        del v
    assert s1 == n * (n + 1) // 2
    assert s2 == (n * (n + 1) * (2 * n + 1) // 6) & 0xFFFFFFFFFFFFFFFF
    reads = eng.synth_reads(ref, 0, share, L, 0.02, 42, 50)
    offsets = torch.arange(share + 1, dtype=torch.int64, device=dev) * L
    m = idx.matcher(share, True, 8 * share, share * L)
    total = m.run(reads, offsets, min_len)
    got, rows = digest_of(m, total, 2 * share)
    ref_h = ref.cpu().numpy()
    reads_h = reads[: share * L].cpu().numpy().reshape(share, L)
    bad, checked = check_sampled_mems(rows, ref_h, reads_h, L, min_len, 30_000)
    assert bad == 0 and checked == 30_000
    assert int(rows[:, 1].max()) > (1 << 31)  # matches beyond text position 2^31 are found
    # the REAL reference on this text (6,842 s and ~30 GB in the build container; known_answers.json: config5_first100k): the
    # first 100,000 reads, both strands -- 298,187 MEMs, 91,832 of them beyond text position 2^31, every line checked against
    # the texts when it was recorded
    from mems_digest import digest_rows
    ref100k = KNOWN["config5_first100k"]
    assert ref100k["reference_completed"] and ref100k["reference_valid"]
    first = rows[rows[:, 0] < 200_000]
    assert digest_rows(first) == {k: ref100k[k] for k in DIGEST_KEYS}
    assert int((first[:, 1].astype(np.int64) > (1 << 31)).sum()) == ref100k["rows_beyond_2p31"]
    # a second, disjoint pin by the REAL reference (round 4): reads 7,000,000 .. 7,099,999 of the same share
    ref7m = KNOWN.get("config5_reads_7M")
    if ref7m is not None:
        assert ref7m["reference_completed"] and ref7m["reference_valid"]
        lo = 2 * 7_000_000
        part = rows[(rows[:, 0] >= lo) & (rows[:, 0] < lo + 200_000)].copy()
        part[:, 0] -= lo  # (the reference numbered its 100,000 queries from 0)
        assert digest_rows(part) == {k: ref7m[k] for k in DIGEST_KEYS}
        assert int((part[:, 1].astype(np.int64) > (1 << 31)).sum()) == ref7m["rows_beyond_2p31"]
    # completeness beyond 2^31 rows: the definitional MEM set (every maximal match >= l, from the text and the reads alone,
    # no index) of 2,600 sampled reads -- 1,500 random, 600 drawn from beyond position 2^31, 500 from planted repeats --
    # equals the engine's output for those reads as a set
    sample, in_repeats, high = verifier_sample(n, L, share, 42, 1500, 600, 500)
    assert in_repeats >= 400 and high == 600
    res = check_complete(ref_h, ref, reads_h, rows, sample, min_len, "config5_share_verifier")
    assert res["definitional_beyond_2p31"] > 500
    rr = np.random.default_rng(2).integers(1, n + 1, size=1500)
    assert check_sampled_rows(idx, ref_h, rr) == 0
    note("config5_share_engine", got)
    assert got == {k: KNOWN["config5_share_engine"][k] for k in DIGEST_KEYS}, got
    idx.close()


@pytest.mark.parametrize("mam", [False, True], ids=["mem", "mam"])
def test_genome_pair_with_exact_repeats_mem_and_mam_cli(eng, tmp_path, mam):
    """The genome pair with exact repeats in the genome (7 x 5,000 bp, 20 x 1,300 bp: ecoli_like_pair(duplicates=True)) -- an
    input on which -mem and -mam print DIFFERENT files (58,125 against 49,965 lines; on the plain pair they are the same
    file, so that pin cannot tell the modes apart).  Both through the slaMEM-hip command line, each byte-identical to the file
    the REAL reference wrote (known_answers.json: config1_dups_pair, config1_dups_pair_mam; slamem.c:131,657)."""
    from golden_cases import ecoli_like_pair
    from slamem_amd import synth
    known = KNOWN["config1_dups_pair_mam" if mam else "config1_dups_pair"]
    assert KNOWN["config1_dups_pair_mam"]["file_sha256"] != KNOWN["config1_dups_pair"]["file_sha256"]
    ref, qry = ecoli_like_pair(duplicates=True)
    synth.write_fasta_reference(str(tmp_path / "ref.fa"), ref, "ecoli_like_ref")
    synth.write_fasta_reference(str(tmp_path / "qry.fa"), qry, "ecoli_like_strain")
    exe = os.path.join(ROOT, "slamem_amd", "host", "slaMEM-hip")
    r = subprocess.run([exe, "-b", "-l", "20", "-o", "out.txt", "ref.fa", "qry.fa"] + (["-mam"] if mam else []),
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-2000:]
    data = (tmp_path / "out.txt").read_bytes()
    assert len(data) == known["file_bytes"]
    assert hashlib.sha256(data).hexdigest() == known["file_sha256"]
    if not mam:  # search -> picture: the -v tool on this output writes the picture the reference drew of ITS output (slamem.c:354-452)
        img = KNOWN["config1_dups_pair_image"]
        assert img["mems_file_sha256"] == known["file_sha256"]
        r = subprocess.run([exe, "-v", "out.txt", "ref.fa", "qry.fa"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0 and r.stdout.endswith(b"> Saving image to <out.bmp> ... OK\n> Done!\n")
        bmp = (tmp_path / "out.bmp").read_bytes()
        assert len(bmp) == img["image_bytes"] and hashlib.sha256(bmp).hexdigest() == img["image_sha256"]


def test_genome_like_generator_gpu_equals_numpy(eng):
    """csrc/synth.hip and slamem_amd/synth.py make the same genome-like text (family, satellite, block of N) and the same reads
    (none drawn from the block of N)."""
    import torch
    from slamem_amd import synth
    n, L = 12_000_000, 150
    ref_h = synth.make_reference(n, 5)
    synth.plant_repeats(ref_h, 5)
    lay = synth.plant_genome_like(ref_h, 5)
    ref_d = eng.synth_reference(n, 5, "cuda:0")
    eng.synth_plant_repeats(ref_d, 5)
    eng.synth_plant_genome_like(ref_d, 5)
    assert np.array_equal(ref_d.cpu().numpy(), ref_h)
    avoid = (lay["n_block_at"], lay["n_block_letters"])
    rh = synth.make_reads(ref_h, 100, 30_000, L, 0.02, 5, 50, avoid=avoid)
    rd = eng.synth_reads(ref_d, 100, 30_000, L, 0.02, 5, 50, avoid=avoid)[: 30_000 * L].cpu().numpy().reshape(30_000, L)
    assert np.array_equal(rd, rh)
    assert not (rh == ord("N")).any() and (ref_h == ord("N")).sum() == lay["n_block_letters"]


def test_config4_genome_like_repeat_load_known_answer(eng):
    """A realistic repeat load on the chr1-sized text: 100,000 copies of a 300 bp family at 5-15 % divergence and a 171 bp x
    10^4 satellite array on top of SURVEY's repeat model (the model's 30 Mbp block of N is left out here: the reference's LCP
    restoration is quadratic on it and does not finish; tools/repeat_load.py and the generator test keep it); the first
    100,000 reads, -b -l 50, against the digest of what the REAL reference printed (known_answers.json: config4_genome_like_first100k; slamem.c:139-193), plus the
    definitional verifier on 400 of the reads (at least 100 of them drawn from family copies)."""
    import torch
    from slamem_amd import synth
    if "config4_genome_like_first100k" not in KNOWN:
        pytest.fail("tests/golden/known_answers.json has no config4_genome_like_first100k (tests/golden/make_known_answers.py)")
    known = KNOWN["config4_genome_like_first100k"]
    assert known["reference_valid"]
    n, nreads, L, min_len = 248_000_000, 100_000, 150, 50
    dev = "cuda:0"
    ref = eng.synth_reference(n, 42, dev)
    eng.synth_plant_repeats(ref, 42)
    eng.synth_plant_genome_like(ref, 42, n_block=False)
    copies, stride, sat_at, n_at, n_len = synth.genome_like_layout(n)
    idx = eng.Index.build(ref, dev)
    assert idx.info.num_n_rows == 0
    reads = eng.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)
    offsets = torch.arange(nreads + 1, dtype=torch.int64, device=dev) * L
    m = idx.matcher(nreads, True, 64 * nreads, nreads * L)
    total = m.run(reads, offsets, min_len)
    got, rows = digest_of(m, total, 2 * nreads)
    note("config4_genome_like_first100k", got)
    assert got == {k: known[k] for k in DIGEST_KEYS}, got
    # completeness on reads drawn from family copies (their matches spread over thousands of BWT rows) and random ones
    p = read_starts(n, L, nreads, 42)
    dst = (np.arange(copies, dtype=np.uint64) * np.uint64(stride)
           + synth.splitmix64_at((42 + synth.GENOME_SALT + 1) & 0xFFFFFFFFFFFFFFFF, np.arange(copies, dtype=np.uint64)) % np.uint64(stride - 300)).astype(np.int64)
    k = np.minimum(p // stride, copies - 1)
    fam = np.nonzero((p + L > dst[k]) & (p < dst[k] + 300))[0]
    rng = np.random.default_rng(77)
    # (a family read's 21-letter seeds hit ~10^5 text positions: 100 + ~40 of them keep the host-side join to seconds)
    sample = np.unique(np.concatenate([rng.choice(nreads, 300, replace=False), rng.choice(fam, 100, replace=False)]))
    assert len(fam) >= 5000
    ref_h = ref.cpu().numpy()
    reads_h = reads[: nreads * L].cpu().numpy().reshape(nreads, L)
    check_complete(ref_h, ref, reads_h, rows, sample, min_len, "config4_genome_like_verifier")
    # Round 4: on this text K8 queues its enumeration jobs (kDefer: k_enum_jobs runs them with the whole chip, K9 puts the MEM
    # numbers right).  The same reads at -l 20 (3,800 MEMs per read, single strands with 10^5), jobs in the queue against jobs in
    # the waves (SLAMEM_ENUM_DEFER=0, the path the line above pins at -l 50 through the same kernels' in-wave form), and the
    # index walk alone against the seed path in front of it: every strand's MEMs row for row in the same order.
    import os
    from conftest import search_path
    small = 20_000
    ms = idx.matcher(small, True, 120_000_000, small * L)
    outs = {}
    for name, defer, path in (("queue", "1", "seed"), ("waves", "0", "seed"), ("queue, index walk only", "1", "walk")):
        os.environ["SLAMEM_ENUM_DEFER"] = defer
        try:
            with search_path(path):
                tot = ms.run(reads[: small * L + 16], offsets[: small + 1], 20)
        finally:
            os.environ.pop("SLAMEM_ENUM_DEFER", None)
        outs[name] = (tot, ms.block_offsets.clone(), ms.mems[:tot].clone())
    t0, b0, r0 = outs["waves"]
    assert t0 > 50_000_000
    for name in ("queue", "queue, index walk only"):
        t1, b1, r1 = outs[name]
        assert t1 == t0 and torch.equal(b1, b0) and torch.equal(r1, r0), name
    idx.close()
