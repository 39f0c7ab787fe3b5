"""The front end's host logic (slamem_amd/host/slamem_host.c) against the reference's observable behaviour."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import hostlib
from golden_cases import CASES, MANIFEST, case_paths, opt_value


@pytest.mark.parametrize("case", CASES)
def test_loader_reports_the_same_records_as_the_reference(case):
    """Record names, sizes and accept/reject decisions equal the '# NN [name] (len bp) OK' lines the reference printed."""
    opts = MANIFEST[case]["opts"]
    ref_fa, q_fa, _, exp_stdout = case_paths(case)
    acgt = 1 if "-n" in opts else 0
    m = int(opt_value(opts, "-m", 0))
    ref = hostlib.Loaded(ref_fa, 1, acgt, m, opt_value(opts, "-r"))
    qs = hostlib.Loaded(q_fa, 0, acgt, m, None)
    got = [(n[:50].decode(), s) for n, s in zip(ref.names + qs.names, ref.sizes + qs.sizes)]
    exp = [(mm.group(1).rstrip(), int(mm.group(2))) for mm in
           re.finditer(r"^# \d+ \[(.{50})\] \((\d+) bp\) OK$", open(exp_stdout, encoding="latin1").read(), re.M)]
    assert [(a.rstrip(), b) for a, b in got] == exp
    mm = re.search(r"^> (\d+) references? and (\d+) quer", open(exp_stdout, encoding="latin1").read(), re.M)
    assert (ref.n, qs.n) == (int(mm.group(1)), int(mm.group(2)))
    # merged text: records joined by single N separators, start positions as sequence.c:260-262
    assert ref.merged_start[0] == 0
    for k in range(1, ref.n):
        assert ref.merged_start[k] == ref.merged_start[k - 1] + ref.sizes[k - 1] + 1
        assert ref.chars[ref.merged_start[k] - 1:ref.merged_start[k]] == b"N"


def test_normalisation_table(tmp_path):
    p = tmp_path / "x.fa"
    p.write_bytes(b">r one\r\nacgtNRYK*-12 xq\nACGT>r2\nnnAC\n>empty\n\n>r3\nTT")
    a = hostlib.Loaded(str(p), 0, 0)
    assert a.names == [b"r one", b"r2", b"r3"] and a.chars == b"ACGTNNNNNNACGT" + b"NNAC" + b"TT"
    assert a.offsets == [0, 14, 18, 20]
    b = hostlib.Loaded(str(p), 0, 1)  # -n drops every non-ACGT letter
    assert b.chars == b"ACGTACGT" + b"AC" + b"TT" and b.sizes == [8, 2, 2]
    c = hostlib.Loaded(str(p), 1, 0, 3)  # merged, -m 3: the 2-letter record is dropped, its separator stays
    assert c.names == [b"r one", b"r2"] and c.chars == b"ACGTNNNNNNACGT" + b"N" + b"NNAC" + b"N"
    d = hostlib.Loaded(str(p), 1, 0, 0, "r2")
    assert d.names == [b"r2"] and d.chars == b"NNAC"
    q = tmp_path / "bad.fa"
    q.write_bytes(b"ACGT\n")
    assert hostlib.Loaded(str(q), 0).n == 0  # must start with '>'
    assert hostlib.Loaded(str(tmp_path / "missing.fa"), 0).n == 0


def test_option_parsing_quirks():
    o = hostlib.parse_options(["slaMEM", "-b", "-l", "10", "ref.fa", "q1.fa", "q2.fa"])
    assert o["both_strands"] == 1 and o["min_mem_len"] == 10 and o["files"] == ["ref.fa", "q1.fa", "q2.fa"]
    o = hostlib.parse_options(["slaMEM", "ref.fa", "q.fa"])
    assert o["min_mem_len"] == 20 and o["both_strands"] == 0 and o["out_arg"] == -1 and o["no_ns"] == 0
    o = hostlib.parse_options(["slaMEM", "-B", "-N", "-L", "7", "-O", "x.txt", "-M", "50", "ref.fa", "q.fa"])
    assert (o["both_strands"], o["no_ns"], o["min_mem_len"], o["min_seq_len"], o["out_arg"]) == (1, 1, 7, 50, 6)
    # any option starting with l/o/m/v swallows the next argument, so "-mem" eats the reference
    o = hostlib.parse_options(["slaMEM", "-mem", "ref.fa", "q.fa"])
    assert o["files"] == ["q.fa"] and o["match_type"] == 0
    o = hostlib.parse_options(["slaMEM", "-mam", "x", "ref.fa", "q.fa"])
    assert o["match_type"] == 1 and o["files"] == ["ref.fa", "q.fa"]
    o = hostlib.parse_options(["slaMEM", "-r", "'chr", "1'", "ref.fa", "q.fa"])
    assert o["ref_name"] == "chr 1" and o["files"] == ["ref.fa", "q.fa"]
    o = hostlib.parse_options(["slaMEM", "-r", "chrB", "ref.fa", "q.fa"])
    assert o["ref_name"] == "chrB" and o["files"] == ["ref.fa", "q.fa"]
    assert hostlib.parse_options(["slaMEM", "ref.fa"])["usage"] == 1
    # a one-letter option must be exactly two characters: "-bx" is not -b
    assert hostlib.parse_options(["slaMEM", "-bx", "ref.fa", "q.fa"])["both_strands"] == 0


def test_default_output_name():
    L = hostlib.lib()

    def f(a):
        p = L.slh_append_to_basename(a.encode(), b"-mems.txt")
        return C.string_at(p).decode()
    assert f("ref.fa") == "ref-mems.txt"
    assert f("./dir.v2/ref") == "./dir-mems.txt"  # last '.' of the whole path
    assert f("noext") == "noext-mems.txt"


def test_progress_dots_and_merged_lookup():
    L = hostlib.lib()
    for n in (0, 1, 9, 10, 11, 150, 4557606):
        step = n // 10
        counter, dots = 0, 0
        for _ in range(min(n, 200000)):
            if counter == step:
                dots += 1
                counter = 0
            else:
                counter += 1
        if n <= 200000:
            assert L.slh_progress_dots(n) == dots
    starts = (C.c_uint32 * 3)(0, 901, 2002)
    for pos, (eid, epos) in [(0, (0, 0)), (900, (0, 900)), (901, (1, 0)), (2001, (1, 1100)), (2002, (2, 0)), (2700, (2, 698))]:
        p = C.c_uint32(pos)
        assert (L.slh_seq_id_from_merged_pos(starts, 3, C.byref(p)), p.value) == (eid, epos)


def test_cli_fails_loudly_without_gpu(tmp_path):
    """No CPU fallback: on a box without a GPU the front end must stop with the library's error, status 255."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", hostlib.HOST_DIR], stdout=subprocess.DEVNULL)
    ref_fa, q_fa, _, _ = case_paths("acgt_l20_fwd")
    r = subprocess.run([exe, "-o", str(tmp_path / "o.txt"), ref_fa, q_fa], stdout=subprocess.PIPE)
    assert r.returncode == 255 and b"no CPU" in r.stdout
    # the same through one process (the default runs the work in a forked worker and passes its status on)
    r1 = subprocess.run([exe, "-o", str(tmp_path / "o.txt"), ref_fa, q_fa], stdout=subprocess.PIPE,
                        env=dict(os.environ, SLAMEM_FOREGROUND="1"))
    assert r1.returncode == 255 and r1.stdout == r.stdout


@pytest.mark.parametrize("env", [{}, {"SLAMEM_FOREGROUND": "1"}, {"SLAMEM_OVERLAP_MB": "0"}], ids=["worker", "one_process", "overlapped"])
def test_cli_error_exits_keep_the_reference_messages_and_status(env, tmp_path):
    """Error paths that end before any GPU work (tools.c:21-25 exit(-1) -> status 255), through the forked worker, in one
    process, and with the loader thread: same text on stdout, same status, nothing left hanging."""
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", hostlib.HOST_DIR], stdout=subprocess.DEVNULL)
    ref_fa, q_fa, _, _ = case_paths("acgt_l20_fwd")
    e = dict(os.environ, **env)
    r = subprocess.run([exe, str(tmp_path / "missing.fa"), q_fa], stdout=subprocess.PIPE, env=e, timeout=60)
    assert r.returncode == 255 and b"> ERROR: No valid sequences found in reference file" in r.stdout
    r = subprocess.run([exe, ref_fa], stdout=subprocess.PIPE, env=e, timeout=60)
    assert r.returncode == 255 and b"Usage:" in r.stdout                      # argc < 3
    r = subprocess.run([exe, "-l", "20", ref_fa], stdout=subprocess.PIPE, env=e, timeout=60)
    assert r.returncode == 255 and b"> ERROR: Not enough input sequence files provided" in r.stdout
    r = subprocess.run([exe, "-r", ref_fa, q_fa], stdout=subprocess.PIPE, env=e, timeout=60)
    assert r.returncode == 255


def test_parallel_loader_equals_sequential(tmp_path):
    """Query files above 64 MB are parsed by several threads; the result must equal the one-thread parse,
    including records whose '>' is not at a line start and headers that contain '>'."""
    rng = np.random.default_rng(4)
    p = tmp_path / "big.fa"
    alpha = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)
    with open(p, "wb") as f:
        size = 0
        k = 0
        while size < (70 << 20):
            L = int(rng.integers(1, 400))
            seq = rng.choice(alpha, size=L).tobytes()
            if k % 1000 == 7:
                rec = b">r%d has > inside\r\n" % k + seq[: L // 2] + b"\n" + seq[L // 2:] + b">glued%d\n" % k + seq + b"\n"
            elif k % 1000 == 9:
                rec = b">empty%d\n\n" % k
            else:
                rec = b">r%d desc\n" % k + seq + b"\n"
            f.write(rec)
            size += len(rec)
            k += 1
    os.environ["SLAMEM_THREADS"] = "1"
    a = hostlib.Loaded(str(p), 0, 0, 30)
    os.environ["SLAMEM_THREADS"] = "7"
    b = hostlib.Loaded(str(p), 0, 0, 30)
    del os.environ["SLAMEM_THREADS"]
    assert a.n == b.n > 100000 and a.names == b.names and a.sizes == b.sizes
    assert a.offsets == b.offsets and a.chars == b.chars


@pytest.mark.parametrize("case", CASES)
def test_hidden_sort_and_clean_utilities_match_the_reference(case, tmp_path):
    """`slaMEM -s <mems_file>` (SortMEMsFile, slamem.c:244-352) and `slaMEM -c <fasta_file>` (CleanFasta,
    slamem.c:455-523): output files, messages and exit status equal what the real reference produced for the same
    files (tests/golden/<case>/sorted.txt, sort-stdout.txt, clean.fasta, clean-stdout.txt).  No GPU involved."""
    import shutil
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", hostlib.HOST_DIR], stdout=subprocess.DEVNULL)
    ref_fa, _, exp_mems, _ = case_paths(case)
    d = os.path.dirname(exp_mems)
    shutil.copy(exp_mems, tmp_path / "x-mems.txt")
    shutil.copy(ref_fa, tmp_path / "r.fa")
    r = subprocess.run([exe, "-s", "x-mems.txt"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.stdout + b"status %d\n" % r.returncode == open(os.path.join(d, "sort-stdout.txt"), "rb").read()
    if os.path.exists(os.path.join(d, "sorted.txt")):
        assert open(tmp_path / "x-mems-sorted.txt", "rb").read() == open(os.path.join(d, "sorted.txt"), "rb").read()
    else:
        assert not os.path.exists(tmp_path / "x-mems-sorted.txt")
    r = subprocess.run([exe, "-c", "r.fa"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.stdout + b"status %d\n" % r.returncode == open(os.path.join(d, "clean-stdout.txt"), "rb").read()
    assert open(tmp_path / "r-clean.fasta", "rb").read() == open(os.path.join(d, "clean.fasta"), "rb").read()


def test_pieces_loader_equals_whole_file_loader(tmp_path):
    """slh_pieces_*: a query file handed out in pieces (what lets the front end search while it parses) holds exactly the
    records of slh_load_file, in order: names, lengths, normalised characters -- with short records that -m drops, CRLF
    lines, lower case / IUPAC letters and a last record without a final newline."""
    import ctypes as C
    import numpy as np
    L = hostlib.lib()
    L.slh_pieces_open.restype = C.c_void_p
    L.slh_pieces_open.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_int, C.c_long, C.c_long, C.c_void_p]
    L.slh_pieces_next.argtypes = [C.c_void_p, C.POINTER(hostlib.SeqSet)]
    L.slh_pieces_close.argtypes = [C.c_void_p]
    rng = np.random.default_rng(3)
    parts = []
    for k in range(30000):
        n = int(rng.integers(1, 400))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtNRYn", dtype=np.uint8), size=n))
        nl = b"\r\n" if k % 7 == 0 else b"\n"
        parts.append(b">r%d some description" % k + nl + seq[: n // 2] + nl + seq[n // 2:] + (b"" if k == 29999 else nl))
    path = str(tmp_path / "q.fa")
    open(path, "wb").write(b"".join(parts))
    for acgt_only, min_len in ((0, 0), (1, 50)):
        whole = hostlib.Loaded(path, 0, acgt_only, min_len)
        h = L.slh_pieces_open(path.encode(), acgt_only, min_len, 1, 100, 1 << 20, None)
        assert h
        names, sizes, chars, npieces = [], [], [], 0
        while True:
            s = hostlib.SeqSet()
            n = L.slh_pieces_next(h, C.byref(s))
            if n == 0:
                break
            npieces += 1
            assert n == s.num
            names += [s.recs[i].name for i in range(s.num)]
            sizes += [s.recs[i].size for i in range(s.num)]
            assert [s.offsets[i + 1] - s.offsets[i] for i in range(s.num)] == sizes[-s.num:]
            chars.append(C.string_at(s.chars, s.total))
            L.slh_free_seqset(C.byref(s))
        L.slh_pieces_close(h)
        assert npieces >= 4
        assert names == whole.names and sizes == whole.sizes and b"".join(chars) == whole.chars


def test_formatter_numbers_of_every_width(tmp_path):
    """slamem.c:148 prints "%d\\t%d\\t%d\\n" of 1-based positions: every digit count from 1 to 10, the pair-table edges."""
    ref_fa = tmp_path / "r.fa"
    ref_fa.write_bytes(b">r\nACGT\n")
    ref = hostlib.Loaded(str(ref_fa), 1)
    vals = [0, 8, 9, 10, 98, 99, 100, 101, 999, 1000, 9999, 10000, 99999, 100000, 999999, 1000000, 9999999, 10000000,
            99999999, 100000000, 999999999, 1000000000, 2147483647, 4294967294]
    mems = np.array([(v, vals[-1 - i], (v % 1000003) + 1) for i, v in enumerate(vals)], dtype=np.uint32)
    got = hostlib.format_block(b"q name", 1, mems, ref)
    want = b">q name Reverse\n" + b"".join(b"%d\t%d\t%d\n" % (int(a) + 1, int(b) + 1, int(c)) for a, b, c in mems)
    assert got == want


def _reference_query_loader(data: bytes, acgt_only: bool):
    """The query loader's rules, byte by byte (sequence.c:128-215 for one file, no filter, no minimum length)."""
    keep = {}
    for ch in range(ord("A"), ord("Z") + 1):
        if not acgt_only:
            keep[ch] = keep[ch + 32] = ord("N")
    for ch in b"ACGT":
        keep[ch] = keep[ch + 32] = ch
    names, seqs, i, n = [], [], 0, len(data)
    if not data or data[0] != ord(">"):
        return names, seqs
    while i < n:
        while i < n and data[i] != ord(">"):
            i += 1
        if i >= n:
            break
        i += 1
        j = i
        while j < n and data[j] not in (10, 13):
            j += 1
        name = data[i:j]
        i = j + 1 if j < n else j
        out = bytearray()
        while i < n and data[i] != ord(">"):
            if data[i] in keep:
                out.append(keep[data[i]])
            i += 1
        if out:
            names.append(name)
            seqs.append(bytes(out))
    return names, seqs


@pytest.mark.parametrize("seed", range(6))
def test_loader_fast_paths_follow_the_byte_by_byte_rules(seed, tmp_path):
    """Whole-line copies (lines of upper-case A,C,G,T) and the one-step header scan must give what the byte-by-byte rules
    give: random files mixing clean lines, lower case, CR LF, other letters, digits, blanks, '>' inside a line, empty
    lines and records, a header ended by CR, no newline at the end."""
    rng = np.random.default_rng(seed)
    parts = []
    for r in range(int(rng.integers(5, 60))):
        parts.append(b">rec%d %s" % (r, bytes(rng.choice(np.frombuffer(b"abc xyz_|", dtype=np.uint8), size=int(rng.integers(0, 12))))))
        parts.append([b"\n", b"\r\n", b"\r"][int(rng.integers(0, 3))])
        for _ in range(int(rng.integers(0, 6))):
            kind = int(rng.integers(0, 7))
            L = int(rng.integers(0, 90))
            if kind <= 2:
                line = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L))
            elif kind == 3:
                line = bytes(rng.choice(np.frombuffer(b"acgtnACGTN", dtype=np.uint8), size=L))
            elif kind == 4:
                line = bytes(rng.choice(np.frombuffer(b"ACGTRYKM*-1 \t", dtype=np.uint8), size=L))
            elif kind == 5:
                line = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)) + b">mid%d\nAC" % r  # '>' inside a line
            else:
                line = b""
            parts.append(line)
            parts.append([b"\n", b"\n", b"\r\n"][int(rng.integers(0, 3))])
    data = b"".join(parts)
    if seed % 2:
        data = data.rstrip(b"\r\n")
    p = tmp_path / "x.fa"
    p.write_bytes(data)
    for acgt_only in (0, 1):
        names, seqs = _reference_query_loader(data, bool(acgt_only))
        a = hostlib.Loaded(str(p), 0, acgt_only)
        assert a.names == names
        assert a.chars == b"".join(seqs) and a.sizes == [len(x) for x in seqs]


def _messy_fasta(rng, big=False):
    """Records of soft-masked / IUPAC / clean lines, CR LF, digits and blanks, '>' inside a line, empty records; `big`: one
    record long enough to cross the loader's 1 M-letter steps several times."""
    alph = [b"ACGT", b"acgt", b"ACGTacgt", b"ACGTNRYKMSWacgtnryk", b"ACGT*-1 \t", b"N", b"n"]
    parts = []
    for r in range(int(rng.integers(3, 25))):
        parts.append(b">chr%d some text %d" % (r, r))
        parts.append([b"\n", b"\r\n"][int(rng.integers(0, 2))])
        nlines = int(rng.integers(0, 12)) if not (big and r == 1) else 40_000
        for _ in range(nlines):
            L = int(rng.integers(0, 120)) if not (big and r == 1) else 70
            a = alph[int(rng.integers(0, len(alph)))] if not (big and r == 1) else alph[2 + int(rng.integers(0, 2))]
            parts.append(bytes(rng.choice(np.frombuffer(a, dtype=np.uint8), size=L)))
            if rng.random() < 0.02:
                parts.append(b">inner%d\nACGTTGCA" % r)
            parts.append([b"\n", b"\n", b"\r\n"][int(rng.integers(0, 3))])
    return b"".join(parts)


@pytest.mark.parametrize("seed", range(4))
def test_loader_whole_line_path_equals_the_byte_loop(seed, tmp_path):
    """Round 4: lines that hold nothing but letters -- soft-masked (lower-case) stretches, IUPAC letters -- are translated 16
    bytes at a time (`line_of_letters`: case folded, every letter that is not A,C,G,T an N, sequence.c:61-81), for reads AND for
    the reference's records (behind a record's first letter and its separator N, between the loader's 1 M-letter steps).
    Against the same library with SLAMEM_LOADER_BYTEWISE=1 (a child process: the switch is read once): names, sizes, letters
    and merged starts equal, with and without -n, with a minimum record length."""
    import json
    import subprocess
    import sys
    rng = np.random.default_rng(4000 + seed)
    data = _messy_fasta(rng, big=(seed == 0))
    if seed % 2:
        data = data.rstrip(b"\r\n")
    p = tmp_path / "m.fa"
    p.write_bytes(data)
    child = r"""
import hashlib, json, sys
sys.path.insert(0, sys.argv[1])
import hostlib
out = {}
for merge in (0, 1):
    for acgt_only in (0, 1):
        for min_len in (0, 50):
            a = hostlib.Loaded(sys.argv[2], merge, acgt_only, min_len)
            out["%d%d%d" % (merge, acgt_only, min_len)] = [a.n, [x.decode("latin1") for x in a.names], a.sizes, hashlib.sha256(a.chars).hexdigest(), len(a.chars),
                                                             a.merged_start, a.offsets]
print(json.dumps(out))
"""
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for mode in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", child, here, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, SLAMEM_LOADER_BYTEWISE=mode), timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        res[mode] = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res["0"] == res["1"]
    assert any(v[4] > 0 for v in res["0"].values())
