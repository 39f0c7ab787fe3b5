#!/usr/bin/env python3
"""Record known answers of the REAL reference (oracle/_ref/slaMEM, compiled from /root/reference by oracle/Makefile;
build container only) for workloads too large for per-MEM fixtures: the answer is the digest of tools/mems_digest.py
-- {mems, sum_len, max_len, sha256 over the sorted (strand block, ref, query, length) rows} -- stored under a case name
in tests/golden/known_answers.json and asserted by the -m gpu tests on the same seeded inputs.

    tests/golden/make_known_answers.py <case> [--keep]

Cases (inputs: slamem_amd/synth.py generators, SURVEY.md Appendix C.2 + the repeat model of 8(d)):
  config4_first1M   BASELINE.json configs[3]: 248 Mbp text WITH the repeat model, the first 1 M of the 150 bp reads
                    (2 % substitutions, half reverse-complemented), -b -l 50
  config1_pair      BASELINE.json configs[0]: a 4.64 Mbp genome against a 1.5 %-diverged strain with three
                    inversions and two deletions (tests/golden_cases.py::ecoli_like_pair), -b -l 20
  config1_pair_mam  the same pair with -mam (sha256 of the reference's output file)
  config4_genome_like_first100k   248 Mbp text with the genome-like repeat load (10^5-copy family, satellite, 30 Mbp of N),
                    the first 100,000 reads, -b -l 50
  config1_dups_pair / config1_dups_pair_mam   the pair with exact repeats in the genome (ecoli_like_pair(duplicates=True)):
                    -mem and -mam print different files there
  config5_first100k BASELINE.json configs[4]: 3.1 Gbp text (> 2^31 rows) with the repeat model, the first 100,000 reads, -b -l 20
  config5_reads_7M  the same text, reads 7,000,000 .. 7,099,999 (a second, disjoint pin)
  config2_mam_first200k   the 100 Mbp reference of configs[1]/[2], its first 200,000 reads, -b -l 20 -mam

Every MEM the reference prints is also checked here against the texts (real match, maximal on both sides) before the
digest is recorded: the reference is known to print impossible MEMs on some texts with an LCP >= 255 (DESIGN.md 5,
B.11); a case that trips this is recorded with "reference_valid": false and the tests pin it on the oracle instead.
"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
from slamem_amd import synth  # noqa: E402
from mems_digest import digest_rows, parse  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "slaMEM")
OUT_JSON = os.path.join(HERE, "known_answers.json")

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


def check_rows_against_text(rows, ref, reads, strands):
    """rows: (block, ref1, query1, len) as printed.  reads: (count, L) array or a list of 1-D arrays.  Returns the
    number of rows that are not real two-sided-maximal matches."""
    bad = 0
    n = ref.shape[0]
    cache = {}
    for blk, r1, q1, ln in rows:
        blk, r, q, ln = int(blk), int(r1) - 1, int(q1) - 1, int(ln)
        k, rev = (blk // strands, blk % strands) if strands == 2 else (blk, 0)
        key = (k, rev)
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            rd = np.asarray(reads[k])
            cache[key] = _COMP[rd[::-1]] if rev else rd
        rd = cache[key]
        L = rd.shape[0]
        ok = r >= 0 and q >= 0 and r + ln <= n and q + ln <= L and bool((ref[r:r + ln] == rd[q:q + ln]).all())
        ok = ok and (r == 0 or q == 0 or ref[r - 1] != rd[q - 1])
        ok = ok and (r + ln == n or q + ln == L or ref[r + ln] != rd[q + ln])
        bad += not ok
    return bad


def run_reference(args, cwd, timeout=None, as_gb=None):
    """rc of the reference (or "timeout"), wall seconds.  as_gb caps the child's address space so that a run that
    needs more memory than the build container has fails by itself instead of taking the container down."""
    t0 = time.time()

    def limit():
        if as_gb:
            import resource
            resource.setrlimit(resource.RLIMIT_AS, (as_gb << 30, as_gb << 30))
    with open(os.path.join(cwd, "stdout.txt"), "wb") as so:
        try:
            rc = subprocess.run([REF_BIN] + args, cwd=cwd, stdout=so, timeout=timeout, preexec_fn=limit).returncode
        except subprocess.TimeoutExpired:
            rc = "timeout"
    return rc, time.time() - t0


def case_config4_first1M(tmp):
    n, nreads, L, min_len = 248_000_000, 1_000_000, 150, 50
    assert (n + 1) % 64 != 0
    ref = synth.make_reference(n, 42)
    planted = synth.plant_repeats(ref, 42)
    synth.write_fasta_reference(os.path.join(tmp, "ref.fa"), ref)
    reads = np.empty((nreads, L), dtype=np.uint8)
    with open(os.path.join(tmp, "qry.fa"), "wb") as f:
        step = 100_000
        for first in range(0, nreads, step):
            part = synth.make_reads(ref, first, step, L, 0.02, 42, 50)
            reads[first:first + step] = part
            f.write(b"".join(b">q%d\n" % (first + i) + part[i].tobytes() + b"\n" for i in range(step)))
    rc, secs = run_reference(["-b", "-l", str(min_len), "-o", "out.txt", "ref.fa", "qry.fa"], tmp)
    rows = parse(os.path.join(tmp, "out.txt"), True)
    bad = check_rows_against_text(rows, ref, reads, 2)
    d = digest_rows(rows)
    d.update({"reference_rc": rc, "reference_seconds": round(secs, 1), "reference_valid": bad == 0, "invalid_rows": bad,
              "workload": f"n={n} seed 42 + repeat model ({planted} planted letters), reads 0..{nreads - 1} of 150 bp, "
                          f"2% substitutions, 50% reverse-complemented, -b -l {min_len}"})
    return d


def case_config1_pair(tmp):
    from golden_cases import ecoli_like_pair
    ref, qry = ecoli_like_pair()
    synth.write_fasta_reference(os.path.join(tmp, "ref.fa"), ref, "ecoli_like_ref")
    synth.write_fasta_reference(os.path.join(tmp, "qry.fa"), qry, "ecoli_like_strain")
    rc, secs = run_reference(["-b", "-l", "20", "-o", "out.txt", "ref.fa", "qry.fa"], tmp)
    import hashlib
    data = open(os.path.join(tmp, "out.txt"), "rb").read()
    rows = []
    blk = -1
    for line in data.split(b"\n"):
        if line[:1] == b">":
            blk += 1
        elif line:
            a, b, c = line.split(b"\t")
            rows.append((blk, int(a), int(b), int(c)))
    rows = np.array(rows, dtype=np.uint32)
    bad = check_rows_against_text(rows, ref, [qry], 2)
    d = digest_rows(rows)
    d.update({"file_bytes": len(data), "file_sha256": hashlib.sha256(data).hexdigest(), "reference_rc": rc,
              "reference_seconds": round(secs, 1), "reference_valid": bad == 0, "invalid_rows": bad,
              "workload": f"ecoli_like_pair(): {ref.shape[0]} bp genome vs {qry.shape[0]} bp strain, -b -l 20"})
    return d


def case_config2_mam_first200k(tmp):
    """-mam at the headline workload's scale: the 100 Mbp reference of BASELINE.json configs[1]/[2], the first 200,000 of
    its reads, both strands"""
    n, nreads, L, min_len = 100_000_000, 200_000, 150, 20
    ref = synth.make_reference(n, 42)
    synth.write_fasta_reference(os.path.join(tmp, "ref.fa"), ref)
    reads = synth.make_reads(ref, 0, nreads, L, 0.02, 42, 50)
    with open(os.path.join(tmp, "qry.fa"), "wb") as f:
        f.write(b"".join(b">q%d\n" % i + reads[i].tobytes() + b"\n" for i in range(nreads)))
    rc, secs = run_reference(["-b", "-l", str(min_len), "-o", "out.txt", "ref.fa", "qry.fa", "-mam"], tmp)
    rows = parse(os.path.join(tmp, "out.txt"), True)
    bad = check_rows_against_text(rows, ref, reads, 2)
    d = digest_rows(rows)
    d.update({"reference_rc": rc, "reference_seconds": round(secs, 1), "reference_valid": bad == 0, "invalid_rows": bad,
              "workload": f"n={n} seed 42, reads 0..{nreads - 1} of 150 bp, 2% substitutions, 50% reverse-complemented, "
                          f"-b -l {min_len} -mam"})
    return d


def case_config1_pair_mam(tmp):
    from golden_cases import ecoli_like_pair
    import hashlib
    ref, qry = ecoli_like_pair()
    synth.write_fasta_reference(os.path.join(tmp, "ref.fa"), ref, "ecoli_like_ref")
    synth.write_fasta_reference(os.path.join(tmp, "qry.fa"), qry, "ecoli_like_strain")
    rc, secs = run_reference(["-b", "-l", "20", "-o", "out.txt", "ref.fa", "qry.fa", "-mam"], tmp)
    data = open(os.path.join(tmp, "out.txt"), "rb").read()
    return {"file_bytes": len(data), "file_sha256": hashlib.sha256(data).hexdigest(), "reference_rc": rc,
            "reference_seconds": round(secs, 1),
            "workload": f"ecoli_like_pair(): {ref.shape[0]} bp genome vs {qry.shape[0]} bp strain, -b -l 20 -mam"}


def case_config5_first100k(tmp, first=0):
    """BASELINE.json configs[4] stand-in: the 3.1 Gbp text (> 2^31 BWT rows) WITH the repeat model and 100,000 of
    its reads (first .. first+99,999), -b -l 20.  One core for hours and ~30 GB in the build container; a run that does not
    finish (time, memory, a crash of the reference at this size) is recorded as such, with the tail of its stdout."""
    n, nreads, L, min_len = 3_100_000_000, 100_000, 150, 20
    assert (n + 1) % 64 != 0
    ref = synth.make_reference(n, 42)
    planted = synth.plant_repeats(ref, 42)
    if not os.path.exists(os.path.join(tmp, "ref.fa")):
        synth.write_fasta_reference(os.path.join(tmp, "ref.fa"), ref)
    reads = synth.make_reads(ref, first, nreads, L, 0.02, 42, 50)
    with open(os.path.join(tmp, "qry.fa"), "wb") as f:
        f.write(b"".join(b">q%d\n" % (first + i) + reads[i].tobytes() + b"\n" for i in range(nreads)))
    workload = (f"n={n} seed 42 + repeat model ({planted} planted letters), reads {first}..{first + nreads - 1} of 150 bp, "
                f"2% substitutions, 50% reverse-complemented, -b -l {min_len}")
    hours = float(os.environ.get("REF_TIMEOUT_HOURS", "5"))
    if os.environ.get("REF_REUSE") == "1" and os.path.exists(os.path.join(tmp, "out.txt")):
        # (the run finished and this script stopped behind it: take its output file; seconds from the files' times)
        rc, secs = 0, os.path.getmtime(os.path.join(tmp, "out.txt")) - os.path.getmtime(os.path.join(tmp, "qry.fa"))
    else:
        rc, secs = run_reference(["-b", "-l", str(min_len), "-o", "out.txt", "ref.fa", "qry.fa"], tmp,
                                 timeout=hours * 3600, as_gb=52)
    if rc != 0:
        tail = open(os.path.join(tmp, "stdout.txt"), "rb").read()[-600:].decode("latin1")
        return {"reference_rc": rc, "reference_seconds": round(secs, 1), "reference_completed": False,
                "stdout_tail": tail, "workload": workload}
    rows = parse(os.path.join(tmp, "out.txt"), True)
    rows[:, 0] -= np.uint32(2 * first)  # (the queries carry their numbers in the share: blocks from 0 here)
    bad = check_rows_against_text(rows, ref, reads, 2)
    d = digest_rows(rows)
    d.update({"reference_rc": rc, "reference_seconds": round(secs, 1), "reference_completed": True,
              "reference_valid": bad == 0, "invalid_rows": bad,
              "rows_beyond_2p31": int((rows[:, 1].astype(np.int64) > (1 << 31)).sum()), "workload": workload})
    return d


def case_config5_reads_7M(tmp):
    """A second, disjoint pin of configs[4]: reads 7,000,000 .. 7,099,999 of the same 12.5 M-read share (block numbers in the
    digest are the file's: 0 .. 199,999)."""
    return case_config5_first100k(tmp, 7_000_000)


def case_config4_genome_like_first100k(tmp):
    """The chr1-sized text with the GENOME-LIKE repeat load on top of SURVEY's model (slamem_amd/synth.py::plant_genome_like:
    100,000 copies of a 300 bp family at 5-15 % divergence, a 171 bp x 10^4 satellite array; WITHOUT the model's 30 Mbp block
    of N, on which the reference's LCP restoration (lcparray.c:650-662) is quadratic and does not finish), the first 100,000
    reads, -b -l 50."""
    n, nreads, L, min_len = 248_000_000, 100_000, 150, 50
    ref = synth.make_reference(n, 42)
    synth.plant_repeats(ref, 42)
    lay = synth.plant_genome_like(ref, 42, n_block=False)
    synth.write_fasta_reference(os.path.join(tmp, "ref.fa"), ref)
    reads = synth.make_reads(ref, 0, nreads, L, 0.02, 42, 50)
    with open(os.path.join(tmp, "qry.fa"), "wb") as f:
        f.write(b"".join(b">q%d\n" % i + reads[i].tobytes() + b"\n" for i in range(nreads)))
    rc, secs = run_reference(["-b", "-l", str(min_len), "-o", "out.txt", "ref.fa", "qry.fa"], tmp, timeout=4 * 3600, as_gb=24)
    if rc != 0:
        tail = open(os.path.join(tmp, "stdout.txt"), "rb").read()[-600:].decode("latin1")
        return {"reference_rc": rc, "reference_seconds": round(secs, 1), "reference_completed": False, "stdout_tail": tail}
    rows = parse(os.path.join(tmp, "out.txt"), True)
    bad = check_rows_against_text(rows, ref, reads, 2)
    d = digest_rows(rows)
    d.update({"reference_rc": rc, "reference_seconds": round(secs, 1), "reference_valid": bad == 0, "invalid_rows": bad,
              "layout": lay,
              "workload": f"n={n} seed 42 + repeat model + genome-like load without its block of N, reads 0..{nreads - 1} of "
                          f"150 bp, 2% substitutions, 50% reverse-complemented, -b -l {min_len}"})
    return d


def _pair_dups(tmp, mam):
    from golden_cases import ecoli_like_pair
    import hashlib
    ref, qry = ecoli_like_pair(duplicates=True)
    synth.write_fasta_reference(os.path.join(tmp, "ref.fa"), ref, "ecoli_like_ref")
    synth.write_fasta_reference(os.path.join(tmp, "qry.fa"), qry, "ecoli_like_strain")
    rc, secs = run_reference(["-b", "-l", "20", "-o", "out.txt", "ref.fa", "qry.fa"] + (["-mam"] if mam else []), tmp)
    data = open(os.path.join(tmp, "out.txt"), "rb").read()
    lines = sum(1 for ln in data.split(b"\n") if ln and ln[:1] != b">")
    return {"file_bytes": len(data), "file_sha256": hashlib.sha256(data).hexdigest(), "lines": lines, "reference_rc": rc,
            "reference_seconds": round(secs, 1),
            "workload": f"ecoli_like_pair(duplicates=True): {ref.shape[0]} bp genome with 7 x 5000 bp and 20 x 1300 bp exact "
                        f"repeats vs {qry.shape[0]} bp strain, -b -l 20" + (" -mam" if mam else "")}


def case_config1_dups_pair(tmp):
    return _pair_dups(tmp, False)


def case_config1_dups_pair_image(tmp):
    """the reference's -v picture (slamem.c:354-452) of its own MEMs file for the pair with exact repeats"""
    import hashlib
    d = _pair_dups(tmp, False)
    rc, secs = run_reference(["-v", "out.txt", "ref.fa", "qry.fa"], tmp)
    img = open(os.path.join(tmp, "out.bmp"), "rb").read()
    os.remove(os.path.join(tmp, "out.bmp"))
    return {"mems_file_sha256": d["file_sha256"], "image_bytes": len(img), "image_sha256": hashlib.sha256(img).hexdigest(),
            "reference_rc": rc, "reference_seconds": round(secs, 1), "workload": d["workload"] + "; then -v out.txt ref.fa qry.fa"}


def case_config1_dups_pair_mam(tmp):
    return _pair_dups(tmp, True)


CASES = {"config1_dups_pair_image": case_config1_dups_pair_image, "config4_genome_like_first100k": case_config4_genome_like_first100k, "config1_dups_pair": case_config1_dups_pair, "config1_dups_pair_mam": case_config1_dups_pair_mam,
         "config5_first100k": case_config5_first100k, "config5_reads_7M": case_config5_reads_7M, "config4_first1M": case_config4_first1M, "config1_pair": case_config1_pair,
         "config2_mam_first200k": case_config2_mam_first200k, "config1_pair_mam": case_config1_pair_mam}


def main():
    case = sys.argv[1]
    tmp = os.path.join("/tmp", "known_answers", case)
    os.makedirs(tmp, exist_ok=True)
    d = CASES[case](tmp)
    known = json.load(open(OUT_JSON)) if os.path.exists(OUT_JSON) else {}
    known[case] = d
    with open(OUT_JSON, "w") as f:
        json.dump(known, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(d))
    if "--keep" not in sys.argv:
        for fn in ("ref.fa", "qry.fa", "out.txt"):
            try:
                os.remove(os.path.join(tmp, fn))
            except OSError:
                pass


if __name__ == "__main__":
    main()
