#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REAL REFERENCE.

Run in the build container only (needs oracle/_ref/slaMEM, which oracle/Makefile
compiles from /root/reference where the sources lie):

    make -C oracle ref && python tests/golden/make_golden.py

For every case this writes <case>/ref.fa, <case>/q*.fa (inputs we generate
here), <case>/expected-mems.txt (the reference's own output file, byte for byte)
and an entry in manifest.json with the command-line options.  Fixtures are data:
no reference source text is stored.  Cases stay inside the reference's validity
domain (n >= 640, (n+1) % 32 != 0, at least one MEM when
there are several query records) and every kept output was additionally checked
against the brute-force MEM definition, because the reference emits garbage
triples on some inputs with LCP >= 255 (seen: a 293-long "MEM" in a 315-long
read); such cases are dropped with a note.
"""
import json
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402  (test infrastructure)

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "slaMEM")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rc(s):
    return "".join(COMP.get(c, c) for c in reversed(s))


def rand_text(rng, n, alpha):
    return [rng.choice(alpha) for _ in range(n)]


def fix_len(n):
    while (n + 1) % 32 == 0 or (n + 1) % 64 == 0:
        n += 1
    return n


def mutate(rng, s, p, alpha):
    return "".join(rng.choice(alpha) if rng.random() < p else c for c in s)


def wrap(s, w):
    return "\n".join(s[i:i + w] for i in range(0, len(s), w)) + "\n"


def case_random(rng, alpha, n, nq, l, both, repeats=0, nrun=0, max_repeat=200, max_q=300):
    n = fix_len(n)
    t = rand_text(rng, n, alpha)
    for _ in range(repeats):
        L = rng.randint(15, max_repeat)
        a, b = rng.randint(0, n - L), rng.randint(0, n - L)
        t[b:b + L] = t[a:a + L]
    if nrun:
        a = rng.randint(0, n - nrun)
        t[a:a + nrun] = "N" * nrun
    t = "".join(t)
    qs = []
    for k in range(nq):
        L = rng.randint(max(l, 25), max_q)
        if k % 4 != 3:
            a = rng.randint(0, n - L)
            q = mutate(rng, t[a:a + L], 0.03, alpha)
        else:
            q = "".join(rand_text(rng, L, alpha))
        if both and k % 2:
            q = rc(q)
        qs.append(q)
    return {"refs": [("ref", t)], "queries": [("q%d" % i, q) for i, q in enumerate(qs)],
            "opts": (["-b"] if both else []) + ["-l", str(l)]}


def build_cases():
    rng = random.Random(20261003)
    cases = {}
    cases["acgt_l20_fwd"] = case_random(rng, "ACGT", 3000, 6, 20, False, repeats=2)
    cases["acgt_l20_both"] = case_random(rng, "ACGT", 3000, 6, 20, True, repeats=3)
    cases["ac_l10_both"] = case_random(rng, "AC", 1500, 4, 10, True, repeats=2)
    cases["acg_l5_fwd"] = case_random(rng, "ACG", 900, 3, 5, False, repeats=1, max_q=80)
    cases["acgtn_l8_both"] = case_random(rng, "ACGTN", 2500, 5, 8, True, repeats=2, nrun=120)
    cases["acgt_l1_fwd"] = case_random(rng, "ACGT", 700, 2, 1, False, max_q=30)
    cases["acgt_l3_both"] = case_random(rng, "ACGT", 800, 2, 3, True, repeats=1, max_q=60)
    cases["acgt_l50_both"] = case_random(rng, "ACGT", 4000, 6, 50, True, repeats=3)
    cases["acgt_l2_nruns"] = case_random(rng, "ACGTN", 1200, 3, 2, False, nrun=200, max_q=40)
    cases["long_repeat_l20"] = case_random(rng, "ACGT", 5000, 5, 20, True, repeats=3, max_repeat=600)

    # reads with N against an N-free reference; an all-N read; reads from both text ends
    n = fix_len(2000)
    t = "".join(rand_text(rng, n, "ACGT"))
    qs = [t[:40], t[-40:], "N" * 30, t[100:130] + "N" + t[131:170], rc(t[500:560])]
    cases["root_fallback_text_ends"] = {"refs": [("ref", t)], "queries": [("q%d" % i, q) for i, q in enumerate(qs)],
                                        "opts": ["-b", "-l", "12"]}

    # multi-record reference: 4-column output, positions relative to the record
    recs = [("chrA some description", "".join(rand_text(rng, 900, "ACGT"))),
            ("chrB", "".join(rand_text(rng, 1100, "ACGT"))),
            ("chrC third", "".join(rand_text(rng, 705, "ACGT")))]
    tot = sum(len(s) for _, s in recs) + 2
    assert (tot + 1) % 32 and (tot + 1) % 64
    qs = [recs[0][1][100:220], recs[1][1][1000:1100] + recs[2][1][0:60], rc(recs[2][1][300:420]),
          recs[0][1][850:900] + recs[1][1][0:50]]
    qs_multi = qs
    cases["multi_record_ref"] = {"refs": recs, "queries": [("read%d extra words" % i, q) for i, q in enumerate(qs)],
                                 "opts": ["-b", "-l", "15"]}
    cases["multi_record_ref_r_filter"] = {"refs": recs, "queries": [("r%d" % i, q) for i, q in enumerate(qs)],
                                          "opts": ["-l", "15", "-r", "chrB"]}
    # -m drops a short scaffold in the middle of the reference AND the 100-bp read (it applies to queries too)
    recs_m = [recs[0], ("scaffold_tiny", "".join(rand_text(rng, 100, "ACGT"))), recs[1], recs[2]]
    cases["multi_record_ref_m_filter"] = {"refs": recs_m, "queries": [("r%d" % i, q) for i, q in enumerate(qs)],
                                          "opts": ["-l", "15", "-m", "110"]}

    # normalisation: lower case, IUPAC codes, digits, '*', '-', CRLF, wrapped lines; with and without -n
    n = fix_len(1500)
    t = "".join(rand_text(rng, n, "ACGT"))
    tl = list(t)
    for i in range(0, n, 97):
        tl[i] = rng.choice("RYKMSWryn")
    for i in range(5, n, 53):
        tl[i] = tl[i].lower()
    t_raw = "".join(tl)
    q_raw = t_raw[200:330] + "xx**--12" + t_raw[330:400]
    q2 = rc(t[700:820])
    messy = {"refs_raw": ">messy ref\r\n" + wrap(t_raw, 60).replace("\n", "\r\n"),
             "queries_raw": ">qa first\n" + wrap(q_raw, 50) + ">qb\n" + q2.lower() + "\n>empty\n\n>qc\n" + t[50:90] + "\n",
             }
    cases["normalise_default"] = dict(messy, opts=["-b", "-l", "10"])
    cases["normalise_dash_n"] = dict(messy, opts=["-b", "-l", "10", "-n"])

    # genome-vs-genome style: one long query record
    n = fix_len(6000)
    t = "".join(rand_text(rng, n, "ACGT"))
    q = mutate(rng, t[300:2500], 0.02, "ACGT") + rc(t[3000:4000]) + mutate(rng, t[4500:5800], 0.01, "ACGT")
    cases["long_single_query"] = {"refs": [("genomeA", t)], "queries": [("genomeB", q)], "opts": ["-b", "-l", "20"]}

    # -mam (slamem.c:131,657): the option has to come LAST on the reference's command line, because it swallows the
    # argument after it.  Repeat-rich references so that many positions have several rows; there is no
    # definition to compare with (the mode's output depends on a stale-interval quirk): these files pin the
    # restatement to what the reference prints.
    for nm, alpha, n, nq, l, both, reps in [("mam_acgt_l8_both", "ACGT", 2500, 6, 8, True, 6),
                                            ("mam_ac_l12_fwd", "AC", 1800, 5, 12, False, 4),
                                            ("mam_acg_l5_both", "ACG", 900, 4, 5, True, 3),
                                            ("mam_acgt_l20_both", "ACGT", 4000, 6, 20, True, 8)]:
        c = case_random(rng, alpha, n, nq, l, both, repeats=reps, max_repeat=150)
        c["tail"] = ["-mam"]
        cases[nm] = c
    c = {"refs": recs, "queries": [("r%d" % i, q) for i, q in enumerate(qs_multi)], "opts": ["-b", "-l", "12"], "tail": ["-mam"]}
    cases["mam_multi_record_ref"] = c
    return cases


def parse_blocks(path):
    blocks = []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            blocks.append([line[1:], []])
        elif line:
            blocks[-1][1].append(line)
    return blocks


def sane_vs_bruteforce(case, out_path, refs_text, queries):
    """Single-record, default-normalisation cases only: compare with the definition."""
    opts = case["opts"]
    l = int(opts[opts.index("-l") + 1])
    both = "-b" in opts
    blocks = parse_blocks(out_path)
    k = 0
    for name, q in queries:
        for strand in range(2 if both else 1):
            qq = q if strand == 0 else rc(q)
            bf = po.sorted_triples(po.brute_force_mems(refs_text.encode(), qq.encode(), l))
            got = sorted(tuple(int(x) for x in ln.split("\t")) for ln in blocks[k][1])
            exp = sorted((int(a) + 1, int(b) + 1, int(c)) for a, b, c in bf)
            if got != exp:
                return False
            k += 1
    return True


def utilities(d):
    """The hidden -s (sort a MEMs file) and -c (clean a FASTA file) tools of the reference, run on this case's files
    under fixed relative names (the names appear in the messages and in the cleaned file's header)."""
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        shutil.copy(os.path.join(d, "expected-mems.txt"), os.path.join(t, "x-mems.txt"))
        shutil.copy(os.path.join(d, "ref.fa"), os.path.join(t, "r.fa"))
        r = subprocess.run([REF_BIN, "-s", "x-mems.txt"], cwd=t, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           stdin=subprocess.DEVNULL)
        open(os.path.join(d, "sort-stdout.txt"), "wb").write(r.stdout + b"status %d\n" % r.returncode)
        if os.path.exists(os.path.join(t, "x-mems-sorted.txt")):
            shutil.copy(os.path.join(t, "x-mems-sorted.txt"), os.path.join(d, "sorted.txt"))
        r = subprocess.run([REF_BIN, "-c", "r.fa"], cwd=t, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        open(os.path.join(d, "clean-stdout.txt"), "wb").write(r.stdout + b"status %d\n" % r.returncode)
        shutil.copy(os.path.join(t, "r-clean.fasta"), os.path.join(d, "clean.fasta"))


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    manifest = {}
    for name, case in build_cases().items():
        d = os.path.join(HERE, name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        if "refs_raw" in case:
            open(os.path.join(d, "ref.fa"), "w", newline="").write(case["refs_raw"])
            open(os.path.join(d, "q.fa"), "w", newline="").write(case["queries_raw"])
        else:
            with open(os.path.join(d, "ref.fa"), "w") as f:
                for nm, s in case["refs"]:
                    f.write(">" + nm + "\n" + wrap(s, 70))
            with open(os.path.join(d, "q.fa"), "w") as f:
                for nm, s in case["queries"]:
                    f.write(">" + nm + "\n" + s + "\n")
        cmd = [REF_BIN] + case["opts"] + ["-o", "expected-mems.txt", "ref.fa", "q.fa"] + case.get("tail", [])
        r = subprocess.run(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            print("DROP %-28s reference exit status %d" % (name, r.returncode))
            shutil.rmtree(d)
            continue
        if "refs" in case and len(case["refs"]) == 1 and "tail" not in case:
            if not sane_vs_bruteforce(case, os.path.join(d, "expected-mems.txt"), case["refs"][0][1], case["queries"]):
                print("DROP %-28s reference output differs from the MEM definition" % name)
                shutil.rmtree(d)
                continue
        open(os.path.join(d, "expected-stdout.txt"), "wb").write(r.stdout)
        utilities(d)
        nm = sum(len(b[1]) for b in parse_blocks(os.path.join(d, "expected-mems.txt")))
        manifest[name] = {"opts": case["opts"], "ref": "ref.fa", "queries": ["q.fa"], "mems": nm}
        if "tail" in case:
            manifest[name]["tail"] = case["tail"]
        print("keep %-28s %5d MEMs" % (name, nm))
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
