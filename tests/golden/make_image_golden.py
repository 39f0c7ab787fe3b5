#!/usr/bin/env python3
"""Fixtures for the "-v <mems_file>" image tool (slamem.c:354-452), made by RUNNING THE REAL REFERENCE.

Build container only (needs oracle/_ref/slaMEM):

    make -C oracle ref && python tests/golden/make_image_golden.py

Writes tests/golden/image/<case>/{mems.txt (input made here, seeded), expected.bmp, expected-stdout.txt (the reference's own
output, byte for byte)} and tests/golden/image/manifest.json (names and lengths of the FASTA records: the tool uses nothing else
of them, so tests/image_cases.py writes the FASTA files again from the manifest instead of storing them).  Fixtures are data;
no reference source text is stored.  Cases stay where the reference's output is defined: no stretch of 255 neighbouring columns without two equal ones in
a row of the picture (there the reference's run-length coder skips bytes and ends by reading behind its pixel buffer, so two runs
of the reference itself differ) -- that regime is covered by tests/test_image_tool.py's decoder-based check instead.
"""
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "slaMEM")
OUT = os.path.join(HERE, "image")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_cases  # noqa: E402


def mems_file(path, rng, queries, ref_len, mode):
    """queries: [(name, length)]"""
    with open(path, "w") as f:
        for name, qlen in queries:
            q = range(qlen)
            for strand in (0, 1):
                f.write(">%s%s\n" % (name, " Reverse" if strand else ""))
                if mode == "noise":  # about one short MEM per column of the picture
                    step = max(1, len(q) // 1000)
                    for qp in range(1, len(q) + 1, step):
                        ln = max(1, min(rng.randint(1, step), len(q) - qp + 1, ref_len))
                        f.write("%d\t%d\t%d\n" % (rng.randint(1, ref_len - ln + 1), qp, ln))
                    continue
                count = {"few": rng.randint(0, 5), "diag": rng.randint(20, 200), "scatter": rng.randint(50, 400)}[mode]
                for _ in range(count):
                    ln = max(1, min(rng.randint(1, max(1, min(len(q), ref_len) // rng.choice([1, 3, 10, 100]))), len(q), ref_len))
                    qp = rng.randint(1, len(q) - ln + 1)
                    rp = min(max(1, qp + rng.randint(-5, 5)), ref_len - ln + 1) if mode == "diag" else rng.randint(1, ref_len - ln + 1)
                    f.write("%d\t%d\t%d\n" % (rp, qp, ln))


def run_reference(case, entry, args):
    """the reference in a scratch directory; its stdout and picture go to the case's directory"""
    d = os.path.join(OUT, case)
    with tempfile.TemporaryDirectory() as tmp:
        image_cases.write_inputs(case, entry, tmp)
        r = subprocess.run([REF_BIN] + args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=False)
        with open(os.path.join(d, "expected-stdout.txt"), "wb") as f:
            f.write(r.stdout)
        made = os.path.exists(os.path.join(tmp, "mems.bmp"))
        if made:
            shutil.copy(os.path.join(tmp, "mems.bmp"), os.path.join(d, "expected.bmp"))
    return r.returncode, made


CASES = {
    # name: (seed, reference length, reference name, [(query name, length)], kind of MEMs file, extra options)
    "few_mems_one_query": (1, 5000, "ref", [("q0", 5000)], "few", []),
    "diagonals_three_queries": (2, 123456, "gi|123|some reference sequence, complete genome",
                                [("q0", 60000), ("read_1 some description here that is long", 123456), ("we|rd~{}[]2", 400)], "diag", []),
    "scatter_query_longer_than_reference": (3, 2500, "R" * 200, [("q0", 5000), ("Reverse1", 1234), ("x" * 120 + "2", 50)], "scatter", []),
    "noise_two_queries": (4, 1000000, "ref", [("q0", 1000000), ("q1", 333333)], "noise", []),
    "nine_small_queries": (5, 300, "ref", [("q%d" % k, n) for k, n in enumerate([50, 400, 150, 300, 600, 1234, 50, 77, 299])], "few", []),
}
ERRORS = {
    # the MEMs file's own problems: name: text of the file (one reference and one query of 5,000 letters)
    "error_unknown_name": ">nobody\n1\t1\t20\n",
    "error_zero_value": ">q0\n0\t1\t20\n",
    "error_not_numbers": ">q0\n1\t1\ttwenty\n",
    "error_four_fields": ">q0\n ref\t1\t1\t20\n",
    "error_second_unknown": ">q0\n1\t1\t20\n>q0 Reverse\n>q9\n5\t5\t5\n",
}


def main():
    assert os.path.exists(REF_BIN), "make -C oracle ref first"
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    manifest = {}

    def add(name, reference, queries, text_or_none, opts, seed=0, mode=None):
        d = os.path.join(OUT, name)
        os.makedirs(d)
        entry = {"reference": reference, "queries": queries, "opts": opts}
        if text_or_none is None:
            mems_file(os.path.join(d, "mems.txt"), random.Random(seed), queries, sum(n for _, n in reference), mode)
        else:
            with open(os.path.join(d, "mems.txt"), "w") as f:
                f.write(text_or_none)
        entry["rc"], entry["image"] = run_reference(name, entry, opts + ["-v", "mems.txt", "ref.fa", "q.fa"])
        manifest[name] = entry

    for name, (seed, rn, rname, qs, mode, opts) in CASES.items():
        add(name, [[rname, rn]], [list(q) for q in qs], None, opts, seed, mode)
        assert manifest[name]["image"] and manifest[name]["rc"] == 0
    for name, text in ERRORS.items():
        add(name, [["ref", 5000]], [["q0", 5000]], text, [])
        assert not manifest[name]["image"]
    # a reference file of two records: refused (slamem.c:362-366)
    add("error_two_reference_records", [["a", 700], ["b", 800]], [["q0", 100]], ">q0\n1\t1\t20\n", [])
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    for name in sorted(manifest):
        d = os.path.join(OUT, name)
        print(name, manifest[name]["rc"], manifest[name]["image"], sum(os.path.getsize(os.path.join(d, x)) for x in os.listdir(d)))


if __name__ == "__main__":
    main()
