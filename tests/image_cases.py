"""Fixtures of the "-v" image tool (tests/golden/image/, made by the real reference: tests/golden/make_image_golden.py).

The FASTA inputs are not stored: the tool uses the records' names and lengths only, so both the generator and the tests write
them from the manifest with the same fixed letters."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
IMAGE_GOLDEN = os.path.join(HERE, "golden", "image")


def manifest():
    return json.load(open(os.path.join(IMAGE_GOLDEN, "manifest.json")))


def write_fasta(path, records):
    """records: [(name, length)]; the letters are a fixed pattern, 70 per line"""
    unit = "ACGTTGCAAGCT" * 6
    with open(path, "w") as f:
        for name, n in records:
            f.write(">%s\n" % name)
            body = (unit * (n // len(unit) + 1))[:n]
            f.write("".join(body[i:i + 70] + "\n" for i in range(0, n, 70)))


def write_inputs(case, entry, where):
    write_fasta(os.path.join(where, "ref.fa"), [tuple(r) for r in entry["reference"]])
    write_fasta(os.path.join(where, "q.fa"), [tuple(q) for q in entry["queries"]])
    with open(os.path.join(IMAGE_GOLDEN, case, "mems.txt"), "rb") as f, open(os.path.join(where, "mems.txt"), "wb") as g:
        g.write(f.read())
