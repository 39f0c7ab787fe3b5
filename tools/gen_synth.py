#!/usr/bin/env python3
"""Write the synthetic reference / read FASTA files of SURVEY.md Appendix C.2 (bench / test support).

    tools/gen_synth.py <n> <nreads> <len> <sub> <seed> <rc_percent> <outdir>

Prints the sha256 of both files (the survey recorded f6bcf2e657079fdf… / 81ac1e6c4b633fa3… for
`100000000 1000000 150 0.02 42 0`)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from slamem_amd import synth  # noqa: E402


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    n, nreads, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    sub, seed, rc = float(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    out = sys.argv[7]
    os.makedirs(out, exist_ok=True)
    ref = synth.make_reference(n, seed)
    synth.write_fasta_reference(os.path.join(out, "ref.fa"), ref)
    with open(os.path.join(out, "qry.fa"), "wb") as f:
        step = 200_000
        for first in range(0, nreads, step):
            cnt = min(step, nreads - first)
            reads = synth.make_reads(ref, first, cnt, L, sub, seed, rc)
            names = np.char.add(np.char.add(">q", np.arange(first, first + cnt).astype(str)), "\n").astype("S")
            rows = [names[i] + reads[i].tobytes() + b"\n" for i in range(cnt)]
            f.write(b"".join(rows))
    print("ref.fa", sha(os.path.join(out, "ref.fa")))
    print("qry.fa", sha(os.path.join(out, "qry.fa")))


if __name__ == "__main__":
    main()
