"""The golden fixtures under tests/golden/ (made by the real reference; see tests/golden/make_golden.py)."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
CASES = sorted(MANIFEST)


def opt_value(opts, flag, default=None):
    return opts[opts.index(flag) + 1] if flag in opts else default


def case_paths(name):
    d = os.path.join(GOLDEN, name)
    return os.path.join(d, "ref.fa"), os.path.join(d, "q.fa"), os.path.join(d, "expected-mems.txt"), \
        os.path.join(d, "expected-stdout.txt")


def ecoli_like_pair(duplicates: bool = False):
    """duplicates=True: the genome additionally carries exact repeats the way a real one does -- seven copies of a
    5,000 bp element (rRNA operons) and twenty copies of a 1,300 bp element (insertion sequences), planted BEFORE the strain is
    derived -- so that matches inside them occupy several BWT rows and -mam (unique in the reference, slamem.c:131) prints
    a different file than -mem.

    BASELINE.json configs[0] stand-in (the real E. coli FASTAs are not available offline):
    a 4,641,652 bp random genome and a 'strain' of it with 1.5 % substitutions, three 40 kbp inversions and two
    deletions.  Fully determined by splitmix64 streams (no library RNG), so the build container and the GPU box make
    the same bytes.  Returns (reference, query) uint8 ASCII arrays."""
    import numpy as np
    from slamem_amd import synth
    n = 4_641_652
    ref = synth.make_reference(n, seed=11)
    if duplicates:
        d = synth.splitmix64_at(0xD0B1E5, np.arange(64, dtype=np.uint64))
        at = 0
        for copies, ln in ((7, 5000), (20, 1300)):
            src = int(d[at] % np.uint64(n - ln)); at += 1
            elem = ref[src:src + ln].copy()
            for _ in range(copies - 1):
                dst = int(d[at] % np.uint64(n - ln)); at += 1
                ref[dst:dst + ln] = elem
    q = ref.copy()
    x = synth.splitmix64_at(0xEC011, np.arange(n, dtype=np.uint64))
    mut = (x & np.uint64(0xFFFFFFFF)) < np.uint64(int(0.015 * 4294967296.0))
    code = np.zeros(n, dtype=np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ref == ch] = i
    alt = np.frombuffer(b"CGTAGTACTACG", dtype=np.uint8).reshape(4, 3)
    q[mut] = alt[code[mut], ((x[mut] >> np.uint64(32)) % np.uint64(3)).astype(np.int64)]
    comp = np.arange(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    for a in (500_000, 2_000_000, 3_500_000):
        q[a:a + 40_000] = comp[q[a:a + 40_000][::-1]]
    q = np.delete(q, np.r_[1_000_000:1_040_000, 3_000_000:3_044_046])
    return ref, q
