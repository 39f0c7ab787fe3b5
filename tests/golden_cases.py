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
