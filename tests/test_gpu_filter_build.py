"""The presence filter built by sorting (k_kfilter_keys + radix sort + k_kfilter_fill: one 64-byte store per line) must hold
exactly the bits of the direct form (k_kfilter_build: one atomic per word and text position; SLAMEM_KFILTER_ATOMIC=1) -- on a
random text, on one with long runs of one (k-2)-mer (satellite array, homopolymer: the runs longer than kFilterRun take the
atomic path inside the sorted form) and on one with N."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_CHILD = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from slamem_amd import engine
n = 3_000_000
rng = np.random.default_rng(7)
text = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=171)
text[500_000:500_000 + 171 * 3000] = np.tile(unit, 3000)          # satellite: every (k-2)-mer 3000 times
text[1_500_000:1_560_000] = ord("A")                               # homopolymer
text[2_000_000:2_050_000] = ord("N")
idx = engine.Index.build(text)
arena = idx.arena_view()
hdr = np.frombuffer(arena[:256].cpu().numpy().tobytes(), dtype=np.uint64)
off, lg = int(hdr[7]), int(np.frombuffer(arena[:256].cpu().numpy().tobytes(), dtype=np.uint32)[22])
filt = arena[off: off + (8 << lg)].cpu().numpy()
print(hashlib.sha256(filt.tobytes()).hexdigest(), int(np.unpackbits(filt).sum()), lg)
"""


def test_sorted_filter_build_equals_the_atomic_one(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for atomic in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", _CHILD, root], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, SLAMEM_KFILTER_ATOMIC=atomic), timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(r.stdout.decode().split())
    assert outs[0] == outs[1], outs
    assert int(outs[0][1]) > 1_000_000  # the filter does hold bits
