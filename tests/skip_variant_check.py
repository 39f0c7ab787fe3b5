"""Run by tests/test_gpu_parity.py::test_skip_variant_is_exact in a child process with SLAMEM_SKIP=1 (the switch is read once
per process): the K8 instantiation with the skipping states (SKV / SKQ / SKP) against the oracle, in order, on reads with
substitutions over texts with and without repeats, and the full-size digest of the bench workload (the REAL reference's)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
assert os.environ.get("SLAMEM_SKIP") == "1"
import torch  # noqa: E402
from slamem_amd import engine, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from mems_digest import digest_rows  # noqa: E402

out = {}
for case, (n, nreads, L, sub, min_len, plant) in {"random": (3_000_000, 20000, 150, 0.02, 20, False),
                                                  "repeats": (2_000_000, 12000, 200, 0.03, 20, True),
                                                  "dense_subs": (1_000_000, 8000, 120, 0.08, 16, False),
                                                  "l30": (2_000_000, 8000, 250, 0.02, 30, True)}.items():
    ref = synth.make_reference(n, seed=5)
    if plant:
        synth.plant_repeats(ref, 5)
    reads = synth.make_reads(ref, 0, nreads, L, sub, seed=5, rc_percent=50).reshape(-1)
    off = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(L)
    idx = engine.Index.build(ref, "cuda:0")
    gm, goff = idx.find_mems(reads, off, min_len, True)
    m = idx.matcher(nreads, True, 8 * nreads, nreads * L)
    qd = torch.zeros(len(reads) + 16, dtype=torch.uint8, device="cuda:0")
    qd[: len(reads)] = torch.from_numpy(reads).to("cuda:0")
    st = engine.search_stats(m, qd, torch.from_numpy(off.view(np.int64)).to("cuda:0"), min_len)
    o = po.OracleIndex(ref.tobytes())
    om, obc = o.match_batch(reads, off, min_len, True)
    ok = np.array_equal(np.diff(goff.astype(np.int64)), obc.astype(np.int64)) and len(gm) == len(om) and all(
        np.array_equal(gm[f], om[f]) for f in ("ref_pos", "query_pos", "length"))
    out[case] = {"equal_in_order": bool(ok), "mems": int(len(om)), "skips": st["skips"], "skip_attempts": st["skip_attempts"]}
    idx.close()
# the bench workload at full size: digest recorded from the REAL reference
known = json.load(open(os.path.join(ROOT, "tests", "golden", "config3_known_answer.json")))
n, nreads, L = 100_000_000, 10_000_000, 150
ref = engine.synth_reference(n, 42, "cuda:0")
reads = engine.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)
offsets = torch.arange(nreads + 1, dtype=torch.int64, device="cuda:0") * L
idx = engine.Index.build(ref, "cuda:0")
m = idx.matcher(nreads, True, 4 * nreads, nreads * L)
total = m.run(reads, offsets, 20)
mems = m.mems[:total].cpu().numpy().view(np.uint32)
boff = m.block_offsets[: 2 * nreads + 1].cpu().numpy()
rows = np.empty((total, 4), dtype=np.uint32)
rows[:, 0] = np.repeat(np.arange(2 * nreads, dtype=np.uint32), np.diff(boff))
rows[:, 1] = mems[:, 0] + 1
rows[:, 2] = mems[:, 1] + 1
rows[:, 3] = mems[:, 2]
got = digest_rows(rows)
out["config3_digest_equal"] = got == {k: known[k] for k in ("mems", "sum_len", "max_len", "sha256")}
print(json.dumps(out))
sys.exit(0 if all(v["equal_in_order"] for k, v in out.items() if isinstance(v, dict)) and out["config3_digest_equal"] else 1)
