#!/usr/bin/env python3
"""Calibration behind bench.py's `port_vs_reference_ratio` (build container only: needs oracle/_ref/slaMEM, the REAL reference
compiled by oracle/Makefile from /root/reference).  Same reference text, same reads, one core each:

    oracle port      oracle/liboracle.so, matching only (index build timed apart)
    real reference   oracle/_ref/slaMEM -b -l 20, matching = wall of the whole run - wall of a run with ONE read
                     (load + index build), stdout to a file (keep stdout off pipes when timing)

Writes profiles/r04_port_vs_reference.json; bench.py reads the ratio from there.
    tests/tools/calibrate_port_vs_reference.py [ref_len=100000000] [reads=200000]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from slamem_amd import synth
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
L, min_len = 150, 20
tmp = "/tmp/calibrate_port"
os.makedirs(tmp, exist_ok=True)
ref = synth.make_reference(n, 42)
reads = synth.make_reads(ref, 0, R, L, 0.02, 42, 50)
synth.write_fasta_reference(os.path.join(tmp, "ref.fa"), ref)
synth.write_fasta_reads(os.path.join(tmp, "qry.fa"), reads)
synth.write_fasta_reads(os.path.join(tmp, "one.fa"), reads[:1])
exe = os.path.join(ROOT, "oracle", "_ref", "slaMEM")


def run(q):
    t0 = time.time()
    with open(os.path.join(tmp, "stdout.txt"), "wb") as so:
        rc = subprocess.run([exe, "-b", "-l", str(min_len), "-o", "out.txt", "ref.fa", q], cwd=tmp, stdout=so).returncode
    assert rc == 0
    return time.time() - t0


t_one = run("one.fa")
t_all = run("qry.fa")
ref_mems = sum(1 for ln in open(os.path.join(tmp, "out.txt"), "rb") if ln[:1] != b">")
t0 = time.time()
idx = po.OracleIndex(ref.tobytes())
t_build = time.time() - t0
offsets = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
t0 = time.time()
mems, _ = idx.match_batch(reads.reshape(-1), offsets, min_len, True)
t_match = time.time() - t0
assert len(mems) == ref_mems, (len(mems), ref_mems)
out = {"ref_len": n, "reads": R, "read_len": L, "options": "-b -l 20", "mems": ref_mems,
       "reference_total_s": round(t_all, 2), "reference_load_and_build_s": round(t_one, 2),
       "reference_matching_s": round(t_all - t_one, 2), "reference_MEMs_per_s": ref_mems / (t_all - t_one),
       "port_build_s": round(t_build, 2), "port_matching_s": round(t_match, 2), "port_MEMs_per_s": ref_mems / t_match,
       "port_vs_reference_ratio": (ref_mems / t_match) / (ref_mems / (t_all - t_one)),
       "host": "build container, one core each (the other cores busy with two reference runs of the round)" if os.environ.get("CAL_BUSY") else "build container, one core each",
       "nproc": os.cpu_count()}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "r04_port_vs_reference.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print(json.dumps(out))
