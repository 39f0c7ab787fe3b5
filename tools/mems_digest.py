#!/usr/bin/env python3
"""Digest of a *-mems.txt file written for the synthetic read sets (records named q<k>): number of MEMs, sum and
maximum of their lengths, and sha256 over the uint32 little-endian rows (block, ref, query, length) sorted
lexicographically, where block = 2*k + strand (as printed: 1-based positions).  The same digest is computed from the
device result in tests/test_gpu_parity.py::test_config3_known_answer_full_size, so a run of the REAL reference
(oracle/_ref/slaMEM, build container only) pins the full-size bench workload.

    tools/mems_digest.py <mems file> [both_strands=1]
"""
import hashlib
import json
import sys

import numpy as np


def digest_rows(rows: np.ndarray) -> dict:
    """rows: int64/uint32 [N,4] = (block, ref, query, length)"""
    rows = np.ascontiguousarray(rows, dtype=np.uint32)
    order = np.lexsort((rows[:, 3], rows[:, 2], rows[:, 1], rows[:, 0]))
    rows = rows[order]
    return {"mems": int(rows.shape[0]), "sum_len": int(rows[:, 3].astype(np.int64).sum()),
            "max_len": int(rows[:, 3].max()) if rows.shape[0] else 0,
            "sha256": hashlib.sha256(rows.tobytes()).hexdigest()}


def parse(path: str, both: bool) -> np.ndarray:
    out = []
    block = -1
    strands = 2 if both else 1
    chunk = []
    with open(path, "rb") as f:
        for line in f:
            if line[:1] == b">":
                name = line[1:].split()
                k = int(name[0][1:])
                rev = len(name) > 1 and name[1] == b"Reverse"
                block = k * strands + (1 if rev else 0)
            else:
                a, b, c = line.split(b"\t")
                chunk.append((block, int(a), int(b), int(c)))
                if len(chunk) >= 1 << 20:
                    out.append(np.array(chunk, dtype=np.uint32))
                    chunk = []
    if chunk:
        out.append(np.array(chunk, dtype=np.uint32))
    return np.concatenate(out) if out else np.zeros((0, 4), dtype=np.uint32)


if __name__ == "__main__":
    both = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
    print(json.dumps(digest_rows(parse(sys.argv[1], both))))
