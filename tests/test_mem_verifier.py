"""The definitional verifier (tests/mem_verifier.py) checked against the oracle and the brute-force MEM definition on small
cases, on the CPU (its text scan restated in numpy; the GPU scan kernel is checked against that restatement under
-m gpu in tests/test_gpu_verifier.py).  Semantics: SURVEY.md A.5 == what slamem.c:139-193 prints."""
import numpy as np
import pytest

import mem_verifier as mv
from oracle import pyoracle as po


def random_case(rng, alphabet, n, nreads, L, plant):
    text = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n)
    for _ in range(plant):  # planted repeats (exact and diverged) and N runs
        ln = int(rng.integers(30, 400))
        src, dst = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
        seg = text[src:src + ln].copy()
        mut = rng.random(ln) < 0.03
        seg[mut] = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=int(mut.sum()))
        text[dst:dst + ln] = seg
    reads = np.empty((nreads, L), dtype=np.uint8)
    for i in range(nreads):
        if rng.random() < 0.8:
            p = int(rng.integers(0, n - L + 1))
            rd = text[p:p + L].copy()
            mut = rng.random(L) < 0.04
            rd[mut] = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=int(mut.sum()))
            if rng.random() < 0.5:
                rd = mv._COMP[rd[::-1]]
        else:
            rd = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=L)
        reads[i] = rd
    reads[0] = text[:L]          # text-boundary maximality on both ends
    reads[1] = text[n - L:]
    return text, reads


def oracle_rows(text, reads, min_len, both):
    o = po.OracleIndex(text.tobytes())
    S, L = reads.shape
    offsets = np.arange(S + 1, dtype=np.uint64) * np.uint64(L)
    mems, bc = o.match_batch(reads.reshape(-1), offsets, min_len, both)
    blk = np.repeat(np.arange(bc.shape[0], dtype=np.int64), bc.astype(np.int64))
    rows = np.stack([blk, mems["ref_pos"].astype(np.int64) + 1, mems["query_pos"].astype(np.int64) + 1,
                     mems["length"].astype(np.int64)], axis=1)
    return rows.astype(np.uint32)


@pytest.mark.parametrize("alphabet,min_len,both,seed", [
    (b"ACGT", 12, True, 1), (b"ACGT", 20, True, 2), (b"ACGTN", 8, True, 3), (b"AC", 25, False, 4),
    (b"ACG", 30, True, 5), (b"ACGT", 21, True, 6), (b"ACGT", 22, False, 7), (b"ACGTN", 5, True, 8)])
def test_verifier_equals_oracle(alphabet, min_len, both, seed):
    rng = np.random.default_rng(seed)
    text, reads = random_case(rng, alphabet, int(rng.integers(3000, 9000)), 24, 120, 6)
    rows = oracle_rows(text, reads, min_len, both)
    sample = np.arange(reads.shape[0])
    res = mv.verify_sample(text, None, reads, sample, rows, min_len, both)
    assert res["missing_count"] == 0 and res["extra_count"] == 0 and res["engine_duplicates"] == 0, res
    assert res["definitional_mems"] == rows.shape[0] > 0


def test_verifier_equals_brute_force_and_sees_a_dropped_mem():
    rng = np.random.default_rng(11)
    text, reads = random_case(rng, b"ACGT", 5000, 8, 100, 5)
    min_len = 10
    strands = mv.strands_of(reads, True)
    keys, s, q, k = mv.window_keys(strands, min_len)
    hits = mv.scan_text_numpy(text, k, np.unique(keys))
    want = mv.definitional_mems(text, strands, min_len, hits, k)
    brute = []
    for b in range(strands.shape[0]):
        m = po.brute_force_mems(text.tobytes(), strands[b].tobytes(), min_len)
        brute += [(b, int(x["ref_pos"]), int(x["query_pos"]), int(x["length"])) for x in m]
    assert sorted(map(tuple, want.tolist())) == sorted(brute)
    # a sampled subset of the reads, with one MEM removed from and one altered in the "engine" rows, must be reported
    rows = oracle_rows(text, reads, min_len, True)
    sample = np.array([1, 4, 6])
    in_sample = np.isin(rows[:, 0] // 2, sample)
    victim = int(np.nonzero(in_sample)[0][3])
    res = mv.verify_sample(text, None, reads, sample, np.delete(rows, victim, axis=0), min_len, True)
    assert res["missing_count"] == 1 and res["extra_count"] == 0
    bent = rows.copy()
    bent[victim, 3] -= 1
    res = mv.verify_sample(text, None, reads, sample, bent, min_len, True)
    assert res["missing_count"] == 1 and res["extra_count"] == 1
