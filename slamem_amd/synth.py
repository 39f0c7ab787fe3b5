"""Synthetic reference / read generator (SURVEY.md Appendix C.2 spec).

Counter-based splitmix64: the k-th draw of a stream seeded with ``seed`` is
``mix(seed + (k+1) * GAMMA)``, so every draw can be computed independently
(numpy here, one thread per draw in ``csrc/synth.hip`` on the GPU).

Draw layout (0-based draw index ``k``):

* reference base ``i``            -> draw ``i``                  (``"ACGT"[x & 3]``)
* read ``r`` start position       -> draw ``n + r*(L+2)``        (``x % (n-L+1)``)
* read ``r`` base ``i``           -> draw ``n + r*(L+2) + 1 + i`` (substitute when
  ``(u32)x < (u32)(sub * 2**32)``; replacement ``alt(c)[(x >> 32) % 3]``)
* read ``r`` strand flip          -> draw ``n + r*(L+2) + 1 + L`` (``x % 100 < rc_percent``)

This is test / bench infrastructure, not part of the MEM engine.
"""
from __future__ import annotations

import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

_ALT = np.frombuffer(b"CGTAGTACTACG", dtype=np.uint8).reshape(4, 3)  # alt(A),alt(C),alt(G),alt(T)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    _COMP[a] = b


def splitmix64_at(seed: int, k: np.ndarray) -> np.ndarray:
    """Value of the k-th (0-based) ``next()`` call of a splitmix64 seeded with ``seed``."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (k.astype(np.uint64) + np.uint64(1)) * GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def make_reference(n: int, seed: int = 42, chunk: int = 1 << 24) -> np.ndarray:
    """n i.i.d. uniform A/C/G/T bytes (uint8 ASCII)."""
    out = np.empty(n, dtype=np.uint8)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = splitmix64_at(seed, np.arange(s, e, dtype=np.uint64))
        out[s:e] = _ACGT[(x & np.uint64(3)).astype(np.int64)]
    return out


def make_reads(ref: np.ndarray, first: int, count: int, length: int = 150, sub: float = 0.02,
               seed: int = 42, rc_percent: int = 0, avoid: tuple = (0, 0)) -> np.ndarray:
    """Reads ``first .. first+count-1`` as a (count, length) uint8 array.  avoid = (at, len): a read that would touch
    text[at, at+len) (the block of N of the genome-like model: sequencers do not read it) is drawn at its start + len + length
    instead."""
    n = ref.shape[0]
    L = length
    thr = np.uint64(int(sub * 4294967296.0) & 0xFFFFFFFF)
    r = np.arange(first, first + count, dtype=np.uint64)
    base = np.uint64(n) + r * np.uint64(L + 2)
    p = splitmix64_at(seed, base) % np.uint64(n - L + 1)
    if avoid[1]:
        hit = (p + np.uint64(L) > np.uint64(avoid[0])) & (p < np.uint64(avoid[0] + avoid[1]))
        p = np.where(hit, p + np.uint64(avoid[1] + L), p)
    idx = p[:, None].astype(np.int64) + np.arange(L, dtype=np.int64)[None, :]
    c = ref[idx]
    x = splitmix64_at(seed, base[:, None] + np.uint64(1) + np.arange(L, dtype=np.uint64)[None, :])
    do_sub = (x & np.uint64(0xFFFFFFFF)) < thr
    code = np.zeros_like(c)
    code[c == ord("C")] = 1
    code[c == ord("G")] = 2
    code[c == ord("T")] = 3
    alt = _ALT[code.astype(np.int64), ((x >> np.uint64(32)) % np.uint64(3)).astype(np.int64)]
    c = np.where(do_sub, alt, c)
    flip = (splitmix64_at(seed, base + np.uint64(1 + L)) % np.uint64(100)) < np.uint64(rc_percent)
    if flip.any():
        c[flip] = _COMP[c[flip][:, ::-1]]
    return np.ascontiguousarray(c)


REPEAT_SALT = 0x7265706561747321


def repeat_segments(n: int, seed: int = 42):
    """Segments (k, length, src, dst) of the repeat model of SURVEY.md 8(d) -- see csrc/synth.hip: 0.5 % of the text
    overwritten by copies of 1-10 kbp segments with 1 % substitutions, fully determined by (n, seed)."""
    if n < 20000:
        raise ValueError("repeat model needs n >= 20000")
    segs, planted, k = [], 0, 0
    sp = (seed + REPEAT_SALT) & 0xFFFFFFFFFFFFFFFF
    while planted < n // 200:
        d = splitmix64_at(sp, np.arange(3 * k, 3 * k + 3, dtype=np.uint64))
        ln = 1000 + int(d[0] % np.uint64(9001))
        src = int(d[1] % np.uint64(n - ln + 1))
        dst = int(d[2] % np.uint64(n - ln + 1))
        segs.append((k, ln, src, dst))
        planted += ln
        k += 1
    return segs


def plant_repeats(ref: np.ndarray, seed: int = 42) -> int:
    """Apply the repeat model in place to a text made by make_reference(n, seed); returns the planted letters."""
    n = ref.shape[0]
    thr = np.uint64(int(0.01 * 4294967296.0))
    sm = (seed + REPEAT_SALT + 1) & 0xFFFFFFFFFFFFFFFF
    planted = 0
    for k, ln, src, dst in repeat_segments(n, seed):
        code = (splitmix64_at(seed, np.arange(src, src + ln, dtype=np.uint64)) & np.uint64(3)).astype(np.int64)
        x = splitmix64_at(sm, np.uint64(k * 16384) + np.arange(ln, dtype=np.uint64))
        alt = _ALT[code, ((x >> np.uint64(32)) % np.uint64(3)).astype(np.int64)]
        ref[dst:dst + ln] = np.where((x & np.uint64(0xFFFFFFFF)) < thr, alt, _ACGT[code])
        planted += ln
    return planted


# ---- "genome-like" repeat load (round 3; an extra model beside SURVEY.md 8(d)'s, same values as csrc/synth.hip) -----------
# What real chromosomes add to the mild model above: ONE interspersed family of n // 2480 copies (10^5 at 248 Mbp, 1.25 M at
# 3.1 Gbp) of a 300 bp consensus, each copy 5-15 % diverged from it; a satellite array of 10^4 units of 171 bp, each unit 2 %
# diverged from the unit consensus; and a block of N of min(n // 8, 30 M) letters.  Copies never overlap (copy k lies in
# the k-th stretch of n // copies letters), so the text does not depend on the order of writes.
GENOME_SALT = 0x67656E6F6D652121
FAMILY_LEN, SAT_UNIT, SAT_COPIES = 300, 171, 10_000


def genome_like_layout(n: int):
    copies = n // 2480
    stride = n // copies
    sat_at = n // 3
    n_len = min(n // 8, 30_000_000)
    n_at = n // 2
    return copies, stride, sat_at, n_at, n_len


def plant_genome_like(ref: np.ndarray, seed: int = 42, n_block: bool = True) -> dict:
    """Apply the genome-like model in place (after plant_repeats, if both are wanted).  Returns its layout.
    n_block=False leaves the block of N out: the REAL reference restores LCP values >= 255 by comparing the text letter by
    letter (lcparray.c:650-662), which is quadratic in the length of a run of N -- a 30 Mbp block never finishes there (30
    CPU-minutes in that loop before the round-3 run was stopped), so the reference-pinned case is made without it."""
    n = ref.shape[0]
    if n < 10_000_000:
        raise ValueError("genome-like model needs n >= 10^7")
    copies, stride, sat_at, n_at, n_len = genome_like_layout(n)
    sf = (seed + GENOME_SALT) & 0xFFFFFFFFFFFFFFFF
    cons = (splitmix64_at(sf, np.arange(FAMILY_LEN, dtype=np.uint64)) & np.uint64(3)).astype(np.int64)
    step = 1 << 16
    for k0 in range(0, copies, step):
        k = np.arange(k0, min(copies, k0 + step), dtype=np.uint64)
        dst = k * np.uint64(stride) + splitmix64_at(sf + 1, k) % np.uint64(stride - FAMILY_LEN)
        thr = ((np.uint64(500) + splitmix64_at(sf + 2, k) % np.uint64(1001)) * np.uint64(429497)).astype(np.uint64)  # (5..15 %) * 2^32
        x = splitmix64_at(sf + 3, k[:, None] * np.uint64(512) + np.arange(FAMILY_LEN, dtype=np.uint64)[None, :])
        sub = (x & np.uint64(0xFFFFFFFF)) < thr[:, None]
        alt = _ALT[cons[None, :].repeat(k.shape[0], 0), ((x >> np.uint64(32)) % np.uint64(3)).astype(np.int64)]
        letters = np.where(sub, alt, _ACGT[cons][None, :])
        idx = dst[:, None].astype(np.int64) + np.arange(FAMILY_LEN, dtype=np.int64)[None, :]
        ref[idx] = letters
    unit = (splitmix64_at(sf + 4, np.arange(SAT_UNIT, dtype=np.uint64)) & np.uint64(3)).astype(np.int64)
    tot = SAT_UNIT * SAT_COPIES
    i = np.arange(tot, dtype=np.uint64)
    x = splitmix64_at(sf + 5, i)
    code = unit[(i % np.uint64(SAT_UNIT)).astype(np.int64)]
    sub = (x & np.uint64(0xFFFFFFFF)) < np.uint64(int(0.02 * 4294967296.0))
    ref[sat_at:sat_at + tot] = np.where(sub, _ALT[code, ((x >> np.uint64(32)) % np.uint64(3)).astype(np.int64)], _ACGT[code])
    if n_block:
        ref[n_at:n_at + n_len] = ord("N")
    else:
        n_len = 0
    return {"family_copies": copies, "family_stride": stride, "satellite_at": sat_at, "satellite_letters": tot,
            "n_block_at": n_at, "n_block_letters": n_len}


def write_fasta_reference(path: str, ref: np.ndarray, name: str | None = None, width: int = 80) -> None:
    n = ref.shape[0]
    name = name or f"synthetic_ref_{n}"
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        full = (n // width) * width
        if full:
            body = np.empty((n // width, width + 1), dtype=np.uint8)
            body[:, :width] = ref[:full].reshape(-1, width)
            body[:, width] = 10
            f.write(body.tobytes())
        if n > full:
            f.write(ref[full:].tobytes() + b"\n")


def write_fasta_reads(path: str, reads: np.ndarray, first: int = 0) -> None:
    with open(path, "wb") as f:
        for k in range(reads.shape[0]):
            f.write(b">q%d\n" % (first + k))
            f.write(reads[k].tobytes())
            f.write(b"\n")
