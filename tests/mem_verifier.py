"""Index-independent definitional MEM verifier -- TEST INFRASTRUCTURE ONLY (never imported by slamem_amd/).

What it computes: for a SAMPLE of fixed-length strands, the full set of SURVEY.md A.5 -- every (r, q, len >= l) with
T[r..r+len) == Q[q..q+len) that cannot be extended to the left or to the right -- from the text and the strands alone:
no suffix array, no BWT, no LCP, nothing of the engine's index.  That is what the reference prints for a strand
(slamem.c:139-193: every row of the interval and of every ancestor >= l deep), so comparing the engine's output for the
same strands with this set, as a set, pins COMPLETENESS (no MEM missing) and soundness at sizes where neither the oracle
nor the reference can run (3.1 Gbp text, > 2^31 BWT rows).

How: the first k = min(l, 21) letters of every l-letter window of the strands become exact keys of 3 bits per letter
(A,C,G,T = 0..3, anything else = N = 4; N equals N as in the reference).  The whole text is streamed ONCE,
every position whose k-mer is a key is a hit (tests/verifier/mem_verifier.hip on the GPU; `scan_text_numpy` restates the
scan on the CPU for the small `-m "not gpu"` cases that check this verifier against the oracle).  The hits are joined
with the windows on the host; a MEM has exactly one seed at its own start, and that seed is left-maximal, so only the
left-maximal seeds are kept and extended to the right against the text.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "verifier", "libmem_verifier.so")
MAX_K = 21

_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _ch in enumerate(b"ACGT"):
    _CODE[_ch] = _i
    _CODE[_ch | 0x20] = _i
_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCATGCA"):
    _COMP[_a] = _b


def strands_of(reads: np.ndarray, both: bool) -> np.ndarray:
    """reads: (S, L) uint8 -> strands (S * (2 if both else 1), L); strand block b = 2 * i + strand (include/slamem_hip.h);
    the reverse strand is the reverse complement with N unchanged (ReverseComplementSequence, sequence.c:413)."""
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    if not both:
        return reads
    out = np.empty((2 * reads.shape[0], reads.shape[1]), dtype=np.uint8)
    out[0::2] = reads
    out[1::2] = _COMP[reads[:, ::-1]]
    return out


def window_keys(strands: np.ndarray, min_len: int):
    """(keys, strand index, window start) of every min_len-letter window of every strand; key = first k letters."""
    S, L = strands.shape
    k = min(min_len, MAX_K)
    W = L - min_len + 1
    if W <= 0 or S == 0:
        z = np.zeros(0, dtype=np.uint64)
        return z, z.astype(np.int64), z.astype(np.int64), k
    codes = _CODE[strands].astype(np.uint64)
    key = np.zeros((S, W), dtype=np.uint64)
    for i in range(k):
        key = (key << np.uint64(3)) | codes[:, i:i + W]
    s = np.repeat(np.arange(S, dtype=np.int64), W)
    q = np.tile(np.arange(W, dtype=np.int64), S)
    return key.reshape(-1), s, q, k


def keys_at(text: np.ndarray, pos: np.ndarray, k: int, chunk: int = 1 << 20) -> np.ndarray:
    """Key of text[p..p+k) for every p (all p + k <= len(text))."""
    out = np.empty(pos.shape[0], dtype=np.uint64)
    ar = np.arange(k, dtype=np.int64)
    for a in range(0, pos.shape[0], chunk):
        p = pos[a:a + chunk].astype(np.int64)
        c = _CODE[text[p[:, None] + ar[None, :]]].astype(np.uint64)
        key = np.zeros(p.shape[0], dtype=np.uint64)
        for i in range(k):
            key = (key << np.uint64(3)) | c[:, i]
        out[a:a + chunk] = key
    return out


def scan_text_numpy(text: np.ndarray, k: int, keys_unique: np.ndarray) -> np.ndarray:
    """CPU restatement of k_scan_text for small texts: positions r with r + k <= n whose k-mer is one of the keys."""
    n = text.shape[0]
    if n < k:
        return np.zeros(0, dtype=np.uint64)
    codes = _CODE[text].astype(np.uint64)
    W = n - k + 1
    key = np.zeros(W, dtype=np.uint64)
    for i in range(k):
        key = (key << np.uint64(3)) | codes[i:i + W]
    return np.nonzero(np.isin(key, keys_unique))[0].astype(np.uint64)


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.memv_table_insert.restype = C.c_int
        L.memv_table_insert.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
        L.memv_scan_text.restype = C.c_int
        L.memv_scan_text.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def scan_text_gpu(text_dev, k: int, keys_unique: np.ndarray, cap: int = 1 << 26) -> np.ndarray:
    """text_dev: torch uint8 tensor on the GPU (the text itself, not the index).  Returns the sorted hit positions."""
    import torch
    L = _lib()
    dev = text_dev.device
    n = text_dev.numel()
    slots = 1 << max(16, int(np.ceil(np.log2(max(1, keys_unique.shape[0]) * 8))))
    table = torch.full((slots,), -1, dtype=torch.int64, device=dev)
    kd = torch.from_numpy(keys_unique.view(np.int64).copy()).to(dev)
    hits = torch.empty(cap, dtype=torch.int64, device=dev)
    nh = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    rc = L.memv_table_insert(kd.data_ptr(), kd.numel(), table.data_ptr(), slots, st)
    assert rc == 0, f"memv_table_insert: {rc}"
    rc = L.memv_scan_text(text_dev.data_ptr(), n, k, table.data_ptr(), slots, hits.data_ptr(), cap, nh.data_ptr(), st)
    assert rc == 0, f"memv_scan_text: {rc}"
    torch.cuda.synchronize(dev)
    count = int(nh.item())
    assert count <= cap, f"verifier hit list too small: {count} > {cap} (sample fewer strands)"
    return np.sort(hits[:count].cpu().numpy().view(np.uint64))


def definitional_mems(text: np.ndarray, strands: np.ndarray, min_len: int, hits: np.ndarray, k: int,
                      wkeys=None, chunk: int = 1 << 21) -> np.ndarray:
    """rows (strand, r, q, len), 0-based, sorted and unique: the definitional MEM set of the strands.
    text: host uint8 array; hits: text positions whose k-mer is a window key (scan_text_*)."""
    n = text.shape[0]
    S, L = strands.shape
    keys, ws, wq, k2 = wkeys if wkeys is not None else window_keys(strands, min_len)
    assert k2 == k
    order = np.argsort(keys, kind="stable")
    ks, ws, wq = keys[order], ws[order], wq[order]
    scodes = _CODE[strands]
    out = []
    for a in range(0, hits.shape[0], chunk):
        h = hits[a:a + chunk].astype(np.int64)
        hk = keys_at(text, h, k)
        lo = np.searchsorted(ks, hk, "left")
        cnt = np.searchsorted(ks, hk, "right") - lo
        tot = int(cnt.sum())
        if not tot:
            continue
        hi = np.repeat(np.arange(h.shape[0], dtype=np.int64), cnt)
        wi = np.repeat(lo, cnt) + (np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        r, s, q = h[hi], ws[wi], wq[wi]
        # a MEM's own first window is left-maximal: keep those seeds only (every MEM has exactly one)
        inner = (r > 0) & (q > 0)
        same = np.zeros(tot, dtype=bool)
        same[inner] = _CODE[text[r[inner] - 1]] == scodes[s[inner], q[inner] - 1]
        keep = ~same
        r, s, q = r[keep], s[keep], q[keep]
        ln = np.full(r.shape[0], k, dtype=np.int64)
        act = np.arange(r.shape[0], dtype=np.int64)
        while act.shape[0]:
            rr, qq = r[act] + ln[act], q[act] + ln[act]
            ok = (rr < n) & (qq < L)
            act, rr, qq = act[ok], rr[ok], qq[ok]
            ok = _CODE[text[rr]] == scodes[s[act], qq]
            act = act[ok]
            ln[act] += 1
        good = ln >= min_len
        out.append(np.stack([s[good], r[good], q[good], ln[good]], axis=1))
    rows = np.concatenate(out) if out else np.zeros((0, 4), dtype=np.int64)
    return np.unique(rows, axis=0) if rows.shape[0] else rows


def verify_sample(text_h: np.ndarray, text_dev, reads_h: np.ndarray, sample_ids: np.ndarray, engine_rows: np.ndarray,
                  min_len: int, both: bool = True) -> dict:
    """Compare, as sets, the engine's MEMs for the sampled reads with the definitional set.
    engine_rows: uint32 [N,4] = (block, ref1, query1, len) as printed (1-based) for ALL reads of the batch;
    sample_ids: read indices (rows of reads_h are the reads of the WHOLE batch).  Returns counts and the differences."""
    sample_ids = np.unique(np.asarray(sample_ids, dtype=np.int64))
    strands = strands_of(reads_h[sample_ids], both)
    per = 2 if both else 1
    wkeys = window_keys(strands, min_len)
    keys, _, _, k = wkeys
    uniq = np.unique(keys)
    hits = scan_text_gpu(text_dev, k, uniq) if text_dev is not None else scan_text_numpy(text_h, k, uniq)
    want = definitional_mems(text_h, strands, min_len, hits, k, wkeys)
    # the engine's rows for the sampled reads, in the verifier's coordinates (sample strand index, 0-based positions)
    read_of_row = engine_rows[:, 0].astype(np.int64) // per
    pos = np.searchsorted(sample_ids, read_of_row)
    pos[pos >= sample_ids.shape[0]] = 0
    sel = sample_ids[pos] == read_of_row
    er = engine_rows[sel].astype(np.int64)
    got = np.stack([pos[sel] * per + er[:, 0] % per, er[:, 1] - 1, er[:, 2] - 1, er[:, 3]], axis=1)
    got_u = np.unique(got, axis=0) if got.shape[0] else got

    def as_set(a):
        return set(map(tuple, a.tolist()))
    gs, wset = as_set(got_u), as_set(want)
    return {"sampled_reads": int(sample_ids.shape[0]), "keys": int(uniq.shape[0]), "text_hits": int(hits.shape[0]),
            "definitional_mems": int(want.shape[0]), "engine_mems": int(got.shape[0]),
            "engine_duplicates": int(got.shape[0] - got_u.shape[0]),
            "missing": sorted(wset - gs)[:10], "missing_count": len(wset - gs),
            "extra": sorted(gs - wset)[:10], "extra_count": len(gs - wset),
            "definitional_beyond_2p31": int((want[:, 1] >= (1 << 31)).sum()) if want.shape[0] else 0}
