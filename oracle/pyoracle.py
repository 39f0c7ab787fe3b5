"""ctypes view of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (see oracle/oracle.h).  The product never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Mem(C.Structure):
    _fields_ = [("ref_pos", C.c_uint32), ("query_pos", C.c_uint32), ("length", C.c_uint32)]


class Counts(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("n_follow", "n_bwtchar", "n_lfstep", "n_locate", "n_parent", "n_querybase", "n_mem")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    def algorithmic_bytes(self) -> int:
        """the fixed reference-layout byte charge."""
        return (72 * self.n_follow + 36 * self.n_bwtchar + 36 * self.n_lfstep + 36 * self.n_locate
                + 40 * self.n_parent + 1 * self.n_querybase + 12 * self.n_mem)


MEM_DTYPE = np.dtype([("ref_pos", "<u4"), ("query_pos", "<u4"), ("length", "<u4")])


def build_lib(force: bool = False) -> str:
    if os.environ.get("SLAMEM_ORACLE_LIB"):  # another build of the restatement (oracle/Makefile: liboracle_asan.so)
        return os.environ["SLAMEM_ORACLE_LIB"]
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build_lib())
        L.oracle_build.restype = C.c_void_p
        L.oracle_build.argtypes = [C.c_char_p, C.c_uint32]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_bwt_size.restype = C.c_uint32
        L.oracle_bwt_size.argtypes = [C.c_void_p]
        for name, ty in (("oracle_sa", C.c_int32), ("oracle_lcp", C.c_int32), ("oracle_bwt", C.c_uint8),
                         ("oracle_psv", C.c_int32), ("oracle_nsv", C.c_int32)):
            f = getattr(L, name)
            f.restype = C.POINTER(ty)
            f.argtypes = [C.c_void_p]
        L.oracle_follow_letter.restype = C.c_uint32
        L.oracle_follow_letter.argtypes = [C.c_void_p, C.c_char, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_enclosing_interval.restype = C.c_int
        L.oracle_enclosing_interval.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_position_in_text.restype = C.c_uint32
        L.oracle_position_in_text.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.oracle_char_at_bwt_pos.restype = C.c_char
        L.oracle_char_at_bwt_pos.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_get_matches.restype = C.c_size_t
        L.oracle_get_matches.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int,
                                         C.POINTER(C.POINTER(Mem)), C.POINTER(C.c_size_t), C.c_size_t,
                                         C.POINTER(Counts)]
        L.oracle_get_matches_mode.restype = C.c_size_t
        L.oracle_get_matches_mode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int, C.c_int,
                                              C.POINTER(C.POINTER(Mem)), C.POINTER(C.c_size_t), C.c_size_t,
                                              C.POINTER(Counts)]
        L.oracle_match_batch_mode.restype = C.c_size_t
        L.oracle_match_batch_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.POINTER(Mem)), C.POINTER(C.c_size_t), C.c_void_p,
                                              C.POINTER(Counts)]
        L.oracle_brute_force_mems.restype = C.c_size_t
        L.oracle_brute_force_mems.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int,
                                              C.POINTER(C.POINTER(Mem)), C.POINTER(C.c_size_t), C.c_size_t]
        L.oracle_match_batch.restype = C.c_size_t
        L.oracle_match_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                                         C.POINTER(C.POINTER(Mem)), C.POINTER(C.c_size_t), C.c_void_p,
                                         C.POINTER(Counts)]
        L.oracle_reverse_complement.argtypes = [C.c_char_p, C.c_int]
        L.oracle_seq_id_from_merged_pos.restype = C.c_int
        L.oracle_seq_id_from_merged_pos.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
        L.oracle_free_mems.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _take(L, out, n) -> np.ndarray:
    if n == (1 << 64) - 1:
        raise MemoryError("oracle allocation failure")
    arr = np.empty(n, dtype=MEM_DTYPE)
    if n:
        C.memmove(arr.ctypes.data, out, n * 12)
    L.oracle_free_mems(out)
    return arr


class OracleIndex:
    """FMI_BuildIndex + BuildSampledLCPArray restated (bwtindex.c:1318, lcparray.c:545)."""

    def __init__(self, text: bytes):
        self.L = lib()
        self.text = bytes(text)
        self.n = len(self.text)
        self.h = self.L.oracle_build(self.text, self.n)
        if not self.h:
            raise MemoryError("oracle_build failed")

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.L.oracle_free(h)

    def _arr(self, fn, count, dtype):
        p = fn(self.h)
        return np.ctypeslib.as_array(p, shape=(count,)).astype(dtype, copy=True)

    @property
    def sa(self): return self._arr(self.L.oracle_sa, self.n + 1, np.int64)
    @property
    def lcp(self): return self._arr(self.L.oracle_lcp, self.n + 2, np.int64)
    @property
    def bwt(self): return self._arr(self.L.oracle_bwt, self.n + 1, np.uint8)
    @property
    def psv(self): return self._arr(self.L.oracle_psv, self.n + 2, np.int64)
    @property
    def nsv(self): return self._arr(self.L.oracle_nsv, self.n + 2, np.int64)

    def follow_letter(self, c: str, top: int, bottom: int):
        t, b = C.c_uint32(top), C.c_uint32(bottom)
        n = self.L.oracle_follow_letter(self.h, c.encode(), C.byref(t), C.byref(b))
        return n, t.value, b.value

    def enclosing_interval(self, top: int, bottom: int):
        t, b = C.c_uint32(top), C.c_uint32(bottom)
        d = self.L.oracle_enclosing_interval(self.h, C.byref(t), C.byref(b))
        return d, t.value, b.value

    def position_in_text(self, row: int) -> int:
        return self.L.oracle_position_in_text(self.h, row, None)

    def char_at_bwt_pos(self, row: int) -> str:
        return self.L.oracle_char_at_bwt_pos(self.h, row).decode()

    def get_matches(self, query: bytes, min_len: int, counts: Counts | None = None, mam: bool = False) -> np.ndarray:
        out = C.POINTER(Mem)()
        cap = C.c_size_t(0)
        n = self.L.oracle_get_matches_mode(self.h, bytes(query), len(query), min_len, int(mam), C.byref(out),
                                           C.byref(cap), 0, C.byref(counts) if counts is not None else None)
        return _take(self.L, out, n)

    def match_batch(self, queries: np.ndarray, offsets: np.ndarray, min_len: int, both: bool,
                    counts: Counts | None = None, mam: bool = False):
        """queries: uint8 concatenation; offsets: uint64[num+1].  Returns (mems, block_counts)."""
        queries = np.ascontiguousarray(queries, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        num = offsets.shape[0] - 1
        bc = np.zeros(num * (2 if both else 1), dtype=np.uint64)
        out = C.POINTER(Mem)()
        cap = C.c_size_t(0)
        n = self.L.oracle_match_batch_mode(self.h, queries.ctypes.data, offsets.ctypes.data, num, min_len, int(both),
                                           int(mam), C.byref(out), C.byref(cap), bc.ctypes.data,
                                           C.byref(counts) if counts is not None else None)
        return _take(self.L, out, n), bc


def brute_force_mems(text: bytes, query: bytes, min_len: int) -> np.ndarray:
    L = lib()
    out = C.POINTER(Mem)()
    cap = C.c_size_t(0)
    n = L.oracle_brute_force_mems(bytes(text), len(text), bytes(query), len(query), min_len,
                                  C.byref(out), C.byref(cap), 0)
    return _take(L, out, n)


def reverse_complement(s: bytes) -> bytes:
    buf = C.create_string_buffer(bytes(s), len(s) + 1)
    lib().oracle_reverse_complement(buf, len(s))
    return buf.raw[:len(s)]


def sorted_triples(m: np.ndarray) -> np.ndarray:
    """(ref_pos, query_pos, length) rows sorted lexicographically -- 'modulo ordering' comparisons."""
    a = np.stack([m["ref_pos"], m["query_pos"], m["length"]], axis=1).astype(np.int64) if len(m) else \
        np.zeros((0, 3), dtype=np.int64)
    order = np.lexsort((a[:, 2], a[:, 1], a[:, 0])) if len(a) else np.zeros(0, dtype=np.int64)
    return a[order]
