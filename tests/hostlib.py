"""ctypes view of slamem_amd/host/libslamem_host.so (the front end's host logic) for the CPU tests."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "slamem_amd", "host")


class Record(C.Structure):
    _fields_ = [("name", C.c_char_p), ("size", C.c_uint32)]


class SeqSet(C.Structure):
    _fields_ = [("recs", C.POINTER(Record)), ("num", C.c_int), ("chars", C.POINTER(C.c_char)), ("total", C.c_uint64),
                ("offsets", C.POINTER(C.c_uint64)), ("merged_start", C.POINTER(C.c_uint32)), ("file_bytes", C.c_long),
                ("name_arena", C.c_void_p)]


class Options(C.Structure):
    _fields_ = [("usage", C.c_int), ("hidden_sort", C.c_int), ("hidden_clean", C.c_int), ("image_arg", C.c_int),
                ("no_ns", C.c_int), ("min_seq_len", C.c_int), ("ref_name", C.c_char_p), ("ref_name_given", C.c_int),
                ("ref_name_empty", C.c_int), ("match_type", C.c_int), ("both_strands", C.c_int), ("min_mem_len", C.c_int),
                ("out_arg", C.c_int), ("num_files", C.c_int), ("file_args", C.POINTER(C.c_int))]


class Buffer(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_char)), ("len", C.c_size_t), ("cap", C.c_size_t)]


_L = None


def lib():
    global _L
    if _L is None:
        so = os.path.join(HOST_DIR, "libslamem_host.so")
        src = os.path.join(HOST_DIR, "slamem_host.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", HOST_DIR, "libslamem_host.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.slh_load_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_char_p, C.c_int, C.c_long,
                                    C.POINTER(SeqSet), C.c_void_p]
        L.slh_free_seqset.argtypes = [C.POINTER(SeqSet)]
        L.slh_parse_options.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(Options)]
        L.slh_free_options.argtypes = [C.POINTER(Options)]
        L.slh_parse_argument.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
        L.slh_append_to_basename.restype = C.c_void_p
        L.slh_append_to_basename.argtypes = [C.c_char_p, C.c_char_p]
        L.slh_format_block.argtypes = [C.POINTER(Buffer), C.c_char_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(Record),
                                       C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint64)]
        L.slh_buffer_free.argtypes = [C.POINTER(Buffer)]
        L.slh_progress_dots.argtypes = [C.c_uint32]
        L.slh_seq_id_from_merged_pos.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
        _L = L
    return _L


def parse_options(args):
    argv = (C.c_char_p * (len(args) + 1))(*[a.encode() for a in args], None)
    o = Options()
    assert lib().slh_parse_options(len(args), argv, C.byref(o)) == 0
    d = {k: getattr(o, k) for k, _ in Options._fields_ if k != "file_args"}
    d["files"] = [args[o.file_args[i]] for i in range(o.num_files)]
    d["ref_name"] = o.ref_name.decode() if o.ref_name else None
    lib().slh_free_options(C.byref(o))
    return d


class Loaded:
    def __init__(self, path, merge, acgt_only=0, min_len=0, name_filter=None, log_limit=100):
        self.s = SeqSet()
        self.n = lib().slh_load_file(path.encode(), merge, acgt_only, min_len,
                                     name_filter.encode() if name_filter is not None else None, 1, log_limit, C.byref(self.s), None)
        s = self.s
        self.names = [s.recs[i].name for i in range(s.num)]
        self.sizes = [s.recs[i].size for i in range(s.num)]
        self.chars = C.string_at(s.chars, s.total) if self.n else b""
        self.offsets = [s.offsets[i] for i in range(s.num + 1)] if (self.n and not merge) else None
        self.merged_start = [s.merged_start[i] for i in range(s.num)] if (self.n and merge) else None

    def __del__(self):
        lib().slh_free_seqset(C.byref(self.s))


def format_block(name: bytes, reverse: int, mems, ref: Loaded) -> bytes:
    """mems: (count,3) uint32 numpy array of 0-based triples."""
    import numpy as np
    m = np.ascontiguousarray(mems, dtype=np.uint32)
    b = Buffer()
    s = C.c_uint64()
    rc = lib().slh_format_block(C.byref(b), name, reverse, m.ctypes.data, m.shape[0], ref.s.recs, ref.s.merged_start,
                                ref.s.num, C.byref(s))
    assert rc == 0
    out = C.string_at(b.data, b.len)
    lib().slh_buffer_free(C.byref(b))
    return out
