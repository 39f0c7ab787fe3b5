"""ctypes binding of libslamem_hip.so -- the C ABI declared in include/slamem_hip.h.

Loading is lazy so that CPU-only tooling (FASTA helpers, the synthetic generator) can import the
package on a machine without the library; every compute entry point goes through :func:`lib`, which
raises if the HIP library is missing.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLAMEM_HIP_LIB: another build of the same library (tools/variants.sh builds A/B variants of a kernel beside the product's
# library instead of over it; the product path never sets it)
LIB_PATH = os.environ.get("SLAMEM_HIP_LIB") or os.path.join(_HERE, "csrc", "libslamem_hip.so")
SYNTH_PATH = os.path.join(_HERE, "csrc", "libslamem_synth.so")

SLAMEM_OK = 0
SLAMEM_ERR_ARG = 1
SLAMEM_ERR_HIP = 2
SLAMEM_ERR_NOMEM = 3
SLAMEM_ERR_CAPACITY = 4
SLAMEM_ERR_FORMAT = 5
SLAMEM_ERR_IO = 6
SLAMEM_ERR_NO_DEVICE = 7

ARRAY_SA, ARRAY_BWT, ARRAY_LCP, ARRAY_PSV, ARRAY_NSV = range(5)
LAYOUT_AUTO, LAYOUT_FULL, LAYOUT_COMPACT = 0, 1, 2

# every symbol include/slamem_hip.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = (
    "slamem_abi_version", "slamem_strerror", "slamem_last_error_message", "slamem_device_count",
    "slamem_get_timings", "slamem_reset_timings", "slamem_device_warmup", "slamem_device_pci_bus_id",
    "slamem_index_build", "slamem_index_build_device", "slamem_index_build_layout", "slamem_index_build_device_layout",
    "slamem_index_build_bytes", "slamem_device_mem_info", "slamem_index_free", "slamem_index_get_info",
    "slamem_index_arena", "slamem_index_export", "slamem_index_attach", "slamem_index_adopt_arena", "slamem_index_save", "slamem_index_load",
    "slamem_index_validate_header", "slamem_search_stats_enable", "slamem_get_search_stats", "slamem_get_search_clock",
    "slamem_index_download", "slamem_index_sampled_lcp_stats",
    "slamem_follow_letter_batch", "slamem_enclosing_interval_batch", "slamem_position_in_text_batch",
    "slamem_char_at_bwt_pos_batch",
    "slamem_find_mems_workspace_bytes", "slamem_find_mems_device", "slamem_find_mems_host", "slamem_host_free",
    "slamem_find_mams_device", "slamem_find_mams_host",
    "slamem_stream_create", "slamem_stream_submit", "slamem_stream_submit_packed", "slamem_pack_reads", "slamem_stream_next",
    "slamem_stream_destroy",
    "slamem_pinned_alloc", "slamem_pinned_free", "slamem_copy_to_host",
)


class SlamemError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"slamem error {code}: {message}")
        self.code = code


class Mem(C.Structure):
    _fields_ = [("ref_pos", C.c_uint32), ("query_pos", C.c_uint32), ("length", C.c_uint32)]


class IndexInfo(C.Structure):
    _fields_ = [("text_length", C.c_uint32), ("bwt_size", C.c_uint32), ("num_n_rows", C.c_uint32),
                ("dollar_row", C.c_uint32), ("max_lcp", C.c_uint32), ("sort_rounds", C.c_uint32),
                ("arena_bytes", C.c_uint64), ("device", C.c_int32), ("owns_arena", C.c_int32),
                ("filter_k", C.c_uint32), ("layout", C.c_uint32), ("seed_k", C.c_uint32), ("reserved1", C.c_uint32)]


class SslcpStats(C.Structure):
    _fields_ = [("num_samples", C.c_uint64), ("num_oversized_lcp", C.c_uint64), ("sum_lcp", C.c_int64),
                ("max_lcp", C.c_uint32), ("pad", C.c_uint32), ("num_oversized_links", C.c_uint64),
                ("sum_link_distance", C.c_uint64), ("max_link_distance", C.c_uint64)]


class Timings(C.Structure):
    _fields_ = [("build_total_ms", C.c_float), ("build_pack_ms", C.c_float), ("build_sort_ms", C.c_float),
                ("build_bwt_ms", C.c_float), ("build_lcp_ms", C.c_float), ("build_links_ms", C.c_float),
                ("search_kernel_ms", C.c_float), ("search_total_ms", C.c_float),
                ("search_launches", C.c_uint64), ("search_kernel_ms_sum", C.c_double),
                ("prefilter_ms", C.c_float), ("k8_ms", C.c_float), ("prefilter_ms_sum", C.c_double),
                ("k8_ms_sum", C.c_double), ("seed_ms", C.c_float), ("reserved0", C.c_float), ("seed_ms_sum", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SearchStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "fm_lines_top", "fm_lines_bottom", "rec_lines_fail", "rec_lines_pend", "rec_lines_flush", "query_loads",
        "lane_trips", "wave_trips", "positions", "enum_jobs", "prefilter_probes", "prefilter_query_loads",
        "prefilter_items", "items", "survivors", "mems", "overflow_records", "valid", "dir_sa_lines", "dir_group_loads",
        "dir_rec_lines", "dir_letters", "jump_lines", "skip_group_loads", "skip_probe_lines", "skip_attempts", "skips",
        "enum_row_steps", "enum_levels", "enum_wave_us")] + [("state_lane_trips", C.c_uint64 * 11),
                                                               ("state_wave_trips", C.c_uint64 * 11)] + [
        (k, C.c_uint64) for k in ("seed_windows", "seed_compares", "seed_letter_masks", "seed_mems", "seed_strands_left",
                                  "seed_reads", "seed_query_bytes")] + [("seed_left_why", C.c_uint64 * 7), ("seed_once_reads", C.c_uint64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if (k.startswith("state_") or k == "seed_left_why") else int(getattr(self, k))) for k, _ in self._fields_}


_LIB = None
_SYNTH = None


def _declare(L):
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.slamem_abi_version.restype = i32
    L.slamem_strerror.restype = C.c_char_p
    L.slamem_strerror.argtypes = [i32]
    L.slamem_last_error_message.restype = C.c_char_p
    L.slamem_device_count.argtypes = [C.POINTER(i32)]
    L.slamem_device_pci_bus_id.argtypes = [i32, C.c_char_p, i32]
    L.slamem_get_timings.argtypes = [C.POINTER(Timings)]
    L.slamem_index_build.argtypes = [C.c_char_p, u32, i32, C.POINTER(vp)]
    L.slamem_index_build_device.argtypes = [vp, u32, i32, vp, C.POINTER(vp)]
    L.slamem_index_build_layout.argtypes = [C.c_char_p, u32, i32, i32, C.POINTER(vp)]
    L.slamem_index_build_device_layout.argtypes = [vp, u32, i32, vp, i32, C.POINTER(vp)]
    L.slamem_index_build_bytes.argtypes = [u32, i32, C.POINTER(u64), C.POINTER(u64)]
    L.slamem_device_mem_info.argtypes = [i32, C.POINTER(u64), C.POINTER(u64)]
    L.slamem_index_free.argtypes = [vp]
    L.slamem_index_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    L.slamem_index_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.slamem_index_export.argtypes = [vp, vp, u64, vp]
    L.slamem_index_attach.argtypes = [vp, u64, i32, C.POINTER(vp)]
    L.slamem_index_adopt_arena.argtypes = [vp]
    L.slamem_index_save.argtypes = [vp, C.c_char_p]
    L.slamem_index_load.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.slamem_index_validate_header.argtypes = [vp, u64, u64]
    L.slamem_search_stats_enable.argtypes = [i32]
    L.slamem_get_search_stats.argtypes = [C.POINTER(SearchStats)]
    L.slamem_get_search_clock.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.slamem_index_download.argtypes = [vp, i32, vp, u64]
    L.slamem_index_sampled_lcp_stats.argtypes = [vp, C.POINTER(SslcpStats)]
    L.slamem_follow_letter_batch.argtypes = [vp, vp, vp, vp, vp, u64, vp]
    L.slamem_enclosing_interval_batch.argtypes = [vp, vp, vp, vp, u64, vp]
    L.slamem_position_in_text_batch.argtypes = [vp, vp, vp, u64, vp]
    L.slamem_char_at_bwt_pos_batch.argtypes = [vp, vp, vp, u64, vp]
    L.slamem_find_mems_workspace_bytes.argtypes = [u32, i32, u64, u64, C.POINTER(u64)]
    L.slamem_find_mems_device.argtypes = [vp, vp, vp, u32, u64, u32, i32, vp, u64, vp, vp, u64, vp, C.POINTER(u64)]
    L.slamem_find_mems_host.argtypes = [vp, C.c_char_p, vp, u32, u32, i32, C.POINTER(C.POINTER(Mem)),
                                        C.POINTER(C.POINTER(u64)), C.POINTER(u64)]
    L.slamem_find_mams_device.argtypes = L.slamem_find_mems_device.argtypes
    L.slamem_find_mams_host.argtypes = L.slamem_find_mems_host.argtypes
    L.slamem_stream_create.argtypes = [vp, i32, u64, u32, i32, i32, C.POINTER(vp)]
    L.slamem_stream_submit.argtypes = [vp, vp, vp, u32, u32]
    L.slamem_copy_to_host.argtypes = [vp, vp, u64]
    L.slamem_stream_submit_packed.argtypes = [vp, vp, vp, vp, u32, u64, u32]
    L.slamem_pack_reads.argtypes = [vp, vp, u32, vp, vp, C.POINTER(u64), i32]
    L.slamem_stream_next.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(u32), C.POINTER(Timings)]
    L.slamem_stream_destroy.argtypes = [vp]
    L.slamem_pinned_alloc.argtypes = [C.POINTER(vp), u64]
    L.slamem_pinned_free.argtypes = [vp]
    L.slamem_host_free.argtypes = [vp]
    L.slamem_host_free.restype = None
    for name in ABI_SYMBOLS:
        f = getattr(L, name)
        if f.restype is C.c_int and name not in ("slamem_abi_version",):
            f.restype = i32


def _load_torch_runtime_first():
    # torch bundles its own libamdhip64; if libslamem_hip.so pulls in /opt/rocm's copy BEFORE torch loads its
    # own, the process ends up with two HIP runtimes and the second one sees no device.  Loading torch first
    # makes both share one runtime (the C front end never loads torch and simply uses /opt/rocm's).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib():
    """The loaded HIP library; raises (loudly) if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C slamem_amd/csrc`).  slamem_amd has no CPU fallback.")
        _load_torch_runtime_first()
        L = C.CDLL(LIB_PATH)
        _declare(L)
        if L.slamem_abi_version() != 4:
            raise ImportError("libslamem_hip.so ABI version mismatch")
        _LIB = L
    return _LIB


def synth_lib():
    global _SYNTH
    if _SYNTH is None:
        if not os.path.exists(SYNTH_PATH):
            raise ImportError(f"{SYNTH_PATH} is missing: run __graft_entry__.build()")
        _load_torch_runtime_first()
        S = C.CDLL(SYNTH_PATH)
        S.slamem_synth_reference.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        S.slamem_synth_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32,
                                         C.c_double, C.c_uint64, C.c_uint32, C.c_void_p]
        S.slamem_gather_modes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        S.slamem_gather_bench.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        S.slamem_synth_plant_repeats.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
        S.slamem_synth_reads_avoid.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32,
                                               C.c_double, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
        S.slamem_synth_plant_genome_like.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        _SYNTH = S
    return _SYNTH


def check(code: int):
    if code != SLAMEM_OK:
        raise SlamemError(code, lib().slamem_last_error_message().decode(errors="replace"))
