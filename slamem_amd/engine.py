"""Python view of the MEM engine: thin wrappers over the C ABI (include/slamem_hip.h).

torch is used for what the task calls plumbing only: device memory (tensors own the HBM buffers handed
to the C ABI as raw pointers), streams and torch.distributed.  All compute happens in libslamem_hip.so.

Method names mirror the reference's own interface for this path (bwtindex.h / lcparray.h as used by
GetMatches, slamem.c:37-218), in batched form.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi

MEM_DTYPE = np.dtype([("ref_pos", "<u4"), ("query_pos", "<u4"), ("length", "<u4")])


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def _stream_handle(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("slamem_amd runs on an MI355X only (device must be cuda:N); there is no CPU path")
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: slamem_amd has no CPU fallback")
    return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


class Index:
    """FM-index + parent-interval structure resident in HBM.

    ``Index.build`` replaces ``FMI_BuildIndex`` + ``BuildSampledLCPArray`` (slamem.c:73-74);
    ``close`` replaces ``FMI_FreeIndex`` + ``FreeSampledSuffixArray`` (slamem.c:208-209).
    """

    def __init__(self, handle: int, device: torch.device, keepalive=None):
        self._h = C.c_void_p(handle)
        self.device = device
        self._keepalive = keepalive  # arena tensor when attached
        info = capi.IndexInfo()
        capi.check(capi.lib().slamem_index_get_info(self._h, C.byref(info)))
        self.info = info
        self.n = int(info.text_length)

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def build(cls, text, device="cuda:0", layout: int = capi.LAYOUT_AUTO) -> "Index":
        """text: bytes / numpy uint8 / torch uint8 tensor (host or device) of A,C,G,T,N.  layout: capi.LAYOUT_AUTO (full
        when its build peak fits the free HBM, else compact), LAYOUT_FULL or LAYOUT_COMPACT (include/slamem_hip.h)."""
        dev = _require_gpu(device)
        L = capi.lib()
        if isinstance(text, (bytes, bytearray)):
            text = np.frombuffer(bytes(text), dtype=np.uint8)
        if isinstance(text, np.ndarray):
            text = torch.from_numpy(np.array(text, dtype=np.uint8, copy=True))
        if text.device != dev:
            text = text.to(dev)
        text = text.contiguous()
        n = text.numel()
        h = C.c_void_p()
        with torch.cuda.device(dev):
            stream = _stream_handle(dev)
            capi.check(L.slamem_index_build_device_layout(_ptr(text), n, dev.index, stream, int(layout), C.byref(h)))
        return cls(h.value, dev)

    @classmethod
    def attach(cls, arena: torch.Tensor) -> "Index":
        """Borrow an arena that was broadcast from a peer GPU (uint8 tensor on this device)."""
        dev = _require_gpu(arena.device)
        h = C.c_void_p()
        capi.check(capi.lib().slamem_index_attach(_ptr(arena), arena.numel(), dev.index, C.byref(h)))
        return cls(h.value, dev, keepalive=arena)

    @classmethod
    def load(cls, path: str, device="cuda:0") -> "Index":
        dev = _require_gpu(device)
        h = C.c_void_p()
        capi.check(capi.lib().slamem_index_load(path.encode(), dev.index, C.byref(h)))
        return cls(h.value, dev)

    def save(self, path: str) -> None:
        capi.check(capi.lib().slamem_index_save(self._h, path.encode()))

    def export_arena(self) -> torch.Tensor:
        """A torch-owned copy of the arena (what rank 0 hands to torch.distributed.broadcast)."""
        out = torch.empty(int(self.info.arena_bytes), dtype=torch.uint8, device=self.device)
        capi.check(capi.lib().slamem_index_export(self._h, _ptr(out), out.numel(), _stream_handle(self.device)))
        return out

    def arena_view(self) -> torch.Tensor:
        """Zero-copy uint8 tensor over the index arena (slamem_index_arena): what rank 0 hands to the broadcast, so the
        index is never held twice.  Valid while this Index is alive."""
        ptr, nbytes = C.c_void_p(), C.c_uint64()
        capi.check(capi.lib().slamem_index_arena(self._h, C.byref(ptr), C.byref(nbytes)))

        class _Arena:  # the CUDA array interface torch.as_tensor understands (HIP pointers on ROCm builds)
            __cuda_array_interface__ = {"shape": (int(nbytes.value),), "typestr": "|u1", "data": (int(ptr.value), False),
                                        "version": 2, "strides": None}
        holder = _Arena()
        t = torch.as_tensor(holder, device=self.device)
        assert t.data_ptr() == ptr.value and t.numel() == nbytes.value, "arena view must alias the arena"
        t._slamem_keepalive = (holder, self)
        return t

    def close(self) -> None:
        h, self._h = self._h, None
        if h:
            capi.lib().slamem_index_free(h)
        self._keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- FMI_GetBWTSize (bwtindex.c:263) ---------------------------------------------------------
    def bwt_size(self) -> int:
        return self.n + 1

    # ---- structure-level parity -----------------------------------------------------------------------
    def download(self, which: int) -> np.ndarray:
        n = self.n
        shape, dt = {capi.ARRAY_SA: (n + 1, np.uint32), capi.ARRAY_BWT: (n + 1, np.uint8),
                     capi.ARRAY_LCP: (n + 2, np.int32), capi.ARRAY_PSV: (n + 2, np.uint32),
                     capi.ARRAY_NSV: (n + 2, np.uint32)}[which]
        out = np.empty(shape, dtype=dt)
        capi.check(capi.lib().slamem_index_download(self._h, which, out.ctypes.data, shape))
        return out

    def sampled_lcp_stats(self) -> dict:
        """What BuildSampledLCPArray would report for this text (lcparray.c:709-711, 999-1000)."""
        st = capi.SslcpStats()
        capi.check(capi.lib().slamem_index_sampled_lcp_stats(self._h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in capi.SslcpStats._fields_ if k != "pad"}

    # ---- fine-grained operations, batched ------------------------------------------------------------
    def follow_letter(self, letters: bytes, top, bottom):
        """FMI_FollowLetter (bwtindex.c:359) for arrays of (letter, top, bottom) -> (size, top, bottom)."""
        dev = self.device
        cnt = len(letters)
        lt = torch.from_numpy(np.frombuffer(bytes(letters), dtype=np.uint8).copy()).to(dev)
        t = torch.as_tensor(np.asarray(top, dtype=np.uint32).view(np.int32)).to(dev)
        b = torch.as_tensor(np.asarray(bottom, dtype=np.uint32).view(np.int32)).to(dev)
        s = torch.zeros(cnt, dtype=torch.int32, device=dev)
        capi.check(capi.lib().slamem_follow_letter_batch(self._h, _ptr(lt), _ptr(t), _ptr(b), _ptr(s), cnt,
                                                         _stream_handle(dev)))
        torch.cuda.synchronize(dev)
        f = lambda x: x.cpu().numpy().view(np.uint32)
        return f(s), f(t), f(b)

    def enclosing_interval(self, top, bottom):
        """GetEnclosingLCPInterval (lcparray.c:330) -> (depth, top, bottom)."""
        dev = self.device
        t = torch.as_tensor(np.asarray(top, dtype=np.uint32).view(np.int32)).to(dev)
        b = torch.as_tensor(np.asarray(bottom, dtype=np.uint32).view(np.int32)).to(dev)
        d = torch.zeros(t.numel(), dtype=torch.int32, device=dev)
        capi.check(capi.lib().slamem_enclosing_interval_batch(self._h, _ptr(t), _ptr(b), _ptr(d), t.numel(),
                                                              _stream_handle(dev)))
        torch.cuda.synchronize(dev)
        return d.cpu().numpy(), t.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32)

    def position_in_text(self, rows):
        """FMI_PositionInText (bwtindex.c:402)."""
        dev = self.device
        r = torch.as_tensor(np.asarray(rows, dtype=np.uint32).view(np.int32)).to(dev)
        o = torch.zeros(r.numel(), dtype=torch.int32, device=dev)
        capi.check(capi.lib().slamem_position_in_text_batch(self._h, _ptr(r), _ptr(o), r.numel(), _stream_handle(dev)))
        torch.cuda.synchronize(dev)
        return o.cpu().numpy().view(np.uint32)

    def char_at_bwt_pos(self, rows) -> bytes:
        """FMI_GetCharAtBWTPos (bwtindex.c:304)."""
        dev = self.device
        r = torch.as_tensor(np.asarray(rows, dtype=np.uint32).view(np.int32)).to(dev)
        o = torch.zeros(r.numel(), dtype=torch.uint8, device=dev)
        capi.check(capi.lib().slamem_char_at_bwt_pos_batch(self._h, _ptr(r), _ptr(o), r.numel(), _stream_handle(dev)))
        torch.cuda.synchronize(dev)
        return o.cpu().numpy().tobytes()

    # ---- GetMatches (slamem.c:90-207) for a batch ---------------------------------------------------------
    def matcher(self, num_queries: int, both_strands: bool, mems_capacity: int, query_bytes: int,
                mam: bool = False) -> "Matcher":
        return Matcher(self, num_queries, both_strands, mems_capacity, query_bytes, mam)

    def find_mems(self, queries, offsets, min_len: int = 20, both_strands: bool = False, mam: bool = False):
        """Convenience: host arrays in, (mems structured array, block_offsets) out.  mam=True: -mam mode
        (slamem_find_mams_device; slamem.c:131,657)."""
        dev = self.device
        q = np.ascontiguousarray(np.frombuffer(queries, dtype=np.uint8) if isinstance(queries, (bytes, bytearray))
                                 else queries, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        num = offsets.shape[0] - 1
        qd = torch.zeros((q.shape[0] + 15) // 8 * 8, dtype=torch.uint8, device=dev)
        if q.shape[0]:
            qd[: q.shape[0]] = torch.from_numpy(q if q.flags.writeable else q.copy()).to(dev)
        od = torch.from_numpy(offsets.view(np.int64)).to(dev)
        cap = max(1024, q.shape[0] // 8 + 4 * num)
        while True:
            m = self.matcher(num, both_strands, cap, int(offsets[-1]) if num else 0, mam)
            try:
                total = m.run(qd, od, min_len)
                break
            except capi.SlamemError as e:
                if e.code != capi.SLAMEM_ERR_CAPACITY:
                    raise
                cap = int(m.last_total)
        mems = m.mems[:total].cpu().numpy().view(np.uint32).reshape(-1, 3)
        out = np.empty(total, dtype=MEM_DTYPE)
        out["ref_pos"], out["query_pos"], out["length"] = mems[:, 0], mems[:, 1], mems[:, 2]
        return out, m.block_offsets.cpu().numpy().view(np.uint64)


class Matcher:
    """Pre-allocated output + workspace buffers for repeated slamem_find_mems_device calls (bench loop)."""

    def __init__(self, index: Index, num_queries: int, both_strands: bool, mems_capacity: int, query_bytes: int,
                 mam: bool = False):
        self.index = index
        self.mam = bool(mam)
        self.num_queries = int(num_queries)
        self.both = bool(both_strands)
        self.capacity = int(mems_capacity)
        self.query_bytes = int(query_bytes)
        dev = index.device
        nb = self.num_queries * (2 if self.both else 1)
        need = C.c_uint64()
        capi.check(capi.lib().slamem_find_mems_workspace_bytes(self.num_queries, int(self.both), self.query_bytes,
                                                               self.capacity, C.byref(need)))
        self.workspace = torch.empty(int(need.value), dtype=torch.uint8, device=dev)
        self.mems = torch.empty((max(self.capacity, 1), 3), dtype=torch.int32, device=dev)
        self.block_offsets = torch.empty(nb + 1, dtype=torch.int64, device=dev)
        self.last_total = 0

    def run(self, queries_dev: torch.Tensor, offsets_dev: torch.Tensor, min_len: int) -> int:
        dev = self.index.device
        total = C.c_uint64()
        fn = capi.lib().slamem_find_mams_device if self.mam else capi.lib().slamem_find_mems_device
        rc = fn(
            self.index._h, _ptr(queries_dev), _ptr(offsets_dev), self.num_queries, self.query_bytes, int(min_len),
            int(self.both),
            _ptr(self.mems), self.capacity, _ptr(self.block_offsets), _ptr(self.workspace), self.workspace.numel(),
            _stream_handle(dev), C.byref(total))
        self.last_total = int(total.value)
        capi.check(rc)
        return self.last_total


class PinnedBuffer:
    """Page-locked host memory from slamem_pinned_alloc, viewed as a numpy uint8 array (a front end's read buffer)."""

    def __init__(self, nbytes: int):
        self._p = C.c_void_p()
        capi.check(capi.lib().slamem_pinned_alloc(C.byref(self._p), int(nbytes)))
        self.nbytes = int(nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self._p.value))

    def close(self):
        p, self._p = self._p, None
        if p:
            self.array = None
            capi.lib().slamem_pinned_free(p)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """slamem_stream_*: host-to-host MEM retrieval, pipelined over `slots` lanes (include/slamem_hip.h).  Replaces the
    query loop of GetMatches (slamem.c:90-207) for reads that live in host memory."""

    def __init__(self, index: Index, slots: int, max_batch_chars: int, max_batch_queries: int, both_strands: bool,
                 mam: bool = False):
        self.index = index
        self.both = bool(both_strands)
        self._h = C.c_void_p()
        capi.check(capi.lib().slamem_stream_create(index._h, int(slots), int(max_batch_chars), int(max_batch_queries),
                                                   int(self.both), int(bool(mam)), C.byref(self._h)))
        self._keep = []

    def submit(self, chars: np.ndarray, offsets: np.ndarray, min_len: int) -> None:
        """chars: uint8 array holding the records; offsets: uint64[num+1] into it (offsets[0] need not be 0)."""
        assert chars.dtype == np.uint8 and offsets.dtype == np.uint64 and offsets.flags.c_contiguous
        self._keep.append((chars, offsets))  # must stay alive and unchanged until collected
        capi.check(capi.lib().slamem_stream_submit(self._h, chars.ctypes.data, offsets.ctypes.data, offsets.shape[0] - 1,
                                                   int(min_len)))

    def submit_packed(self, planes: np.ndarray, other, offsets: np.ndarray, min_len: int, units: int = 0) -> None:
        """planes: uint8 view of the batch's 16-byte units (slamem_pack_reads layout), other: uint64 per unit or None; offsets:
        uint64[num+1] in letters.  All must stay alive and unchanged until collected."""
        assert planes.dtype == np.uint8 and offsets.dtype == np.uint64 and offsets.flags.c_contiguous
        self._keep.append((planes, other, offsets))
        capi.check(capi.lib().slamem_stream_submit_packed(self._h, planes.ctypes.data, other.ctypes.data if other is not None else None,
                                                          offsets.ctypes.data, offsets.shape[0] - 1, int(units), int(min_len)))

    def next(self, copy: bool = True):
        """(mems structured array, block_offsets uint64 array, timings dict) of the oldest batch.  copy=False returns
        views of the stream's pinned buffers, valid until the next call."""
        mems, boff, total, nq, tm = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint32(), capi.Timings()
        rc = capi.lib().slamem_stream_next(self._h, C.byref(mems), C.byref(boff), C.byref(total), C.byref(nq), C.byref(tm))
        if self._keep:
            self._keep.pop(0)
        capi.check(rc)
        nb = nq.value * (2 if self.both else 1)
        m = np.ctypeslib.as_array((C.c_uint8 * (12 * max(1, total.value))).from_address(mems.value))[: 12 * total.value]
        m = m.view(MEM_DTYPE)
        b = np.ctypeslib.as_array((C.c_uint64 * (nb + 1)).from_address(boff.value))
        if copy:
            m, b = m.copy(), b.copy()
        return m, b, tm.as_dict()

    def close(self):
        h, self._h = self._h, None
        if h:
            capi.lib().slamem_stream_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_reads(chars: np.ndarray, offsets: np.ndarray, planes_out: np.ndarray, other_out, threads: int = 16) -> int:
    """slamem_pack_reads: letters -> bit-planes (16-byte units) on the host; returns the number of units."""
    units = C.c_uint64()
    capi.check(capi.lib().slamem_pack_reads(chars.ctypes.data, offsets.ctypes.data, offsets.shape[0] - 1, planes_out.ctypes.data,
                                            other_out.ctypes.data if other_out is not None else None, C.byref(units), int(threads)))
    return int(units.value)


def host_to_host_leg(index: Index, reads_dev: torch.Tensor, count: int, read_len: int, min_len: int, both: bool,
                     steps: int = 2, batch_reads: int = 1_000_000, slots: int = 6, schedule=None, packed: bool = False) -> dict:
    """SURVEY.md 8(d)'s metric as defined -- reads resident in host memory -> MEM triples in host memory -- through
    slamem_stream_*: the reads sit in pinned host memory, batches of `batch_reads` are pipelined over `slots` lanes, and
    the clock runs from the first submit to the last result.  Returns fields for the bench line."""
    import time
    L = read_len
    buf = PinnedBuffer(count * L + 64)
    # (one DMA into the pinned buffer: torch's .cpu() of 1.5 GB goes through the runtime's staged copy of pageable memory)
    torch.cuda.synchronize(reads_dev.device)
    capi.check(capi.lib().slamem_copy_to_host(buf.array.ctypes.data, _ptr(reads_dev), count * L))
    # the record offsets live in pinned memory like the reads: 8 bytes per read go up with every batch, and from pageable
    # memory that copy runs at a fifth of the link's rate (measured: 0.8 ms of a 3.6 ms upload per million reads)
    obuf = PinnedBuffer((count + 1) * 8)
    offsets = obuf.array.view(np.uint64)
    offsets[:] = np.arange(count + 1, dtype=np.uint64) * np.uint64(L)
    # batch boundaries: short batches first (the pipeline starts searching after a quarter of a batch is up), or the sizes
    # the caller asks for (`schedule`, in reads; the last size repeats)
    cuts, pos = [0], 0
    sizes = list(schedule) if schedule else [batch_reads // 4, batch_reads // 2]
    for size in sizes:
        if 0 < size and pos + size < count:
            pos += size
            cuts.append(pos)
    tail = (sizes[-1] if schedule else batch_reads) or batch_reads
    while pos < count:
        pos = min(count, pos + tail)
        cuts.append(pos)
    nb = len(cuts) - 1
    biggest = int(np.diff(np.array(cuts)).max())
    pbuf, upr = None, (L + 63) // 64  # packed: the reads as bit-planes in pinned memory (made once, before the clock starts)
    if packed:
        pbuf = PinnedBuffer(count * upr * 16 + 64)
        assert pack_reads(buf.array, offsets, pbuf.array, None) == count * upr
    st = Stream(index, slots, biggest * L, biggest, both)

    def submit(b):
        if packed:
            st.submit_packed(pbuf.array[cuts[b] * upr * 16:], None, offsets[cuts[b]: cuts[b + 1] + 1], min_len, units=(cuts[b + 1] - cuts[b]) * upr)
        else:
            st.submit(buf.array, offsets[cuts[b]: cuts[b + 1] + 1], min_len)
    passes, total_mems, kernel_ms = [], 0, 0.0
    try:
        for rep in range(steps + 1):  # first pass warms the stream's buffers up
            t0 = time.perf_counter()
            got, kms = 0, 0.0
            for b in range(min(slots - 1, nb)):
                submit(b)
            marks = []
            for b in range(nb):
                m, _, tm = st.next(copy=False)
                got += len(m)
                kms += tm["search_kernel_ms"]
                marks.append((time.perf_counter(), len(m)))
                nxt = b + slots - 1
                if nxt < nb:
                    submit(nxt)
            dt = time.perf_counter() - t0
            if rep:
                # the pipeline's rate once it is full: results of the full-size batches in the middle of the run
                lo, hi = min(3, nb - 1), max(nb - 2, 0)
                steady = (sum(c for _, c in marks[lo + 1: hi + 1]) / (marks[hi][0] - marks[lo][0])) if hi > lo + 1 else None
                passes.append((dt, steady))
            total_mems, kernel_ms = got, kms
    finally:
        st.close()
        offsets = None
        obuf.close()
        buf.close()
        if pbuf is not None:
            pbuf.close()
    passes.sort(key=lambda x: x[0])
    best = passes[0][0]
    med, steady = passes[len(passes) // 2] if len(passes) % 2 else \
        ((passes[len(passes) // 2 - 1][0] + passes[len(passes) // 2][0]) / 2, passes[len(passes) // 2][1])
    return {"value_host_to_host": total_mems / med, "host_to_host_ms": med * 1e3, "host_to_host_mems": int(total_mems),
            "host_to_host": {"batches": nb, "batch_reads": batch_reads, "slots": slots,
                             "h2d_bytes": (count * upr * 16 if packed else count * L) + 8 * (count + nb), "packed": bool(packed),
                             "passes_ms": [round(p[0] * 1e3, 3) for p in passes], "best_ms": best * 1e3,
                             "d2h_bytes": 12 * int(total_mems) + 8 * (count * (2 if both else 1) + nb),
                             "kernel_ms_sum": kernel_ms,
                             "steady_state_MEMs_per_s": steady,
                             "steady_state_note": "results of the full-size batches in the middle of the run / the time between "
                                                  "them: the pipeline without its ramp and drain (a longer job tends to this)",
                             "note": "reads and record offsets in pinned host memory -> MEMs in pinned host memory through slamem_stream_* "
                                     "(uploads, kernels and downloads of neighbouring batches overlap); MEDIAN of "
                                     f"{steps} passes over the same {count} reads behind one warm-up pass"}}


def build_bytes(n: int, layout: int = capi.LAYOUT_FULL) -> tuple[int, int]:
    """(arena bytes, build peak bytes) of a text of n letters in `layout` (slamem_index_build_bytes; host arithmetic)."""
    a, p = C.c_uint64(), C.c_uint64()
    capi.check(capi.lib().slamem_index_build_bytes(int(n), int(layout), C.byref(a), C.byref(p)))
    return int(a.value), int(p.value)


def timings() -> dict:
    t = capi.Timings()
    capi.check(capi.lib().slamem_get_timings(C.byref(t)))
    return t.as_dict()


def reset_timings() -> None:
    capi.lib().slamem_reset_timings()


def search_stats(matcher: "Matcher", queries_dev: torch.Tensor, offsets_dev: torch.Tensor, min_len: int) -> dict:
    """One extra launch of the search through the diagnostic kernel instantiations: the load counters of K8a / K8 for
    this batch (slamem_search_stats; same results as the normal launch, slower)."""
    L = capi.lib()
    L.slamem_search_stats_enable(1)
    try:
        matcher.run(queries_dev, offsets_dev, min_len)
    finally:
        L.slamem_search_stats_enable(0)
    st = capi.SearchStats()
    capi.check(L.slamem_get_search_stats(C.byref(st)))
    out = st.as_dict()
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    L.slamem_get_search_clock(C.byref(a), C.byref(b), C.byref(c))
    out["k8_us_until_list_empty"], out["k8_us_tail"], out["k8_wave_us_sum"] = a.value, b.value, c.value
    return out


def random_line_ceiling(index: "Index", lanes: int = 256 * 32 * 64 * 4, iters: int = 64) -> float:
    """Dependent random 64-byte-line gathers per second over THIS index arena (read-only), measured in a few ms: every
    lane walks a chain of random lines of the arena, one 16-byte access per line (csrc/synth.hip::k_gather_modes, mode
    0).  The search kernel's request rate is read against this ceiling."""
    dev = index.device
    S = capi.synth_lib()
    arena = index.arena_view()
    nblk = arena.numel() // 64
    sink = torch.zeros(8, dtype=torch.int64, device=dev)
    st = _stream_handle(dev)
    best = 0.0
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = S.slamem_gather_modes(arena.data_ptr(), nblk, lanes, iters if rep else 4, 0, sink.data_ptr(), st)
        e1.record()
        if rc:
            raise RuntimeError(f"gather probe launch failed: hip error {rc}")
        torch.cuda.synchronize(dev)
        if rep:
            best = max(best, lanes * iters / (e0.elapsed_time(e1) * 1e-3))
    return best


# ---- bench / test support: synthetic inputs generated on the GPU (csrc/synth.hip) ----------------------
def synth_reference(n: int, seed: int = 42, device="cuda:0") -> torch.Tensor:
    dev = _require_gpu(device)
    out = torch.empty(n + 16, dtype=torch.uint8, device=dev)[:n]
    rc = capi.synth_lib().slamem_synth_reference(_ptr(out), n, seed, _stream_handle(dev))
    if rc:
        raise RuntimeError(f"synth kernel launch failed: hip error {rc}")
    return out


def synth_plant_repeats(ref: torch.Tensor, seed: int = 42) -> int:
    """Repeat model of SURVEY.md 8(d) applied in place to a text made by synth_reference(n, seed); returns the planted
    letters.  Same values as slamem_amd.synth.plant_repeats."""
    planted = C.c_uint64()
    rc = capi.synth_lib().slamem_synth_plant_repeats(_ptr(ref), ref.numel(), seed, _stream_handle(ref.device), C.byref(planted))
    if rc:
        raise RuntimeError(f"synth kernel launch failed: hip error {rc}")
    return int(planted.value)


def synth_plant_genome_like(ref: torch.Tensor, seed: int = 42, n_block: bool = True) -> None:
    """The genome-like repeat load (interspersed family, satellite array, block of N unless n_block=False) applied in place;
    same values as slamem_amd.synth.plant_genome_like."""
    rc = capi.synth_lib().slamem_synth_plant_genome_like(_ptr(ref), ref.numel(), seed, int(bool(n_block)), _stream_handle(ref.device))
    if rc:
        raise RuntimeError(f"synth kernel launch failed: hip error {rc}")


def synth_reads(ref: torch.Tensor, first: int, count: int, length: int = 150, sub: float = 0.02, seed: int = 42,
                rc_percent: int = 0, avoid: tuple = (0, 0)) -> torch.Tensor:
    """avoid = (at, len): reads that would touch text[at, at+len) are drawn behind it (slamem_amd.synth.make_reads)."""
    if count * length >= 1 << 32:
        raise ValueError("synth_reads: one thread per letter, fewer than 2^32 letters per call (make the reads in pieces)")
    dev = ref.device
    out = torch.zeros(count * length + 16, dtype=torch.uint8, device=dev)
    rc = capi.synth_lib().slamem_synth_reads_avoid(_ptr(ref), ref.numel(), _ptr(out), first, count, length, float(sub), seed,
                                                   rc_percent, int(avoid[0]), int(avoid[1]), _stream_handle(dev))
    if rc:
        raise RuntimeError(f"synth kernel launch failed: hip error {rc}")
    return out
