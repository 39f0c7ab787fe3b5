"""The C-ABI library loads and exports every symbol include/slamem_hip.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "slamem_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slamem_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from slamem_amd import capi
    L = capi.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libslamem_hip.so does not export {n}"
    assert set(names) == set(capi.ABI_SYMBOLS)
    assert L.slamem_abi_version() == 4
    assert L.slamem_strerror(0) == b"ok"


def test_no_silent_cpu_fallback():
    """Without a GPU every compute entry point must fail with an error code and a message."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from slamem_amd import capi, engine
    L = capi.lib()
    h = C.c_void_p()
    rc = L.slamem_index_build(b"ACGTACGT", 8, 0, C.byref(h))
    assert rc == capi.SLAMEM_ERR_NO_DEVICE and not h.value
    assert b"no CPU" in L.slamem_last_error_message()
    with pytest.raises(RuntimeError):
        engine.Index.build(b"ACGT")


def test_product_does_not_reference_the_oracle():
    """oracle/ is test infrastructure: nothing under slamem_amd/, include/ or tools/ may import, include or link it (the
    scripts that check against it live under tests/tools/)."""
    bad = []
    for base in ("slamem_amd", "include", "tools"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".c", ".h", ".hip", ".sh", "Makefile")):
                    s = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|oracle\.h|liboracle|pyoracle", s):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_rccl_companion_library_exports_replicate():
    """libslamem_rccl.so (include/slamem_rccl.h): the C front end's multi-GPU step; separate from libslamem_hip.so so
    that torch's own RCCL is never doubled in the Python harness."""
    so = os.path.join(ROOT, "slamem_amd", "csrc", "libslamem_rccl.so")
    assert os.path.exists(so), "run __graft_entry__.build()"
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", so], stdout=subprocess.PIPE).stdout.decode()
    assert " T slamem_index_replicate" in syms
    text = open(os.path.join(ROOT, "include", "slamem_rccl.h")).read()
    assert "slamem_index_replicate" in text


def test_reference_named_layer_exports_the_reference_names():
    """libslamem_refapi.so (include/slamem_refapi.h): SURVEY 8(b)(2) over the GPU engine -- every function the header declares is
    exported under the reference's own name, and the layer is built on the C ABI (it needs libslamem_hip.so, not the oracle)."""
    so = os.path.join(ROOT, "slamem_amd", "csrc", "libslamem_refapi.so")
    assert os.path.exists(so), "run __graft_entry__.build()"
    import subprocess
    text = open(os.path.join(ROOT, "include", "slamem_refapi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b([A-Za-z_]+)\s*\(", text)) - {"defined"})
    assert names == ["BuildSampledLCPArray", "FMI_BuildIndex", "FMI_FollowLetter", "FMI_FreeIndex", "FMI_GetBWTSize", "FMI_GetCharAtBWTPos",
                     "FMI_GetTextSize", "FMI_PositionInText", "FreeSampledSuffixArray", "GetEnclosingLCPInterval"]
    syms = subprocess.run(["nm", "-D", "--defined-only", so], stdout=subprocess.PIPE).stdout.decode()
    for n in names:
        assert " T %s\n" % n in syms, n
    needed = subprocess.run(["readelf", "-d", so], stdout=subprocess.PIPE).stdout.decode()
    assert "libslamem_hip.so" in needed and "oracle" not in needed


def _arena_header(n=100_000, num_n=3, with_filter=True):
    """A consistent arena header, laid out as slamem_amd/csrc/common.h::ArenaHeader does (host-side mirror for the test)."""
    import struct
    R = n + 1
    al = lambda x: (x + 255) // 256 * 256
    nblocks = (R + 1 + 127) >> 7
    off = 4096
    off_fm = off; off = al(off + nblocks * 64)
    off_rec = off; off = al(off + (R + 1) * 16)
    off_sa = off; off = al(off + R * 4)
    off_nrows = off; off = al(off + max(num_n, 1) * 4)
    lg, k = 17, 13
    off_kf = off if with_filter else 0
    if with_filter:
        off = al(off + (8 << lg))
    off_tg = off; off = al(off + ((R >> 4) + 2) * 16)
    off_pr = off; off = al(off + R * 16)
    kj = 7
    off_kj = off; off = al(off + (8 << (2 * kj)))
    kb = 11
    off_kb = off; off = al(off + ((1 << (2 * kb)) >> 3))
    # seed-and-compare sections (arena version 12): seed table, text bit-planes, letter mask, unit mask; 13: spill list
    sk, slg = 11, 15
    units = (n + 63) // 64 + 4
    off_seed = off; off = al(off + (64 << slg))
    off_tpl = off; off = al(off + units * 32)
    off_tnm = off_tnb = 0
    spill_cap = ((n // 16 + 3) & ~3) + 64
    off_spill = off; off = al(off + spill_cap * 8)
    off_tuq = 0
    fields = dict(magic_lo=0x4D414C53, magic_hi=0x58494845, version=13, seed_k=sk, seed_log2=slg, off_seed=off_seed,
                  off_tpl=off_tpl, off_tnm=off_tnm, off_tnb=off_tnb, off_spill=off_spill, spill_cap=spill_cap, spill_used=8, off_tuq=off_tuq, layout=1, lcp_ge=0, n=n, total_bytes=off, off_fm=off_fm, off_rec=off_rec,
                  off_sa=off_sa, off_nrows=off_nrows, off_kfilter=off_kf, r0=off_tg, r1=off_pr, r2=off_kj, kjump_k=kj, C=0, kbits_k=kb, z=0, off_kbits=off_kb, kfilter_log2=lg if with_filter else 0,
                  kfilter_k=k if with_filter else 0, nblocks=nblocks, dollar_row=17, num_n=num_n, max_lcp=20, sort_rounds=1)
    order = ["magic_lo", "magic_hi", "version", "n", "total_bytes", "off_fm", "off_rec", "off_sa", "off_nrows", "off_kfilter",
             "r0", "r1", "r2", "kfilter_log2", "kfilter_k", "nblocks", "dollar_row", "num_n", "max_lcp", "sort_rounds",
             "C", "C", "C", "C", "C", "C", "kjump_k", "kbits_k", "z", "off_kbits", "layout"] + ["lcp_ge"] * 10 + [
             "seed_k", "seed_log2", "z", "off_seed", "off_tpl", "off_tnm", "off_tnb", "off_spill", "spill_cap", "spill_used", "off_tuq"]
    fmt = "<4I9Q7I6III" + "IQ" + "I10I" + "2II4Q" + "Q2IQ"
    def pack(**over):
        f = dict(fields); f.update(over)
        return struct.pack(fmt, *[f[k] for k in order]) + b"\0" * (4096 - struct.calcsize(fmt))
    return fields, pack


def test_header_validation_rejects_corrupt_arenas(tmp_path):
    """every section offset / size the kernels index is checked on the host; a truncated, stale or corrupted
    index is SLAMEM_ERR_FORMAT before any device call (so this runs without a GPU)."""
    from slamem_amd import capi
    L = capi.lib()
    ERR_FORMAT = 5
    f, pack = _arena_header()
    good = pack()
    assert L.slamem_index_validate_header(good, len(good), f["total_bytes"]) == 0
    f2, pack2 = _arena_header(with_filter=False)
    assert L.slamem_index_validate_header(pack2(), 4096, f2["total_bytes"]) == 0
    assert L.slamem_index_validate_header(good, len(good), f["total_bytes"] - 1) == ERR_FORMAT       # truncated file
    bad = [dict(magic_lo=1), dict(version=10), dict(version=12), dict(spill_cap=f["spill_cap"] + 4), dict(spill_used=f["spill_cap"] + 1),
           dict(off_spill=f["total_bytes"] - 64), dict(off_spill=0), dict(off_tuq=4096), dict(seed_k=17), dict(seed_k=3), dict(seed_log2=f["seed_log2"] + 8),
           dict(seed_log2=9), dict(off_seed=f["off_seed"] + 64), dict(off_tpl=f["off_spill"]), dict(off_tnm=256), dict(off_tnb=f["total_bytes"]),
           dict(off_seed=0), dict(kbits_k=17), dict(kbits_k=f['kbits_k'] + 1), dict(off_kbits=f['off_kbits'] + 32), dict(kjump_k=13), dict(kjump_k=8), dict(r2=f['r2'] + 64), dict(r0=f['r0'] + 8), dict(r1=f['r1'] + 4096), dict(r0=0), dict(n=0), dict(nblocks=f["nblocks"] - 1), dict(nblocks=f["nblocks"] + 1),
           dict(off_fm=8192), dict(off_rec=f["off_rec"] + 64), dict(off_rec=f["off_fm"]), dict(off_sa=f["total_bytes"]),
           dict(off_sa=f["off_rec"] + 256), dict(off_nrows=f["off_sa"]), dict(off_kfilter=f["total_bytes"] - 256),
           dict(kfilter_log2=48), dict(kfilter_log2=f["kfilter_log2"] + 1), dict(kfilter_k=40), dict(num_n=f["n"] + 1),
           dict(num_n=1 << 24), dict(dollar_row=f["n"] + 1), dict(total_bytes=f["total_bytes"] - 256),
           dict(total_bytes=100), dict(n=f["n"] + 4096)]
    for over in bad:
        assert L.slamem_index_validate_header(pack(**over), 4096, f["total_bytes"]) == ERR_FORMAT, over
    # through slamem_index_load: format errors come before the device check, so they are the same on any machine
    import ctypes as C
    p = tmp_path / "bad.idx"
    p.write_bytes(pack(off_sa=f["off_rec"] + 256) + b"\0" * 4096)
    h = C.c_void_p()
    assert L.slamem_index_load(str(p).encode(), 0, C.byref(h)) == ERR_FORMAT and not h.value
    p.write_bytes(good)  # header fine, file far shorter than total_bytes
    assert L.slamem_index_load(str(p).encode(), 0, C.byref(h)) == ERR_FORMAT and not h.value


def test_index_build_bytes_is_host_arithmetic():
    """slamem_index_build_bytes (what a front end compares with slamem_device_mem_info before it builds): no GPU needed; the
    sizes are the ones measured on the MI355X (profiles/r03_compact_layout.jsonl: 6.03 / 3.26 GB at 100 Mbp, 150.7 / 80.9 GB
    at 3.1 Gbp -- 188.2 since texts of 2^28 letters and more carry a seed table too; since round 4 the full layout of a text below 2^28 letters also carries the seed table and the text
    bit-planes: 8.27 GB at 100 Mbp), the compact layout is smaller, the peak is above the arena, bad arguments are refused."""
    from slamem_amd import capi
    L = capi.lib()
    a, p = C.c_uint64(), C.c_uint64()

    def sizes(n, layout):
        assert L.slamem_index_build_bytes(n, layout, C.byref(a), C.byref(p)) == capi.SLAMEM_OK
        return a.value, p.value
    f100, pf100 = sizes(100_000_000, capi.LAYOUT_FULL)
    c100, pc100 = sizes(100_000_000, capi.LAYOUT_COMPACT)
    f3g, pf3g = sizes(3_100_000_000, capi.LAYOUT_FULL)
    c3g, pc3g = sizes(3_100_000_000, capi.LAYOUT_COMPACT)
    assert abs(f100 - 8.2794e9) < 2e7 and abs(c100 - 3.2580e9) < 2e7
    assert abs(f3g - 188.204e9) < 1e8 and abs(c3g - 80.864e9) < 1e8  # (full: + 34.4 GB of seed table, 1.6 of text units, 1.6 of spill list)
    for arena, peak in ((f100, pf100), (c100, pc100), (f3g, pf3g), (c3g, pc3g)):
        assert arena < peak < arena + 40 * 3_100_000_001
    assert pc100 < pf100 and pc3g < pf3g < 288 * 2**30  # the full layout of a 3.1 Gbp text fits an MI355X while it is built
    assert L.slamem_index_build_bytes(0, capi.LAYOUT_FULL, C.byref(a), C.byref(p)) == capi.SLAMEM_ERR_ARG
    assert L.slamem_index_build_bytes(1000, capi.LAYOUT_AUTO, C.byref(a), C.byref(p)) == capi.SLAMEM_ERR_ARG


def test_pack_reads_is_host_code_and_matches_numpy():
    """slamem_pack_reads (what a caller puts in front of slamem_stream_submit_packed): letters -> two bit-planes + the plane of
    letters that are not A,C,G,T, 16-byte units, a record starts a new unit; no GPU involved."""
    import numpy as np
    from slamem_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(5)
    lens = [0, 1, 63, 64, 65, 128, 150, 150, 300, 7]
    recs = [rng.choice(np.frombuffer(b"ACGTacgtNRY", dtype=np.uint8), size=n) for n in lens]
    q = np.concatenate(recs)
    off = np.zeros(len(recs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    units = sum((n + 63) // 64 for n in lens)
    for threads in (1, 4):
        pl = np.zeros(2 * units + 2, dtype=np.uint64)
        ot = np.zeros(units + 1, dtype=np.uint64)
        got = C.c_uint64()
        assert L.slamem_pack_reads(q.ctypes.data, off.ctypes.data, len(recs), pl.ctypes.data, ot.ctypes.data, C.byref(got), threads) == 0
        assert got.value == units
        u = 0
        code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
        for r in recs:
            for k in range(0, len(r), 64):
                p0 = p1 = o = 0
                for i, ch in enumerate(r[k:k + 64]):
                    c = code.get(int(ch) & 0xDF)
                    if c is None:
                        o |= 1 << i
                    else:
                        p0 |= (c & 1) << i
                        p1 |= (c >> 1) << i
                assert (int(pl[2 * u]), int(pl[2 * u + 1]), int(ot[u])) == (p0, p1, o)
                u += 1
        assert u == units
