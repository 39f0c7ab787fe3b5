"""SURVEY.md 8(b)(2) over the GPU engine: libslamem_refapi.so (include/slamem_refapi.h) carries the reference's own function
names (bwtindex.h:1-10, lcparray.h:1-4) on top of the index in HBM.  The interval bookkeeping of the reference's scan
(slamem.c:105-129) written against THOSE names walks exactly the intervals the pinned oracle walks, on a text with a repeat
and N; locate, BWT letters, the LCP byte array, the sample count and the sizes agree too.  (A compatibility layer: one launch
per call -- the product's boundary is slamem_find_mems_* / slamem_stream_*.)"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def layer():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (torch.cuda.is_available() is False)")
    so = os.path.join(ROOT, "slamem_amd", "csrc", "libslamem_refapi.so")
    assert os.path.exists(so), "run __graft_entry__.build()"
    L = C.CDLL(so)
    u, up = C.c_uint, C.POINTER(C.c_uint)
    L.FMI_BuildIndex.argtypes = [C.POINTER(C.c_char_p), up, u, C.POINTER(C.POINTER(C.c_ubyte)), C.c_char]
    L.FMI_BuildIndex.restype = None
    L.BuildSampledLCPArray.argtypes = [C.c_char_p, u, C.POINTER(C.c_ubyte), C.c_int, C.c_int]
    L.FMI_GetBWTSize.restype = u
    L.FMI_GetTextSize.restype = u
    L.FMI_FollowLetter.argtypes = [C.c_char, up, up]
    L.FMI_FollowLetter.restype = u
    L.GetEnclosingLCPInterval.argtypes = [up, up]
    L.FMI_GetCharAtBWTPos.argtypes = [u]
    L.FMI_GetCharAtBWTPos.restype = C.c_char
    L.FMI_PositionInText.argtypes = [u]
    L.FMI_PositionInText.restype = u
    return L


def test_reference_named_functions_over_the_gpu_index():
    from oracle import pyoracle as po
    L = layer()
    rng = np.random.default_rng(3)
    text = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=5000, p=[0.24, 0.24, 0.24, 0.24, 0.04])
    text[1000:1400] = text[3000:3400]  # a repeat
    tb = text.tobytes()
    o = po.OracleIndex(tb)
    texts = (C.c_char_p * 1)(tb)
    sizes = (C.c_uint * 1)(len(tb))
    lcp = C.POINTER(C.c_ubyte)()
    L.FMI_BuildIndex(texts, sizes, 1, C.byref(lcp), b"\x00")
    ol = o.lcp
    assert L.BuildSampledLCPArray(tb, len(tb), lcp, 20, 0) == int((ol[:-1] != ol[1:]).sum())
    assert [lcp[i] for i in range(len(tb) + 1)] == [min(255, max(0, int(v))) for v in ol[:len(tb) + 1]]
    assert L.FMI_GetBWTSize() == len(tb) + 1 and L.FMI_GetTextSize() == len(tb)
    # the scan of slamem.c:105-129 over 500 letters that cross the repeat, against the reference-named functions ...
    q = tb[2950:3450]
    top, bot, depth = C.c_uint(0), C.c_uint(L.FMI_GetBWTSize()), 0  # slamem.c:110-111 (bottom = n+1, as the reference passes it)
    otop, obot, odepth = 0, len(tb), 0
    for i in range(len(q) - 1, -1, -1):
        c = q[i:i + 1]
        t0, b0 = top.value, bot.value
        while True:
            if L.FMI_FollowLetter(c, C.byref(top), C.byref(bot)):  # :121
                depth += 1
                break
            assert (top.value, bot.value) == (t0, min(b0, len(tb)))  # left as they were (:122-123 restores them anyway)
            d = L.GetEnclosingLCPInterval(C.byref(top), C.byref(bot))  # :124
            if d == -1:
                depth = 0
                break
            depth = d
            t0, b0 = top.value, bot.value
        # ... and the same steps on the oracle
        ch = chr(q[i])
        while True:
            r, t2, b2 = o.follow_letter(ch, otop, obot)
            if r:
                otop, obot, odepth = t2, b2, odepth + 1
                break
            d, otop, obot = o.enclosing_interval(otop, obot)
            if d == -1:
                odepth = 0
                break
            odepth = d
        assert (top.value, bot.value, depth) == (otop, obot, odepth), i
    rows = rng.integers(0, len(tb) + 1, size=200)
    assert [L.FMI_PositionInText(int(r)) for r in rows] == [o.position_in_text(int(r)) for r in rows]
    assert b"".join(L.FMI_GetCharAtBWTPos(int(r)) for r in rows).decode() == "".join(o.char_at_bwt_pos(int(r)) for r in rows)
    L.FMI_FreeIndex()
    L.FreeSampledSuffixArray()


def test_the_reference_driver_runs_on_the_gpu_index(tmp_path):
    """oracle/_ref/slaMEM-gpu-index (oracle/Makefile `hybrid`, built where /root/reference exists and carried along with the
    built tree): the REFERENCE'S OWN driver -- slamem.c with its GetMatches loop, sequence.c, tools.c -- linked against
    libslamem_refapi.so in place of bwtindex.c / lcparray.c / packednumbers.c.  Every index call of the reference's loop
    (slamem.c:73-77, 111-192) is answered by the GPU engine, and the files it prints for all golden cases are the files the
    real reference made, byte for byte -- options, both strands, N, multi-record references, -mam.  Where the binary was
    not built (a tree without the reference's sources) the ctypes-driven scan above is the check of the layer."""
    import subprocess
    from golden_cases import CASES, MANIFEST, case_paths
    exe = os.path.join(ROOT, "oracle", "_ref", "slaMEM-gpu-index")
    layer()  # (fails without a GPU)
    if not os.path.exists(exe):
        # (the binary is built from /root/reference where that exists and travels with the tree: a box without it must say so)
        pytest.skip("oracle/_ref/slaMEM-gpu-index not present: the reference's driver on the GPU index was NOT run (22 file comparisons)")
    for case in CASES:
        ref_fa, q_fa, exp_mems, _ = case_paths(case)
        out = str(tmp_path / (case + ".txt"))
        r = subprocess.run([exe] + MANIFEST[case]["opts"] + ["-o", out, ref_fa, q_fa] + MANIFEST[case].get("tail", []),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, (case, r.stdout.decode(errors="replace")[-500:])
        assert open(out, "rb").read() == open(exp_mems, "rb").read(), case
        assert b"> Done!" in r.stdout
