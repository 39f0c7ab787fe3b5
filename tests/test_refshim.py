"""the reference-named index functions (oracle/refshim.h) behave as slamem.c's hot loop expects -- the
scan's interval bookkeeping (slamem.c:105-129) written against THOSE names and driven through ctypes
walks exactly the intervals the pinned oracle walks, on a text with a repeat and N; locate, BWT letters, the LCP byte array,
the sample count, reverse complement and the merged-position lookup agree too."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
ODIR = os.path.join(os.path.dirname(HERE), "oracle")


def shim():
    subprocess.check_call(["make", "-C", ODIR, "liboracle_refshim.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ODIR, "liboracle_refshim.so"))
    u, up = C.c_uint, C.POINTER(C.c_uint)
    L.FMI_BuildIndex.argtypes = [C.POINTER(C.c_char_p), up, u, C.POINTER(C.POINTER(C.c_ubyte)), C.c_char]
    L.FMI_BuildIndex.restype = None
    L.BuildSampledLCPArray.argtypes = [C.c_char_p, u, C.POINTER(C.c_ubyte), C.c_int, C.c_int]
    L.FMI_GetBWTSize.restype = u
    L.FMI_FollowLetter.argtypes = [C.c_char, up, up]
    L.FMI_FollowLetter.restype = u
    L.GetEnclosingLCPInterval.argtypes = [up, up]
    L.FMI_GetCharAtBWTPos.argtypes = [u]
    L.FMI_GetCharAtBWTPos.restype = C.c_char
    L.FMI_PositionInText.argtypes = [u]
    L.FMI_PositionInText.restype = u
    L.ReverseComplementSequence.argtypes = [C.c_char_p, C.c_int]
    L.RefShim_SetMergedStarts.argtypes = [up, C.c_int]
    L.GetSeqIdFromMergedSeqsPos.argtypes = [up]
    return L


def scan_states(L, query: bytes):
    """The interval bookkeeping of slamem.c:105-129 against the reference-named functions: (position, top, bottom, depth)
    after every query letter, scanning right to left from the root."""
    out = []
    n = len(query)
    top, bot = C.c_uint(0), C.c_uint(L.FMI_GetBWTSize())  # slamem.c:110-111 (bottom = n+1, as the reference passes it)
    depth = 0
    i = n
    while i > 0:
        i -= 1
        c = query[i:i + 1]
        t0, b0 = top.value, bot.value
        while True:
            if L.FMI_FollowLetter(c, C.byref(top), C.byref(bot)):  # :121
                depth += 1
                break
            top.value, bot.value = t0, b0  # :122-123
            d = L.GetEnclosingLCPInterval(C.byref(top), C.byref(bot))  # :124
            if d == -1:
                depth = 0
                break
            depth = d
            t0, b0 = top.value, bot.value
        out.append(("state", i, top.value, bot.value, depth))
    return out


def test_reference_named_functions_agree_with_the_oracle():
    L = shim()
    rng = np.random.default_rng(3)
    text = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=5000, p=[0.24, 0.24, 0.24, 0.24, 0.04])
    text[1000:1400] = text[3000:3400]  # a repeat
    tb = text.tobytes()
    o = po.OracleIndex(tb)
    texts = (C.c_char_p * 1)(tb)
    sizes = (C.c_uint * 1)(len(tb))
    lcp = C.POINTER(C.c_ubyte)()
    L.FMI_BuildIndex(texts, sizes, 1, C.byref(lcp), b"\x00")
    samples = L.BuildSampledLCPArray(tb, len(tb), lcp, 20, 0)
    ol = o.lcp
    assert samples == int((ol[:-1] != ol[1:]).sum())
    assert [lcp[i] for i in range(1, 200)] == [min(255, max(0, int(v))) for v in ol[1:200]]
    assert L.FMI_GetBWTSize() == len(tb) + 1
    # the scan's intervals position by position: FollowLetter / GetEnclosingLCPInterval as slamem.c:114-129 uses them
    q = tb[2950:3450]
    states = scan_states(L, q)
    top, bot, depth = 0, len(tb), 0
    for (_, i, st, sb, sd) in states:
        c = chr(q[i])
        while True:
            r, t2, b2 = o.follow_letter(c, top, bot)
            if r:
                top, bot, depth = t2, b2, depth + 1
                break
            d, top, bot = o.enclosing_interval(top, bot)
            if d == -1:
                depth = 0
                break
            depth = d
        assert (st, sb, sd) == (top, bot, depth), i
    rows = rng.integers(0, len(tb) + 1, size=300)
    assert [L.FMI_PositionInText(int(r)) for r in rows] == [o.position_in_text(int(r)) for r in rows]
    assert b"".join(L.FMI_GetCharAtBWTPos(int(r)) for r in rows).decode() == "".join(o.char_at_bwt_pos(int(r)) for r in rows)
    s = C.create_string_buffer(b"ACGTNNAC", 9)
    L.ReverseComplementSequence(s, 8)
    assert s.raw[:8] == b"GTNNACGT"
    starts = (C.c_uint * 3)(0, 101, 250)
    L.RefShim_SetMergedStarts(starts, 3)
    p = C.c_uint(260)
    assert L.GetSeqIdFromMergedSeqsPos(C.byref(p)) == 2 and p.value == 10
    L.FMI_FreeIndex()
    L.FreeSampledSuffixArray()
