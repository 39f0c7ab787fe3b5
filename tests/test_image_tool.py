"""The "-v <mems_file>" image tool of the command line (slamem.c:354-452, graphics.c, bitmap.c -> slamem_amd/host/mem_image.c):
host C only, so these run without a GPU.  The picture file and stdout are compared BYTE FOR BYTE with what the real reference
wrote for the same inputs (tests/golden/image/, made by tests/golden/make_image_golden.py)."""
import os
import struct
import subprocess

import pytest

import image_cases
from golden_cases import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "slamem_amd", "host", "slaMEM-hip")
MANIFEST = image_cases.manifest()


def run_tool(where, opts=()):
    r = subprocess.run([EXE, *opts, "-v", "mems.txt", "ref.fa", "q.fa"], cwd=where, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=120)
    return r.returncode, r.stdout


def decode_bmp(data):
    """8-bit BMP, plain or BI_RLE8 -> (width, height, palette bytes, rows bottom-up); an independent decoder, not the tool's"""
    assert data[:2] == b"BM"
    size, _, _, offset = struct.unpack_from("<IHHI", data, 2)
    hdr, w, h, planes, bpp, comp, datasize, _, _, ncol, _ = struct.unpack_from("<IIIHHIIIIII", data, 14)
    assert (hdr, planes, bpp) == (40, 1, 8) and size == len(data) == offset + datasize and offset == 54 + 4 * ncol
    d = data[offset:]
    if comp == 0:
        return w, h, data[54:offset], [d[r * w:(r + 1) * w] for r in range(h)]
    assert comp == 1
    rows, cur, i = [], bytearray(), 0
    while True:
        c, v = d[i], d[i + 1]
        i += 2
        if c:
            cur += bytes([v]) * c
        elif v == 0:
            rows.append(bytes(cur))
            cur = bytearray()
        elif v == 1:
            break
        else:
            assert v >= 3
            cur += d[i:i + v]
            i += v + (v & 1)
    assert not cur and all(x == 0 for x in d[i:]) and len(d) - i < 4
    return w, h, data[54:offset], rows


@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_image_tool_equals_the_reference_byte_for_byte(case, tmp_path):
    entry = MANIFEST[case]
    image_cases.write_inputs(case, entry, str(tmp_path))
    rc, out = run_tool(str(tmp_path), entry["opts"])
    assert rc == entry["rc"], out.decode(errors="replace")[-400:]
    assert out == open(os.path.join(image_cases.IMAGE_GOLDEN, case, "expected-stdout.txt"), "rb").read()
    bmp = tmp_path / "mems.bmp"
    assert bmp.exists() == entry["image"]
    if entry["image"]:
        mine, ref = bmp.read_bytes(), open(os.path.join(image_cases.IMAGE_GOLDEN, case, "expected.bmp"), "rb").read()
        assert mine == ref
        w, h, palette, rows = decode_bmp(mine)
        n = 1 + len(entry["queries"])
        assert (w, h) == (1024, 24 + 16 + 30 * n + 12 * (n - 1)) and len(rows) == h and all(len(r) == w for r in rows)
        assert palette[:12] == bytes([255, 255, 255, 0, 0, 0, 0, 0, 222, 222, 222, 0]) and len(palette) == 4 * 255


def test_picture_content_follows_the_mems(tmp_path):
    """What the bytes MEAN, through an independent decoder: a grey track where no MEM lies, the reference bar's colour of the
    matching columns where one does (forward MEMs from the bar's upper half, reverse ones from its lower half, mirrored on the
    query), the longer MEM on top."""
    entry = {"reference": [["ref", 98700]], "queries": [["q0", 98700]]}  # 100 letters per column (987 columns for 5 digits)
    image_cases.write_fasta(str(tmp_path / "ref.fa"), [tuple(entry["reference"][0])])
    image_cases.write_fasta(str(tmp_path / "q.fa"), [tuple(entry["queries"][0])])
    (tmp_path / "mems.txt").write_text(">q0\n50001\t10001\t5000\n50001\t12001\t1000\n>q0 Reverse\n20001\t1\t3000\n")
    rc, out = run_tool(str(tmp_path))
    assert rc == 0 and b"(2 MEMs)" in out and b"(1 MEMs)" in out
    w, h, palette, rows = decode_bmp((tmp_path / "mems.bmp").read_bytes())
    rows = rows[::-1]                     # top row first
    bar_fwd, bar_rev, track = rows[28 + 5], rows[28 + 16 + 5], rows[28 + 42 + 10]
    grey = 2
    assert track[10 + 50] == grey and track[10 + 99] == grey and track[10 + 150] == grey
    for col in range(100, 150):            # query 10,000..14,999 <-> reference 50,000..54,999, forward
        assert track[10 + col] == bar_fwd[10 + 500 + (col - 100)]
    # the 1,000-letter MEM at query 12,000 (reference 50,000 as well) lies under the longer one: nothing of it shows
    assert track[10 + 120] == bar_fwd[10 + 520]
    for col in range(957, 987):            # reverse strand: query positions count from the other end
        assert track[10 + col] == bar_rev[10 + 200 + (col - 957)]
    assert track[10 + 956] == grey and track[10 + 987] == 1  # the track's black frame


def test_long_stretches_without_equal_neighbours(tmp_path):
    """983 neighbouring columns, no two alike: the reference's run-length coder wraps its 8-bit counter at 255, skips a byte of
    the picture each time and reads behind its pixel buffer at the end (two runs of the reference differ there), so there is no
    fixture; the tool follows the reference's decisions up to that point (checked in the build container against the reference:
    the files are equal up to the last row's tail) and must still write a well-formed file that decodes to 1024-wide rows."""
    image_cases.write_fasta(str(tmp_path / "ref.fa"), [("ref", 983000)])
    image_cases.write_fasta(str(tmp_path / "q.fa"), [("q0", 983000)])
    with open(tmp_path / "mems.txt", "w") as f:
        f.write(">q0\n")
        for j in range(983):
            f.write("%d\t%d\t1000\n" % (1 if j % 2 == 0 else 500001, 1 + 1000 * j))
    rc, out = run_tool(str(tmp_path))
    assert rc == 0 and out.endswith(b"> Saving image to <mems.bmp> ... OK\n> Done!\n")
    w, h, palette, rows = decode_bmp((tmp_path / "mems.bmp").read_bytes())
    assert (w, h) == (1024, 112) and len(rows) == h and all(len(r) == w for r in rows)


def test_image_of_a_golden_case_of_the_search(tmp_path):
    """the search's own output format goes in unchanged: the reference's MEMs file of a golden case of the main path"""
    src = os.path.join(GOLDEN, "acgt_l20_both")
    for name, to in (("ref.fa", "ref.fa"), ("q.fa", "q.fa"), ("expected-mems.txt", "mems.txt")):
        (tmp_path / to).write_bytes(open(os.path.join(src, name), "rb").read())
    rc, out = run_tool(str(tmp_path))
    assert rc == 0 and out.count(b" MEMs)\n") == 12
    import hashlib
    assert hashlib.sha256((tmp_path / "mems.bmp").read_bytes()).hexdigest() == \
        "cc08bfe75f081c60ba0ce6f6c64993079116c5c0006aa51b3a361a7123da305f"  # the reference's picture of these files (build container)


def test_random_inputs_against_the_reference_binary_when_it_is_here(tmp_path):
    """Where oracle/_ref/slaMEM exists (the build container, and boxes the built tree travels to), twelve seeded random inputs --
    1 to 20 queries shorter and longer than the reference, odd names, few / diagonal / scattered / per-column MEMs, names the
    tool must refuse -- go through both programs: same exit status, same stdout, same picture bytes.  Without the binary the
    committed fixtures above are the check (the test then only exercises the tool)."""
    import random
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "slaMEM")
    have_ref = os.path.exists(ref_bin)
    for seed in range(12):
        rng = random.Random(1000 + seed)
        d = tmp_path / ("c%d" % seed)
        d.mkdir()
        rn = rng.choice([300, 5000, 123456, 400000, 2500])
        queries = []
        for k in range(rng.choice([1, 2, 3, 5, 9, 20])):
            ql = rng.choice([50, 400, rn // 2, rn, rn * 2 if rn < 200000 else rn // 3, 1234])
            name = rng.choice(["q%d" % k, "read_%d some description here that is long" % k, "x" * rng.randint(1, 300) + str(k),
                               "Reverse%d" % k, "we|rd~{}[]%d" % k, "tab\tname%d" % k])
            queries.append((name, ql))
        image_cases.write_fasta(str(d / "ref.fa"), [(rng.choice(["ref", "gi|123|some reference, complete genome", "R" * 200]), rn)])
        image_cases.write_fasta(str(d / "q.fa"), queries)
        mode = rng.choice(["scatter", "diag", "few", "noise"])
        with open(d / "mems.txt", "w") as f:
            for name, ql in queries:
                for strand in (0, 1):
                    f.write(">%s%s\n" % (name, " Reverse" if strand else ""))
                    if mode == "noise":
                        step = max(1, ql // 1000)
                        for qp in range(1, ql + 1, step):
                            ln = max(1, min(rng.randint(1, step), ql - qp + 1, rn))
                            f.write("%d\t%d\t%d\n" % (rng.randint(1, rn - ln + 1), qp, ln))
                        continue
                    for _ in range(rng.randint(0, 5) if mode == "few" else rng.randint(0, 400)):
                        ln = max(1, min(rng.randint(1, max(1, min(ql, rn) // rng.choice([1, 3, 10, 100]))), ql, rn))
                        qp = rng.randint(1, ql - ln + 1)
                        rp = min(max(1, qp + rng.randint(-5, 5)), rn - ln + 1) if mode == "diag" else rng.randint(1, rn - ln + 1)
                        f.write("%d\t%d\t%d\n" % (rp, qp, ln))
        rc, out = run_tool(str(d))
        assert rc in (0, 255), out[-300:]
        mine = (d / "mems.bmp").read_bytes() if (d / "mems.bmp").exists() else None
        assert (mine is not None) == (rc == 0)
        if mine is not None:
            w, h, _, rows = decode_bmp(mine)
            assert w == 1024 and len(rows) == h and all(len(r) == w for r in rows)
            (d / "mems.bmp").unlink()
        if have_ref:
            r = subprocess.run([ref_bin, "-v", "mems.txt", "ref.fa", "q.fa"], cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=120)
            assert (r.returncode, r.stdout) == (rc, out), seed
            theirs = (d / "mems.bmp").read_bytes() if (d / "mems.bmp").exists() else None
            assert theirs == mine, seed


def test_file_writer_round_trips_any_picture(tmp_path):
    """The writer behind the tool (slh_write_bmp8 of libslamem_host.so) on pictures the MEM map never makes, read back with the
    independent decoder: runs, lone bytes between runs, literal stretches of odd and even length, a row of one colour -- and
    NOISE, which does not compress: the file is then written plain (compression 0), as the reference does (bitmap.c:617)."""
    import ctypes as C
    import numpy as np
    import hostlib
    L = C.CDLL(os.path.join(hostlib.HOST_DIR, "libslamem_host.so"))
    L.slh_write_bmp8.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(12)
    w, h = 1024, 40
    pics = {}
    a = np.zeros((h, w), dtype=np.uint8)
    a[3, 100:400] = 7; a[4, 0:240:2] = 9; a[5, 10:13] = [1, 2, 3]; a[6, 10:14] = [1, 2, 3, 4]; a[7, :] = 200
    a[8, 5] = 1; a[8, 6:8] = 2; a[8, 8] = 3; a[8, 9:11] = 4                       # lone bytes between pairs
    a[9, 0:200] = (np.arange(200) % 7 + 1).astype(np.uint8)                        # 200 neighbours, no two alike
    a[7, -4:] = 0
    pics["shapes"] = a
    pics["runs"] = np.repeat(rng.integers(0, 255, size=(h, w // 8), dtype=np.uint8), 8, axis=1)
    # (stretches of 255 and more neighbours without two alike are the reference's broken regime -- see the test above; the
    #  noise here has an equal pair every 100 columns so that it stays out of it)
    noise = rng.integers(0, 250, size=(h, w), dtype=np.uint8)
    noise[:, 50::100] = noise[:, 49::100][:, : noise[:, 50::100].shape[1]]
    noise[:, -4:] = 0   # (a row that ends inside a stretch of unequal neighbours is the other way into that regime)
    pics["noise"] = noise
    pics["noise_rows"] = np.where(np.arange(h)[:, None] % 4 == 0, noise, 0).astype(np.uint8)
    for name, pic in pics.items():
        path = str(tmp_path / (name + ".bmp"))
        assert L.slh_write_bmp8(path.encode(), w, h, np.ascontiguousarray(pic).ctypes.data) == 1
        data = open(path, "rb").read()
        ww, hh, palette, rows = decode_bmp(data)
        assert (ww, hh) == (w, h) and len(rows) == h
        assert np.array_equal(np.frombuffer(b"".join(rows[::-1]), dtype=np.uint8).reshape(h, w), pic), name
        comp = struct.unpack_from("<I", data, 30)[0]
        assert comp == (0 if name == "noise" else 1), (name, comp, len(data))
