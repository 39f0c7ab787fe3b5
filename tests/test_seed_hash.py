"""The seed table's key functions on the host (no GPU): tests/hostcheck/seed_hash_check.cpp includes slamem_amd/csrc/common.h
and checks that the hashes are bijections of their key widths (32-bit form and the 64-bit form of seeds of 17 / 18 letters) and
that a k-mer and its reverse complement are filed in the same bucket under the same tag with opposite orientation bits;
tests/hostcheck/seed_form_check.cpp checks the rule that picks the seed kernel's form for the next batch."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_hostcheck(tmp_path, name, says):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = str(tmp_path / name)
    src = os.path.join(ROOT, "tests", "hostcheck", name + ".cpp")
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-o", exe, src], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and says in r.stdout, r.stdout.decode()[-500:]


def test_seed_hashes_are_bijections_and_strand_symmetric(tmp_path):
    run_hostcheck(tmp_path, "seed_hash_check", b"seed hash ok")


def test_the_note_about_the_seed_kernels_form(tmp_path):
    """slamem::seed_words_next (common.h): which form of the seed kernel the next batch against an index takes, from what the
    last one counted -- a run of batches through the rule (tests/hostcheck/seed_form_check.cpp)."""
    run_hostcheck(tmp_path, "seed_form_check", b"seed form ok")
