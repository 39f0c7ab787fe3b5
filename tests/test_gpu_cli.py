"""The C front end (slaMEM-hip) on the golden fixtures: its *-mems.txt must be byte-identical to the file the
real reference wrote for the same command line."""
import os
import subprocess

import pytest

import hostlib
from golden_cases import CASES, MANIFEST, case_paths

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES)
def test_cli_output_file_is_byte_identical(case, tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    ref_fa, q_fa, exp_mems, _ = case_paths(case)
    out = str(tmp_path / "out-mems.txt")
    r = subprocess.run([exe] + MANIFEST[case]["opts"] + ["-o", out, ref_fa, q_fa] + MANIFEST[case].get("tail", []),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert open(out, "rb").read() == open(exp_mems, "rb").read()
    kind = b"MAMs" if "-mam" in MANIFEST[case].get("tail", []) else b"MEMs"
    assert b"> Done!" in r.stdout and b"> Saving " + kind + b" to <" in r.stdout


def _stable_stdout(raw):
    """stdout without the figures that differ from run to run (seconds, milliseconds)"""
    import re
    return re.sub(rb"[0-9]+\.[0-9]+ (s|ms)|(sort|BWT|LCP|links) [0-9.]+", b"T", raw).split(b"\n")


@pytest.mark.parametrize("case", CASES)
def test_cli_overlapped_loading_changes_nothing(case, tmp_path):
    """SLAMEM_OVERLAP_MB=0 sends every run through the loader thread (query files parsed in pieces beside the search,
    the main thread's lines held back meanwhile): the output file and stdout must be those of the sequential run."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    ref_fa, q_fa, exp_mems, _ = case_paths(case)
    out = str(tmp_path / "out-mems.txt")
    cmd = [exe] + MANIFEST[case]["opts"] + ["-o", out, ref_fa, q_fa] + MANIFEST[case].get("tail", [])
    # the plain run: everything loaded first, one process (no forked worker)
    plain = subprocess.run(cmd, stdout=subprocess.PIPE, env=dict(os.environ, SLAMEM_OVERLAP_MB="-1", SLAMEM_FOREGROUND="1"))
    assert plain.returncode == 0
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=dict(os.environ, SLAMEM_OVERLAP_MB="0"))
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert open(out, "rb").read() == open(exp_mems, "rb").read()
    assert _stable_stdout(r.stdout) == _stable_stdout(plain.stdout)


@pytest.mark.parametrize("overlap", ["-1", "0"])
def test_cli_without_a_valid_query_record(overlap, tmp_path):
    """slamem.c:648: "No query files provided" when no query record could be loaded -- also when the run only finds out
    after it has started (queries parsed beside the search): status 255, no output file left behind."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    ref_fa, _, _, _ = case_paths("acgt_l20_both")
    bad = tmp_path / "bad.fa"
    bad.write_bytes(b"this is not FASTA\nACGT\n")
    out = tmp_path / "o.txt"
    r = subprocess.run([exe, "-o", str(out), ref_fa, str(bad)], stdout=subprocess.PIPE, env=dict(os.environ, SLAMEM_OVERLAP_MB=overlap))
    assert r.returncode == 255
    assert b"> ERROR: No query files provided" in r.stdout and b"successfully loaded" not in r.stdout
    assert not out.exists()


def test_cli_default_output_name_and_batches(tmp_path):
    import shutil
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    ref_fa, q_fa, exp_mems, _ = case_paths("acgt_l20_both")
    shutil.copy(ref_fa, tmp_path / "myref.fa")
    env = dict(os.environ, SLAMEM_BATCH_MB="1")
    r = subprocess.run([exe, "-b", "-l", "20", str(tmp_path / "myref.fa"), q_fa], stdout=subprocess.PIPE, env=env)
    assert r.returncode == 0
    assert open(tmp_path / "myref-mems.txt", "rb").read() == open(exp_mems, "rb").read()


@pytest.mark.parametrize("case", CASES)
def test_cli_prints_the_reference_structure_statistics(case, tmp_path):
    """BuildSampledLCPArray's statistics (sample count, oversized samples, average / max LCP, oversized links,
    average / max link distance; lcparray.c:709-711, 999-1000) are reproduced from the per-row records."""
    import re
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    ref_fa, q_fa, _, exp_stdout = case_paths(case)
    r = subprocess.run([exe] + MANIFEST[case]["opts"] + ["-o", str(tmp_path / "o.txt"), ref_fa, q_fa] +
                       MANIFEST[case].get("tail", []), stdout=subprocess.PIPE)
    assert r.returncode == 0
    ours = r.stdout.decode(errors="replace")
    exp = open(exp_stdout, encoding="latin1").read()
    pats = [r"^:: [0-9.]+% samples \(\d+ of \d+\)$", r"^:: [0-9.]+% oversized samples \(\d+ of \d+\)$",
            r"^:: Average LCP value = -?\d+ \(max=\d+\)$", r"^:: [0-9.]+% oversized values \(\d+ of \d+\)$",
            r"^:: Average SV distance = [0-9.]+ \(max=\d+\)$"]
    for p in pats:
        a, b = re.search(p, ours, re.M), re.search(p, exp, re.M)
        assert a and b and a.group(0) == b.group(0), (p, a and a.group(0), b and b.group(0))


def test_cli_config2_full_size_output_hash(tmp_path):
    """BASELINE.json configs[1] end to end through the C front end: FASTA files of the SURVEY.md C.2 generator
    (their sha256 equal the ones recorded when the REAL reference was run, App. C.3) -> slaMEM-hip -l 20 ->
    the output file must be the reference's, byte for byte: 44,723,866 bytes, sha256 8f711ed6cd088ee1…."""
    import hashlib
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    root = hostlib.ROOT
    gen = os.path.join(root, "tools", "gen_synth.py")
    d = str(tmp_path)
    g = subprocess.run([sys.executable, gen, "100000000", "1000000", "150", "0.02", "42", "0", d], stdout=subprocess.PIPE)
    assert g.returncode == 0
    out = g.stdout.decode()
    assert "ref.fa f6bcf2e657079fdf" in out and "qry.fa 81ac1e6c4b633fa3" in out
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    r = subprocess.run([exe, "-l", "20", "-o", os.path.join(d, "out.txt"), os.path.join(d, "ref.fa"),
                        os.path.join(d, "qry.fa")], stdout=subprocess.PIPE)
    assert r.returncode == 0
    assert b"(total = 2412288, avg size = 55 bp)" in r.stdout
    h = hashlib.sha256()
    with open(os.path.join(d, "out.txt"), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    assert os.path.getsize(os.path.join(d, "out.txt")) == 44_723_866
    assert h.hexdigest().startswith("8f711ed6cd088ee1")
    # the same with the query file parsed in 16 MB pieces by the loader thread while the search runs, every piece released
    # as soon as its batches are formatted
    r2 = subprocess.run([exe, "-l", "20", "-o", os.path.join(d, "out2.txt"), os.path.join(d, "ref.fa"), os.path.join(d, "qry.fa")],
                        stdout=subprocess.PIPE, env=dict(os.environ, SLAMEM_OVERLAP_MB="0", SLAMEM_PIECE_MB="16", SLAMEM_RELEASE_EARLY_GB="0"))
    assert r2.returncode == 0
    h2 = hashlib.sha256()
    with open(os.path.join(d, "out2.txt"), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h2.update(chunk)
    assert h2.hexdigest() == h.hexdigest()
    assert _stable_stdout(r2.stdout.replace(b"out2.txt", b"out.txt")) == _stable_stdout(r.stdout)


def test_cli_rccl_replication_selftest(tmp_path):
    """The multi-GPU step of the C front end on a one-GPU box: the index arena goes through a real RCCL communicator
    (ncclCommInitAll + ncclBroadcast into a second arena), the search then runs on the broadcast copy."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    ref_fa, q_fa, exp_mems, _ = case_paths("acgt_l20_both")
    out = str(tmp_path / "o.txt")
    env = dict(os.environ, SLAMEM_REPLICATE_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "-b", "-l", "20", "-o", out, ref_fa, q_fa], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert b"replicated to 1 GPU (self-test copy) by RCCL broadcast ... OK" in r.stdout
    assert open(out, "rb").read() == open(exp_mems, "rb").read()
    # SLAMEM_GPUS=1 takes the sharding code path with one share
    r = subprocess.run([exe, "-b", "-l", "20", "-o", out, ref_fa, q_fa], stdout=subprocess.PIPE, env=dict(os.environ, SLAMEM_GPUS="1"))
    assert r.returncode == 0 and open(out, "rb").read() == open(exp_mems, "rb").read()
    # asking for more GPUs than exist fails loudly
    r = subprocess.run([exe, "-o", out, ref_fa, q_fa], stdout=subprocess.PIPE, env=dict(os.environ, SLAMEM_GPUS="64"))
    assert r.returncode == 255


def _sha(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def test_cli_n_gpu_schedule_on_logical_gpus(tmp_path):
    """The N-GPU path of the C front end (batch b on GPU b mod N, results collected in order; shard rule slamem.c:90-95:
    records are independent) with N = 2 and N = 3 LOGICAL GPUs mapped onto the one device of the test box
    (SLAMEM_LOGICAL_GPUS: every logical GPU gets its own RCCL-broadcast copy of the index and its own slamem_stream).
    20 Mbp reference, 200,000 reads in ~17 batches, -b -l 20: the output file must be the N = 1 file byte for byte."""
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    gen = os.path.join(hostlib.ROOT, "tools", "gen_synth.py")
    d = str(tmp_path)
    g = subprocess.run([sys.executable, gen, "20000000", "200000", "150", "0.02", "7", "50", d], stdout=subprocess.PIPE)
    assert g.returncode == 0
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    base = dict(os.environ, SLAMEM_BATCH_MB="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    shas = {}
    for n in (1, 2, 3):
        out = os.path.join(d, f"out{n}.txt")
        env = dict(base) if n == 1 else dict(base, SLAMEM_LOGICAL_GPUS=str(n))
        r = subprocess.run([exe, "-b", "-l", "20", "-o", out, os.path.join(d, "ref.fa"), os.path.join(d, "qry.fa")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
        if n > 1:
            assert b"replicated to %d logical GPUs by RCCL broadcast ... OK" % n in r.stdout
        shas[n] = (_sha(out), os.path.getsize(out))
    assert shas[1][1] > 10_000_000
    assert shas[2] == shas[1] and shas[3] == shas[1]


def test_cli_back_to_back_large_runs(tmp_path):
    """Two large jobs in a row on one GPU.  The first runs with SLAMEM_DETACH_TEARDOWN=1 (returns when its results are
    written; its worker still holds the 164 GB index, here for 7 more seconds: SLAMEM_TEST_LINGER_MS); the second starts
    at once, needs a 225 GB build peak (197 without the seed table) on the 288 GiB (309 GB) device, and must WAIT for the memory instead of failing
    (wait_for_hbm in host/main.c; the reference frees before it reports, slamem.c:208-216).  Both exit 0 with identical
    output files.  2.7 Gbp text, 100,000 reads, -b -l 20."""
    import time
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from slamem_amd import engine, synth
    n, nreads, L = 2_700_000_000, 100_000, 150
    ref = engine.synth_reference(n, 42, "cuda:0")
    reads = engine.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)[: nreads * L].cpu().numpy().reshape(nreads, L)
    ref_h = ref.cpu().numpy()
    del ref
    torch.cuda.empty_cache()
    d = str(tmp_path)
    synth.write_fasta_reference(os.path.join(d, "ref.fa"), ref_h)
    del ref_h
    synth.write_fasta_reads(os.path.join(d, "qry.fa"), reads)
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    cmd = lambda o: [exe, "-b", "-l", "20", "-o", os.path.join(d, o), os.path.join(d, "ref.fa"), os.path.join(d, "qry.fa")]
    t0 = time.time()
    r1 = subprocess.run(cmd("o1.txt"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                        env=dict(os.environ, SLAMEM_DETACH_TEARDOWN="1", SLAMEM_TEST_LINGER_MS="7000"))
    t1 = time.time()
    r2 = subprocess.run(cmd("o2.txt"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    t2 = time.time()
    assert r1.returncode == 0, (r1.stdout[-2000:], r1.stderr[-2000:])
    assert r2.returncode == 0, (r2.stdout[-2000:], r2.stderr[-2000:])
    assert b"Waiting for HBM" in r2.stderr and b"Waited " in r2.stderr, r2.stderr[-2000:]
    assert b"full layout" in r2.stdout
    assert _sha(os.path.join(d, "o1.txt")) == _sha(os.path.join(d, "o2.txt"))
    assert os.path.getsize(os.path.join(d, "o1.txt")) > 1_000_000
    try:
        with open(os.path.join(hostlib.ROOT, "gpurun_out", "back_to_back.txt"), "a") as f:
            f.write(f"2.7 Gbp, 100k reads: first run (detached teardown, worker lingers 7 s) returned after {t1 - t0:.2f} s; "
                    f"second run (default: returns when its memory is back) {t2 - t1:.2f} s; stderr of the second: "
                    f"{r2.stderr.decode(errors='replace').strip()}\n")
    except OSError:
        pass


def test_cli_detached_worker_when_the_front_is_pid_1(tmp_path):
    """SLAMEM_DETACH_TEARDOWN=1 forks a worker that must not mistake a front end that IS pid 1 (a container entry point)
    for a front end that has died (host/main.c front_is_gone).  Runs the binary as pid 1 of a new pid namespace when the
    box allows unprivileged namespaces; the detached mode itself is checked either way."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    exe = os.path.join(hostlib.HOST_DIR, "slaMEM-hip")
    ref_fa, q_fa, exp_mems, _ = case_paths("acgt_l20_both")
    out = str(tmp_path / "o.txt")
    env = dict(os.environ, SLAMEM_DETACH_TEARDOWN="1")
    r = subprocess.run([exe, "-b", "-l", "20", "-o", out, ref_fa, q_fa], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    assert r.returncode == 0 and open(out, "rb").read() == open(exp_mems, "rb").read()
    probe = subprocess.run(["unshare", "-U", "-r", "-p", "-f", "--mount-proc", "sh", "-c", "echo $$"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE)
    if probe.returncode != 0 or probe.stdout.strip() != b"1":
        pytest.skip("this box does not allow unprivileged pid namespaces (the pid-1 case needs one)")
    os.remove(out)
    r = subprocess.run(["unshare", "-U", "-r", "-p", "-f", "--mount-proc", exe, "-b", "-l", "20", "-o", out, ref_fa, q_fa],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    assert open(out, "rb").read() == open(exp_mems, "rb").read()
