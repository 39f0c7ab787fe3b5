"""The verifier's GPU text scan (tests/verifier/mem_verifier.hip) against its numpy restatement, and the verifier
applied to the headline workload's text: the engine's MEMs for sampled reads == the definitional set."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (there is no CPU path)")
    from slamem_amd import engine
    return engine


@pytest.mark.parametrize("k", [1, 5, 20, 21])
def test_scan_kernel_equals_numpy_restatement(eng, k):
    import torch
    import mem_verifier as mv
    rng = np.random.default_rng(k)
    n = 3_000_017
    text = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n, p=[0.24, 0.24, 0.24, 0.24, 0.04])
    text[1000:1400] = ord("N")
    text[n - 7:] = np.frombuffer(b"acgtnAC", dtype=np.uint8)  # lower case and the last window of the text
    starts = np.concatenate([rng.integers(0, n - k + 1, size=5000), [0, n - k, 1000, 1390]])
    keys = np.unique(mv.keys_at(text, starts.astype(np.int64), k))
    want = mv.scan_text_numpy(text, k, keys)
    got = mv.scan_text_gpu(torch.from_numpy(text).to("cuda:0"), k, keys)
    assert np.array_equal(want, got)
    assert len(got) >= len(starts) - 10


def test_headline_text_sampled_reads_equal_the_definitional_set(eng):
    """100 Mbp reference of configs[1]/[2], its first 400,000 reads, -b -l 20: for 3,000 sampled reads the engine's MEMs
    equal, as a set, every maximal match >= 20 found by streaming the text against the reads' 20-mers."""
    import torch
    import mem_verifier as mv
    n, nreads, L, min_len = 100_000_000, 400_000, 150, 20
    dev = "cuda:0"
    ref = eng.synth_reference(n, 42, dev)
    reads = eng.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)
    offsets = torch.arange(nreads + 1, dtype=torch.int64, device=dev) * L
    idx = eng.Index.build(ref, dev)
    m = idx.matcher(nreads, True, 8 * nreads, nreads * L)
    total = m.run(reads, offsets, min_len)
    mems = m.mems[:total].cpu().numpy().view(np.uint32)
    boff = m.block_offsets[: 2 * nreads + 1].cpu().numpy()
    rows = np.empty((total, 4), dtype=np.uint32)
    rows[:, 0] = np.repeat(np.arange(2 * nreads, dtype=np.uint32), np.diff(boff))
    rows[:, 1] = mems[:, 0] + 1
    rows[:, 2] = mems[:, 1] + 1
    rows[:, 3] = mems[:, 2]
    sample = np.random.default_rng(5).choice(nreads, size=3000, replace=False)
    res = mv.verify_sample(ref.cpu().numpy(), ref, reads[: nreads * L].cpu().numpy().reshape(nreads, L), sample, rows,
                           min_len, True)
    assert res["missing_count"] == 0 and res["extra_count"] == 0 and res["engine_duplicates"] == 0, res
    assert res["definitional_mems"] > 6000
    idx.close()
