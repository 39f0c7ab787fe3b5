#!/usr/bin/env python3
"""Large-n check of the index build and the search (sizes where no CPU oracle fits in the time budget):
size-independent properties only -- every sampled MEM is a real match and maximal on both sides (checked
against the text itself), MEMs per read and structure statistics stay where theory puts them.

    tests/tools/scale_check.py <n> [reads] [min_len] [repeats]
        e.g. 1000000000, 2200000000 (> 2^31: 32-bit row arithmetic)
        repeats=1 plants the repeat model of SURVEY.md 8(d): 0.5 % of the text copied as 1-10 kbp segments with 1 %
        divergence (BASELINE.json configs[3] = 248000000 6250000 50 1 per GPU, configs[4] = 3100000000 12500000 20 1)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from slamem_amd import engine  # noqa: E402


def main():
    n = int(sys.argv[1])
    nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    repeats = len(sys.argv) > 4 and sys.argv[4] != "0"
    L = 150
    dev = torch.device("cuda:0")
    ref = engine.synth_reference(n, 42, dev)
    planted = engine.synth_plant_repeats(ref, 42) if repeats else 0  # seeded: the same text in every run (csrc/synth.hip)
    torch.cuda.synchronize()
    t0 = time.time()
    idx = engine.Index.build(ref, dev)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    tm = {k: round(v, 1) for k, v in engine.timings().items() if k.startswith("build_")}
    st = idx.sampled_lcp_stats()
    reads = engine.synth_reads(ref, 0, nreads, L, 0.02, 42, 50)
    offsets = torch.arange(nreads + 1, dtype=torch.int64, device=dev) * L
    m = idx.matcher(nreads, True, 8 * nreads, nreads * L)
    m.run(reads, offsets, min_len)
    engine.reset_timings()
    total = m.run(reads, offsets, min_len)
    tm_s = engine.timings()
    kms = tm_s["search_kernel_ms"]
    sst = engine.search_stats(m, reads, offsets, min_len) if os.environ.get("SCALE_STATS", "1") != "0" else {}
    total = m.run(reads, offsets, min_len)
    mems = m.mems[:total].cpu().numpy().view(np.uint32).astype(np.int64)
    boff = m.block_offsets[: 2 * nreads + 1].cpu().numpy()
    blk = np.repeat(np.arange(2 * nreads), np.diff(boff))
    ref_h = ref.cpu().numpy()
    reads_h = reads[: nreads * L].cpu().numpy().reshape(nreads, L)
    comp = np.zeros(256, dtype=np.uint8)
    comp[:] = np.arange(256)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    r, q, ln = mems[:, 0], mems[:, 1], mems[:, 2]
    assert (ln >= min_len).all() and (q + ln <= L).all() and (r + ln <= n).all()
    bad = 0
    sel = np.random.default_rng(0).choice(total, size=min(total, 50_000), replace=False)
    for i in sel:
        a, b, c, g = int(r[i]), int(q[i]), int(ln[i]), int(blk[i])
        rd = reads_h[g >> 1]
        if g & 1:
            rd = comp[rd[::-1]]
        ok = (ref_h[a:a + c] == rd[b:b + c]).all()
        ok = ok and (a == 0 or b == 0 or ref_h[a - 1] != rd[b - 1])
        ok = ok and (a + c == n or b + c == L or ref_h[a + c] != rd[b + c])
        bad += not ok
    out = {"n": n, "reads": nreads, "min_len": min_len, "planted_repeat_bp": planted, "build_wall_s": round(build_s, 3), "build_ms": tm, "sort_rounds": int(idx.info.sort_rounds),
           "max_lcp": int(idx.info.max_lcp), "index_GB": round(idx.info.arena_bytes / 1e9, 2),
           "samples_pct": round(100.0 * st["num_samples"] / (n + 1), 2), "mean_lcp": st["sum_lcp"] // (n + 1),
           "mems": int(total), "mems_per_read": round(total / nreads, 3), "checked": int(len(sel)), "bad": int(bad),
           "search_kernel_ms": round(kms, 2), "k8_ms": round(tm_s["k8_ms"], 2), "k8a_ms": round(tm_s["prefilter_ms"], 2),
           "Mreads_per_s": round(nreads / kms / 1e3, 2), "counters": sst}
    if sst:
        lines = sum(sst[k] for k in ("fm_lines_top", "fm_lines_bottom", "rec_lines_fail", "rec_lines_pend", "rec_lines_flush",
                                    "dir_sa_lines", "dir_group_loads", "dir_rec_lines", "jump_lines"))
        out["k8_lines_per_read"] = round(lines / nreads, 1)
        out["k8_Glines_per_s"] = round(lines / tm_s["k8_ms"] / 1e6, 2)
        out["ceiling_Glines_per_s"] = round(engine.random_line_ceiling(idx) / 1e9, 2)
    print(json.dumps(out))
    assert bad == 0


if __name__ == "__main__":
    main()
