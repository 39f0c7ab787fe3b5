"""python tools/compact_layout_cost.py [n ...]: the same reads against the FULL and the COMPACT index layout (include/slamem_hip.h:
SLAMEM_LAYOUT_*): arena size, build time, search step, K8a / K8 time, survivors of the prefilter.  100 Mbp: the headline reads
(10 M x 150 bp, -b -l 20); 3.1 Gbp: 5 M reads of the repeat-model text.  One JSON line per (n, layout)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import capi, engine
dev = "cuda:0"
for n in [int(a) for a in sys.argv[1:]] or [100_000_000, 3_100_000_000]:
    M, L = (10_000_000 if n <= 300_000_000 else 5_000_000), 150
    ref = engine.synth_reference(n, 42, dev)
    if n > 200_000_000:
        engine.synth_plant_repeats(ref, 42)
    reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
    offsets = torch.arange(M + 1, dtype=torch.int64, device=dev) * L
    want = None
    for name, layout in (("full", capi.LAYOUT_FULL), ("compact", capi.LAYOUT_COMPACT)):
        torch.cuda.synchronize()
        t0 = time.time()
        idx = engine.Index.build(ref, dev, layout=layout)
        torch.cuda.synchronize()
        build_s = time.time() - t0
        bt = engine.timings()["build_total_ms"]
        m = idx.matcher(M, True, 4 * M + 1024, M * L)
        total = m.run(reads, offsets, 20)
        engine.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.run(reads, offsets, 20)
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / 5 * 1e3
        tm = engine.timings()
        st = engine.search_stats(m, reads, offsets, 20)
        import hashlib
        dig = hashlib.sha256(m.mems[:total].cpu().numpy().tobytes()).hexdigest()[:16]
        want = want or dig
        a, p = engine.build_bytes(n, layout)
        print(json.dumps({"n": n, "layout": name, "arena_GB": idx.info.arena_bytes / 1e9, "bytes_per_letter": idx.info.arena_bytes / n,
                          "build_peak_GB_estimate": p / 1e9, "build_wall_s": round(build_s, 3), "build_device_ms": round(bt, 1),
                          "reads": M, "mems": total, "same_output_as_full": dig == want, "step_ms": round(step, 3),
                          "k8_ms": round(tm["k8_ms_sum"] / 5, 3), "k8a_ms": round(tm["prefilter_ms_sum"] / 5, 3),
                          "survivors": st["survivors"], "filter_k": int(idx.info.filter_k)}), flush=True)
        del m
        idx.close()
        torch.cuda.empty_cache()
    del ref, reads, offsets
    torch.cuda.empty_cache()
