"""python tools/repeat_load.py [n] [reads_l50] [reads_l20]: the search on a chr1-sized text under the GENOME-LIKE repeat load
(slamem_amd/synth.py::plant_genome_like: 10^5 copies of a 300 bp family at 5-15 % divergence, a 171 bp x 10^4 satellite
array, a 30 Mbp block of N) beside SURVEY's mild model, at -l 50 and -l 20: kernel times, MEMs per read, the share of MEMs
that went through the overflow list, capacity retries.  One JSON line per run (profiles/r03_repeat_load.jsonl)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import capi, engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 248_000_000
R50 = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
R20 = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
L, dev = 150, "cuda:0"


def run(model, idx, ref, lay, reads_n, min_len):
    avoid = (lay["n_block_at"], lay["n_block_letters"]) if lay else (0, 0)
    reads = engine.synth_reads(ref, 0, reads_n, L, 0.02, 42, 50, avoid=avoid)
    offsets = torch.arange(reads_n + 1, dtype=torch.int64, device=dev) * L
    cap, retries = 8 * reads_n, 0
    while True:
        m = idx.matcher(reads_n, True, cap, reads_n * L)
        try:
            total = m.run(reads, offsets, min_len)
            break
        except capi.SlamemError as e:
            if e.code != capi.SLAMEM_ERR_CAPACITY:
                print(json.dumps({"model": model, "min_len": min_len, "reads": reads_n, "error": str(e)}), flush=True)
                return
            cap, retries = int(m.last_total * 1.05) + 1024, retries + 1
            del m
    engine.reset_timings()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.run(reads, offsets, min_len)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / 3 * 1e3
    tm = engine.timings()
    st = engine.search_stats(m, reads, offsets, min_len)
    boff = m.block_offsets[: 2 * reads_n + 1]
    per_block = (boff[1:] - boff[:-1])
    print(json.dumps({"model": model, "n": n, "min_len": min_len, "reads": reads_n, "mems": total, "mems_per_read": total / reads_n,
                      "max_mems_per_strand": int(per_block.max().item()), "strands_over_1000": int((per_block > 1000).sum().item()),
                      "step_ms": round(step_ms, 3), "k8s_ms": round(tm["seed_ms_sum"] / 3, 3), "k8_ms": round(tm["k8_ms_sum"] / 3, 3), "k8a_ms": round(tm["prefilter_ms_sum"] / 3, 3),
                      "seed": {k: st[k] for k in ("seed_reads", "seed_windows", "seed_compares", "seed_mems", "seed_strands_left", "seed_left_why")},
                      "ms_per_million_reads": round(step_ms / reads_n * 1e6, 2), "capacity_retries": retries,
                      "overflow_share": st["overflow_records"] / max(1, total), "survivors": st["survivors"], "items": st["items"],
                      "lines_per_read": (st["fm_lines_top"] + st["fm_lines_bottom"] + st["rec_lines_fail"] + st["rec_lines_pend"]
                                         + st["dir_sa_lines"] + st["dir_group_loads"] + st["dir_rec_lines"] + st["jump_lines"]) / reads_n,
                      "enum_jobs": st["enum_jobs"], "enum_row_steps": st["enum_row_steps"], "enum_levels": st["enum_levels"],
                      "wave_trips": st["wave_trips"], "lane_trips": st["lane_trips"],
                      "k8_us_until_list_empty": st["k8_us_until_list_empty"], "k8_us_tail": st["k8_us_tail"], "k8_wave_us_sum": st["k8_wave_us_sum"], "enum_wave_us_sum": st["enum_wave_us"], "max_lcp": int(idx.info.max_lcp)}), flush=True)
    del m


for model in ("survey_8d", "genome_like"):
    ref = engine.synth_reference(n, 42, dev)
    engine.synth_plant_repeats(ref, 42)
    lay = None
    if model == "genome_like":
        engine.synth_plant_genome_like(ref, 42)
        c, s, sa, na, nl = synth.genome_like_layout(n)
        lay = {"n_block_at": na, "n_block_letters": nl}
    t0 = time.time()
    idx = engine.Index.build(ref, dev)
    torch.cuda.synchronize()
    print(json.dumps({"model": model, "n": n, "build_s": round(time.time() - t0, 3), "build_ms": {k: round(v, 1) for k, v in engine.timings().items() if k.startswith("build_")},
                      "arena_GB": idx.info.arena_bytes / 1e9, "max_lcp": int(idx.info.max_lcp), "sort_rounds": int(idx.info.sort_rounds)}), flush=True)
    run(model, idx, ref, lay, R50, 50)
    run(model, idx, ref, lay, R20, 20)
    idx.close()
    del ref
    torch.cuda.empty_cache()
