#!/usr/bin/env python3
"""Measured ceiling for dependent random 64-B-line gathers on this GPU (bench support, not product).

Prints one JSON line per (table size, ILP): G lines/s and the equivalent GB/s at 64 B per line.  The
search kernel's rank queries have exactly this access shape, so its HBM-roofline fraction can be read
against this ceiling as well as against the 8 TB/s streaming peak.  Run under
`rocprofv3 --pmc FETCH_SIZE` with --calibrate to calibrate the counter on this access pattern.
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slamem_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calibrate", action="store_true", help="one launch per table size only (for --pmc runs)")
    ap.add_argument("--modes", action="store_true", help="address-path experiment: lane / quad / row cooperative reads")
    ap.add_argument("--inline", action="store_true", help="several dword reads inside one random 64/128-byte line per step")
    a = ap.parse_args()
    S = capi.synth_lib()
    S.slamem_gather_bench.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    S.slamem_gather_modes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    dev = torch.device("cuda:0")
    sink = torch.zeros(8, dtype=torch.int64, device=dev)
    lanes = 256 * 32 * 64 * 4  # 4 full waves of the chip
    if a.inline:
        S.slamem_gather_inline.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for mb in (800, 1600, 6400):
            table = torch.randint(0, 2 ** 31, (mb * (1 << 20) // 4,), dtype=torch.int32, device=dev)
            for occupancy_lanes in (lanes, lanes // 2):
                for line in (64, 128):
                    for loads in (1, 3, 9):
                        iters = 64
                        assert S.slamem_gather_inline(table.data_ptr(), mb << 20, occupancy_lanes, 4, loads, line, sink.data_ptr(), None) == 0
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        S.slamem_gather_inline(table.data_ptr(), mb << 20, occupancy_lanes, iters, loads, line, sink.data_ptr(), None)
                        e1.record()
                        torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1)
                        print(json.dumps({"table_MB": mb, "lanes": occupancy_lanes, "line_bytes": line, "dword_loads_per_line": loads,
                                          "ms": round(ms, 3), "Glines_per_s": round(occupancy_lanes * iters / ms / 1e6, 2)}), flush=True)
            del table
        return
    for mb in (2, 16, 50, 150, 1750, 8000):
        nblk = mb * (1 << 20) // 64
        table = torch.randint(0, 2 ** 31, (nblk * 16,), dtype=torch.int32, device=dev)
        if a.modes:
            for mode in (0, 1, 2, 3):
                iters = 64
                S.slamem_gather_modes(table.data_ptr(), nblk, lanes, 4, mode, sink.data_ptr(), None)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                S.slamem_gather_modes(table.data_ptr(), nblk, lanes, iters, mode, sink.data_ptr(), None)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                lane_acc = lanes * iters
                lines = lane_acc // (1 if mode == 0 else 4)
                print(json.dumps({"table_MB": mb, "mode": ["lane16B", "quad64B", "row256B", "pair128B"][mode], "ms": ms,
                                  "Glane_accesses_per_s": lane_acc / ms / 1e6, "Glines_per_s": lines / ms / 1e6}), flush=True)
            del table
            continue
        for ilp in ((1,) if a.calibrate else (1, 2, 4)):
            iters = 64 // ilp
            S.slamem_gather_bench(table.data_ptr(), nblk, lanes, 4, ilp, sink.data_ptr(), None)  # warm
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            S.slamem_gather_bench(table.data_ptr(), nblk, lanes, iters, ilp, sink.data_ptr(), None)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            lines = lanes * iters * ilp
            print(json.dumps({"table_MB": mb, "ilp": ilp, "lanes": lanes, "iters": iters, "lines": lines, "ms": ms,
                              "Glines_per_s": lines / ms / 1e6, "GBps_at_64B": lines * 64 / ms / 1e6}), flush=True)
        del table


if __name__ == "__main__":
    main()
