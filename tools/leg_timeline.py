"""python tools/leg_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> : the LAST pass of the host-to-host
leg (rocprofv3 --kernel-trace --memory-copy-trace -- python3 tools/trace_leg.py) as a per-stream timeline of the kernels that
matter, with the gaps between consecutive K8 launches."""
import csv, glob, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/*_kernel_trace.csv")[0])))
cp = list(csv.DictReader(open(glob.glob(d + "/*_memory_copy_trace.csv")[0])))
k8 = [r for r in rows if "k_find_mems_v3" in r["Kernel_Name"]]
half = len(k8) // 2
start = int(k8[half]["Start_Timestamp"]) - 6_000_000
sel = [r for r in rows if int(r["Start_Timestamp"]) >= start]
base = min(int(r["Start_Timestamp"]) for r in sel)


def short(n):
    return n.replace("void ", "").replace("slamem::", "").split("(")[0][:30]


prev_end = None
for r in sel:
    n = short(r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
    if (e - s) > 60_000 or "k_find" in n or "k_prefilter" in n:
        gap = ""
        if "k_find" in n:
            gap = "   gap since previous K8 end %.3f" % ((s - prev_end) / 1e6) if prev_end is not None else ""
            prev_end = e
        print("%9.3f %9.3f %7.3f q%-2s st%-2s %-30s grid %-9s%s" % (s / 1e6, e / 1e6, (e - s) / 1e6, r["Queue_Id"], r["Stream_Id"], n, r["Grid_Size_X"], gap))
print("copies")
for c in cp:
    if int(c["Start_Timestamp"]) >= start:
        s, e = int(c["Start_Timestamp"]) - base, int(c["End_Timestamp"]) - base
        if e - s > 100_000:
            print("%9.3f %9.3f %7.3f %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, c["Direction"]))
