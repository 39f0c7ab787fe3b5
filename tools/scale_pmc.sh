#!/bin/bash
# Why is the search slower per read on big texts?  Same reads / settings at 100 Mbp and 3.1 Gbp: lines per read (kernel
# counters), lines/s against the ceiling probed on that arena, and the translation-miss / request counters of rocprofv3.
export TMPDIR=/tmp; mkdir -p gpurun_out/prof
for n in 100000000 3100000000; do
  python tests/tools/scale_check.py $n 5000000 20 1 2>&1 | grep -v amdgpu.ids | tail -1
  SCALE_STATS=0 timeout -k 10 280 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_EA0_RDREQ_sum TCC_HIT_sum --output-format csv -d gpurun_out/prof/scale_$n -- python3 tests/tools/scale_check.py $n 5000000 20 1 > gpurun_out/prof/scale_$n.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof/scale_$n/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_find_mems_v3" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print($n, {k: sum(v) / len(v) for k, v in sorted(agg.items())})
PY
done
