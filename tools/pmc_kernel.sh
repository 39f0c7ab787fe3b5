#!/bin/bash
# SQ counters (one pass) of the kernels whose name contains $1, for the program given behind it: per-launch means.
# usage: tools/pmc_kernel.sh <kernel substring> <tag> python3 tools/xyz.py args...
KERN=$1; TAG=$2; shift 2
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, len(v), round(sum(v) / len(v) / 1e6, 2), "M")
PY
