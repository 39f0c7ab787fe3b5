#!/bin/bash
# usage: tools/pmc_once.sh <outdir-name> "<counters>" <program...>   -- one rocprofv3 --pmc pass, CSV
NAME=$1; CTRS=$2; shift 2
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof/$NAME
timeout -k 10 280 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/prof/$NAME -- "$@" > $R/gpurun_out/prof/$NAME.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/prof/$NAME/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("k_find_mems", "k_gather")):
            agg[(r["Kernel_Name"][:40], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, len(v), max(v))
PY
