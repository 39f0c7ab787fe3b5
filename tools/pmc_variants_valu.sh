#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS per launch of k_seed_mems for each variant library (tools/variants.sh build ...): which part of
# the kernel issues how many instructions (diagnostic variants that stop after a stage).
# usage: tools/pmc_variants_valu.sh name1 name2 ...
export TMPDIR=/tmp
R=$PWD
for name in "$@"; do
  if [ "$name" = default ]; then lib=$R/slamem_amd/csrc/libslamem_hip.so; else lib=$R/slamem_amd/csrc/variants/libslamem_hip_$name.so; fi
  OUT=$R/gpurun_out/prof/valu_$name
  mkdir -p $OUT
  SLAMEM_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT -- python3 tools/seed_bench.py > $OUT.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_seed_mems" in r["Kernel_Name"] and "ILb1" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(agg["SQ_WAVES"]) / max(1, len(agg["SQ_WAVES"]))
print("$name", {k: round(sum(v) / len(v) / max(w, 1), 1) for k, v in sorted(agg.items()) if k != "SQ_WAVES"}, "per wave; waves", w)
PY
done
