#!/bin/bash
# One PMC pass (SQ counters) of tools/seed_bench.py per library variant: per-launch means for k_seed_mems.
# usage: tools/pmc_variants.sh <tag> name1 name2 ...   (names as in tools/variants.sh; `default` = the product's library)
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
for name in "$@"; do
  if [ "$name" = default ]; then lib=$R/slamem_amd/csrc/libslamem_hip.so; else lib=$R/slamem_amd/csrc/variants/libslamem_hip_$name.so; fi
  OUT=$R/gpurun_out/prof/$TAG/$name
  mkdir -p $OUT
  SLAMEM_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python3 tools/seed_bench.py > $OUT.log 2>&1 || echo "pass $name failed"
  python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_seed_mems" in r["Kernel_Name"] and "ILb1" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$name", json.dumps({k: round(sum(v) / len(v) / 1e6, 1) for k, v in sorted(agg.items())}))
PY
  grep ms_per_step $OUT.log | cut -c1-200
done
