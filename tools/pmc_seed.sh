#!/bin/bash
# PMC passes over tools/seed_bench.py: per-launch means of the counters of the kernels whose name matches $2 (default k_seed_mems).
# usage: tools/pmc_seed.sh <tag> [kernel substring] [seed_bench args]
TAG=${1:-seed}; KERN=${2:-k_seed_mems}; shift 2 || true
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/seed_bench.py $EXTRA > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
EXTRA="$@"
run sq    SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2   SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT
run tcc   TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp   TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE
python3 - <<PY > $OUT/summary.json
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"] and "ILb1" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({"kernel": "$KERN", "launches": {k: len(v) for k, v in agg.items()}, "mean_per_launch": {k: sum(v) / len(v) for k, v in sorted(agg.items())}}, indent=1))
PY
cat $OUT/summary.json
