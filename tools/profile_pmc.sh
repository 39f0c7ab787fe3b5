#!/bin/bash
# PMC passes over bench.py (one timed step, no CPU baseline); one counter group per pass, never combined with
# the trace domains gpurun refuses.  Summaries land in gpurun_out/prof/<tag>; tools/summarize_pmc.py reads them.
# usage: tools/profile_pmc.sh <tag> [extra bench args]
set -e
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  timeout -k 10 280 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stats --no-host-leg $EXTRA > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
EXTRA="$@"
run sq    SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2   SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR
run tcp   TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum
run tcp2  TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tcc   TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run tcc2  TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_READ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run ta    TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr GRBM_GUI_ACTIVE
python3 tools/summarize_pmc.py $OUT > $OUT/summary.json
cat $OUT/summary.json
