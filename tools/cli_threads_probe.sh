#!/bin/bash
# The command line against the number of host threads (SLAMEM_THREADS) on a box whose cgroup gives 16 CPUs of 256: parse,
# format and the whole run; and how many text buffers were recycled.
D=/tmp/c3; mkdir -p $D
python tools/gen_synth.py 100000000 10000000 150 0.02 42 50 $D > $D/gen.log
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
for t in 32 16 8 32; do
  sleep 2
  T0=$(date +%s.%N)
  SLAMEM_THREADS=$t SLAMEM_TIMING=1 slamem_amd/host/slaMEM-hip -b -l 20 -o $D/o.txt $D/ref.fa $D/qry.fa > /dev/null 2> $D/err.txt
  T1=$(date +%s.%N)
  echo "threads $t: wall $(python3 -c "print('%.3f' % ($T1-$T0))") $(grep -o 'load [0-9.]* s\|pieces beside the search in [0-9.]* s\|index build of [0-9.]* s\|overlapped with formatting) [0-9.]* s\|format [0-9.]* s\|text buffers: [0-9]* recycled, [0-9]* fresh\|total [0-9.]* s' $D/err.txt | tr '\n' ';')"
done
