#!/bin/bash
# BASELINE.json configs[2] end to end through the command line: FASTA files in, *-mems.txt out; three timed runs, 2 s apart.
# Default (round 3): one process; the command returns when its GPU memory is back (process wall = what a scheduler sees).
# Then the same with SLAMEM_DETACH_TEARDOWN=1 (opt-in: returns when the results are written, a forked worker gives 6 GB of
# HBM and the pinned buffers back behind the caller's back for another ~0.35 s): BOTH walls are reported.
set -e
D=${1:-/tmp/c3}
mkdir -p $D
if [ ! -f $D/qry.fa ]; then python tools/gen_synth.py 100000000 10000000 150 0.02 42 50 $D > $D/gen.log; fi
for i in 1 2 3; do
  sleep ${SPACING:-2}
  T0=$(date +%s.%N)
  SLAMEM_TIMING=1 slamem_amd/host/slaMEM-hip -b -l 20 -o $D/out-mems.txt $D/ref.fa $D/qry.fa > $D/stdout.txt 2> $D/stderr.txt || { tail -5 $D/stdout.txt; cat $D/stderr.txt; exit 1; }
  T1=$(date +%s.%N)
  python3 -c "print('process wall %.3f s' % ($T1 - $T0))"
  grep -h "timing" $D/stderr.txt
done
for i in 1 2; do
  sleep ${SPACING:-2}
  T0=$(date +%s.%N)
  SLAMEM_OVERLAP_MB=-1 SLAMEM_TIMING=1 slamem_amd/host/slaMEM-hip -b -l 20 -o $D/out-seq.txt $D/ref.fa $D/qry.fa > $D/stdout_seq.txt 2> $D/stderr_seq.txt
  T1=$(date +%s.%N)
  python3 -c "print('sequential loading: process wall %.3f s' % ($T1 - $T0))"
  grep -h "timing" $D/stderr_seq.txt
done
for i in 1 2 3; do
  sleep ${SPACING:-2}
  T0=$(date +%s.%N)
  SLAMEM_DETACH_TEARDOWN=1 slamem_amd/host/slaMEM-hip -b -l 20 -o $D/out-det.txt $D/ref.fa $D/qry.fa > $D/stdout_det.txt 2> $D/stderr_det.txt
  T1=$(date +%s.%N)
  python3 -c "print('SLAMEM_DETACH_TEARDOWN=1: time to results %.3f s (the worker ends ~0.35 s later)' % ($T1 - $T0))"
done
cmp $D/out-mems.txt $D/out-det.txt && echo "detached == default output"
cmp $D/out-mems.txt $D/out-seq.txt && echo "overlapped == sequential output"
tail -3 $D/stdout.txt
ls -l $D/out-mems.txt | awk '{print $5, "bytes"}'
sha256sum $D/out-mems.txt
