#!/bin/bash
# BASELINE.json configs[2] end to end through the command line: FASTA files in, *-mems.txt out; three timed runs, 2 s apart
# (the command returns when its results are written; its worker then gives 5 GB of HBM and the pinned buffers back for
#  another ~0.35 s, and a run started within that time waits for it: SPACING=0 shows that)
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
cmp $D/out-mems.txt $D/out-seq.txt && echo "overlapped == sequential output"
tail -3 $D/stdout.txt
ls -l $D/out-mems.txt | awk '{print $5, "bytes"}'
sha256sum $D/out-mems.txt
