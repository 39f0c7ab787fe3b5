#!/bin/bash
# a larger run through the command line: 100 Mbp reference x ${1:-30000000} reads of 150 bp (4.9 GB of FASTA at 30 M), -b -l 20
N=${1:-30000000}
D=/tmp/cbig; mkdir -p $D
T0=$(date +%s.%N); python tools/gen_synth.py 100000000 $N 150 0.02 42 50 $D > $D/gen.log; T1=$(date +%s.%N)
python3 -c "print('generator %.1f s' % ($T1 - $T0))"; ls -l $D/qry.fa | awk '{print $5, "bytes of reads"}'
for i in 1 2; do
  sleep 2
  T0=$(date +%s.%N)
  SLAMEM_TIMING=1 slamem_amd/host/slaMEM-hip -b -l 20 -o $D/out.txt $D/ref.fa $D/qry.fa > $D/stdout.txt 2> $D/stderr.txt || { tail -3 $D/stdout.txt; cat $D/stderr.txt; exit 1; }
  T1=$(date +%s.%N)
  python3 -c "print('process wall %.3f s' % ($T1 - $T0))"
  grep -h timing $D/stderr.txt | sed -e "s/(index build.*pipeline set-up/... set-up/"
done
tail -3 $D/stdout.txt; ls -l $D/out.txt | awk '{print $5, "bytes of output"}'
