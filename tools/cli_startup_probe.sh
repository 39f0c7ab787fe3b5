#!/bin/bash
# Where the wall time of the command line goes outside main(): start-up (dynamic loading of the HIP runtime) and exit.
D=${1:-/tmp/c3}
mkdir -p $D
if [ ! -f $D/qry.fa ]; then python tools/gen_synth.py 100000000 10000000 150 0.02 42 50 $D > $D/gen.log; fi
wall() { local T0=$(date +%s.%N); "$@" > /dev/null 2> $D/err.txt; local T1=$(date +%s.%N); python3 -c "print('%.3f s' % ($T1 - $T0))"; }
echo "usage message only (exec + dynamic loading + exit):"; for i in 1 2 3; do wall slamem_amd/host/slaMEM-hip; done
echo "ld.so statistics:"; LD_DEBUG=statistics slamem_amd/host/slaMEM-hip 2>&1 >/dev/null | grep -E "total startup|relocation|load" | head -8
echo "a tiny input (golden case): the fixed cost of a process that used the GPU"
R=tests/golden/acgt_l20_both/ref.fa; Q=tests/golden/acgt_l20_both/q.fa
for i in 1 2 3; do
  T0=$(date +%s.%N)
  SLAMEM_TIMING=1 slamem_amd/host/slaMEM-hip -l 20 -o $D/tiny.txt $R $Q > /dev/null 2> $D/stderr.txt
  T1=$(date +%s.%N)
  python3 -c "print('process wall %.3f s' % ($T1 - $T0))"
  grep -h -o "total [0-9.]* s" $D/stderr.txt
done
echo "full runs, 3 s apart (default: forked worker, the command returns when the results are written):"
for i in 1 2 3; do
  sleep 3
  T0=$(date +%s.%N)
  SLAMEM_TIMING=1 slamem_amd/host/slaMEM-hip -b -l 20 -o $D/out-mems.txt $D/ref.fa $D/qry.fa > $D/stdout.txt 2> $D/stderr.txt
  T1=$(date +%s.%N)
  python3 -c "print('process wall %.3f s' % ($T1 - $T0))"
  grep -h timing $D/stderr.txt | sed -e "s/(index build.*pipeline set-up/... set-up/"
done
echo "one process (SLAMEM_FOREGROUND=1), _exit at the end:"
for i in 1 2; do
  sleep 3
  T0=$(date +%s.%N)
  SLAMEM_FOREGROUND=1 SLAMEM_TIMING=1 slamem_amd/host/slaMEM-hip -b -l 20 -o $D/out-mems.txt $D/ref.fa $D/qry.fa > $D/stdout.txt 2> $D/stderr.txt
  T1=$(date +%s.%N)
  python3 -c "print('process wall %.3f s' % ($T1 - $T0))"
  grep -h timing $D/stderr.txt | sed -e "s/(index build.*pipeline set-up/... set-up/"
done
echo "full teardown inside the process:"
for i in 1 2; do
  sleep 3
  T0=$(date +%s.%N)
  SLAMEM_FULL_TEARDOWN=1 SLAMEM_TIMING=1 slamem_amd/host/slaMEM-hip -b -l 20 -o $D/out-mems.txt $D/ref.fa $D/qry.fa > $D/stdout.txt 2> $D/stderr.txt
  T1=$(date +%s.%N)
  python3 -c "print('process wall %.3f s' % ($T1 - $T0))"
  grep -h timing $D/stderr.txt | sed -e "s/(index build.*pipeline set-up/... set-up/"
done
