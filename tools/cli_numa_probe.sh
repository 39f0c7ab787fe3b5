D=/tmp/c3; mkdir -p $D
python tools/gen_synth.py 100000000 10000000 150 0.02 42 50 $D > $D/gen.log
which taskset numactl lscpu 2>/dev/null; lscpu | grep -i "numa\|socket\|model name" | head -8
for f in /sys/class/drm/card*/device/local_cpulist; do echo "$f: $(cat $f)"; done 2>/dev/null | head -4
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | head -2
run() { sleep 2; T0=$(date +%s.%N); SLAMEM_TIMING=1 "$@" slamem_amd/host/slaMEM-hip -b -l 20 -o $D/o.txt $D/ref.fa $D/qry.fa > /dev/null 2> $D/err.txt; T1=$(date +%s.%N); echo "$* wall $(python3 -c "print('%.3f' % ($T1-$T0))") $(grep -o 'beside the search in [0-9.]* s\|format [0-9.]* s\|busy [0-9.]* s in all' $D/err.txt | tr '\n' ';')"; }
run
run taskset -c 0-31
run taskset -c 0-63
run taskset -c 64-127
run taskset -c 128-191
run taskset -c 192-255
run
