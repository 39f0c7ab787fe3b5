#!/bin/bash
# AddressSanitizer + UBSan sweep of the C front end on the CPU (no GPU needed): for every golden case run the hidden -s and
# -c utilities, the -v image tool on the case's own MEMs file (single-record references) and a normal invocation (which parses both FASTA files with the loader threads and then stops at the GPU
# step when there is no device).  Prints "sanitizer findings: 0" when clean.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$ROOT/slamem_amd/host" slaMEM-hip-asan || exit 1
EXE="$ROOT/slamem_amd/host/slaMEM-hip-asan"
T=$(mktemp -d)
bad=0
for d in "$ROOT"/tests/golden/*/; do
  [ -f "$d/expected-mems.txt" ] || continue
  cp "$d/expected-mems.txt" "$T/x-mems.txt"; cp "$d/ref.fa" "$T/r.fa"
  (cd "$T" && "$EXE" -s x-mems.txt > s.out 2> s.err; "$EXE" -c r.fa > c.out 2> c.err
   ASAN_OPTIONS=detect_leaks=0 "$EXE" -v x-mems.txt "$d/ref.fa" "$d/q.fa" > v.out 2>> s.err
   SLAMEM_THREADS=4 ASAN_OPTIONS=detect_leaks=0 "$EXE" -b -l 10 -o o.txt "$d/ref.fa" "$d/q.fa" > m.out 2> m.err
   # the same with the query file parsed in pieces by the loader thread, in one process and through the forked worker
   SLAMEM_OVERLAP_MB=0 SLAMEM_FOREGROUND=1 SLAMEM_THREADS=4 ASAN_OPTIONS=detect_leaks=0 "$EXE" -b -l 10 -o o.txt "$d/ref.fa" "$d/q.fa" > m2.out 2>> m.err
   SLAMEM_OVERLAP_MB=0 SLAMEM_DETACH_TEARDOWN=1 SLAMEM_THREADS=4 ASAN_OPTIONS=detect_leaks=0 "$EXE" -b -l 10 -o o.txt "$d/ref.fa" "$d/q.fa" > m3.out 2>> m.err)
  if [ -s "$T/s.err" ] || [ -s "$T/c.err" ] || grep -q "AddressSanitizer\|runtime error" "$T/m.err"; then
    echo "== $d"; head -5 "$T/s.err" "$T/c.err"; grep -m3 "AddressSanitizer\|runtime error" "$T/m.err"; bad=$((bad + 1))
  fi
done
rm -rf "$T"
echo "sanitizer findings: $bad"
[ "$bad" = 0 ]
