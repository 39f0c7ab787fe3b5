#!/bin/bash
# like variant_bench.sh, reporting K7q's own time from rocprofv3 kernel stats
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for flags in "$@"; do
  (cd slamem_amd/csrc && rm -f mem_search.o && make HIPFLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-result $flags" libslamem_hip.so > gpurun_out/variant_build.log 2>&1) || { echo "build failed for: $flags"; tail -5 gpurun_out/variant_build.log; continue; }
  rm -rf gpurun_out/prof_k7q
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k7q -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stats --no-host-leg > gpurun_out/prof_k7q.log 2>&1
  f=$(find gpurun_out/prof_k7q -name "*kernel_stats.csv" | head -1)
  echo "$flags $(grep k_pack_queries $f | cut -d, -f2-4) $(tail -1 gpurun_out/prof_k7q.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3))" 2>/dev/null)"
done
# leave the tree with the DEFAULT library: whatever runs next (tests, bench, evidence) must not measure a variant
(cd slamem_amd/csrc && rm -f mem_search.o && make libslamem_hip.so > gpurun_out/variant_build.log 2>&1) || { echo "restoring the default build FAILED"; exit 1; }
