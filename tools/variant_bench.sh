#!/bin/bash
mkdir -p gpurun_out
# rebuild mem_search.o with extra -D flags on the GPU box and time the bench step: tools/variant_bench.sh "<flags>" ...
for flags in "$@"; do
  (cd slamem_amd/csrc && rm -f mem_search.o && make HIPFLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-result $flags" libslamem_hip.so > gpurun_out/variant_build.log 2>&1) || { echo "build failed for: $flags"; tail -5 gpurun_out/variant_build.log; continue; }
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-leg --no-stats 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$flags', {k:round(d[k],3) for k in ('ms_per_step','kernel_ms','k8_ms','k8a_ms')})"
done
# leave the tree with the DEFAULT library: whatever runs next (tests, bench, evidence) must not measure a variant
(cd slamem_amd/csrc && rm -f mem_search.o && make libslamem_hip.so > gpurun_out/variant_build.log 2>&1) || { echo "restoring the default build FAILED"; exit 1; }
