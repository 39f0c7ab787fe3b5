#!/bin/bash
# Build k_find_mems variants (on the GPU box) and time each with bench.py; prints kernel_ms per variant.
# usage: tools/variant_sweep.sh "<define list 1>" "<define list 2>" ...
cd slamem_amd/csrc
for V in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $V -c mem_search.hip -o mem_search.o 2> /dev/null || { echo "build failed: $V"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libslamem_hip.so capi.o index_build.o mem_search.o prims.o
  (cd ../.. && timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VARIANT', '$V', 'kernel_ms', round(d['kernel_ms'],2), 'mems', d['mems_per_step'])")
done
