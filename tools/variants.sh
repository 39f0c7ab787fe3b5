#!/bin/bash
# A/B variants of the search kernels WITHOUT touching the product's library: each variant is mem_search.hip compiled
# with extra -D flags and linked with the product's other objects into slamem_amd/csrc/variants/libslamem_hip_<name>.so.
#   build (CPU container, hipcc cross-compiles):   tools/variants.sh build name1:"-DX=1" name2:"-DY=2 -DZ" ...
#   run   (GPU box):                               tools/variants.sh run "<program and args>" name1 name2 ...
# `run` executes the program once per variant with SLAMEM_HIP_LIB pointing at that variant (`default` = the product's library).
set -e
cd "$(dirname "$0")/.."
V=slamem_amd/csrc/variants
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p $V
  (cd slamem_amd/csrc && make -j8 libslamem_hip.so > /dev/null)
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    ( cd slamem_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-result $flags -c mem_search.hip -o variants/mem_search_$name.o \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libslamem_hip_$name.so capi.o index_build.o variants/mem_search_$name.o radix_sort.o scan.o stream.o -lpthread \
      && rm -f variants/mem_search_$name.o && echo "built $name: $flags" ) &
  done
  wait
elif [ "$mode" = run ]; then
  prog=$1; shift
  for name in "$@"; do
    if [ "$name" = default ]; then lib=$PWD/slamem_amd/csrc/libslamem_hip.so; else lib=$PWD/$V/libslamem_hip_$name.so; fi
    [ -f "$lib" ] || { echo "$name: no library"; continue; }
    echo "== $name"
    SLAMEM_HIP_LIB=$lib $prog 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-3}
  done
fi
