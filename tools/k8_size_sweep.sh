#!/bin/bash
# K8 / K8a time against batch size and persistent-wave count (device-resident inputs)
for w in 4096; do for r in 500000 1000000 2000000 5000000; do
  SLAMEM_K8_WAVES=$w python bench.py --reads $r --steps 8 --warmup 2 --no-cpu-baseline --no-host-leg --no-stats 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($w, $r, {k:round(d[k],3) for k in ('ms_per_step','kernel_ms','k8_ms','k8a_ms')}, 'K8 ns/read', round(d['k8_ms']*1e6/$r,2))"
done; done
