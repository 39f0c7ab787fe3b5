#!/bin/bash
# quick GPU iteration: the -m gpu suite (stop at first failure), then the bench line without the CPU legs
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-q}
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_$TAG.log
tail -15 gpurun_out/pytest_$TAG.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-leg ${@:2} > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc $?"
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print({k:d[k] for k in ("value","ms_per_step","kernel_ms","k8_ms","k8a_ms")})
    print({k:r[k] for k in ("frac","achieved","lines_64B","lines_per_s","random_line_ceiling_per_s","request_rate_frac","lane_use")})
    print(r["counters"])
except Exception as e:
    print("no bench line", e); print(open("gpurun_out/bench_$TAG.err").read()[-2000:])
PY
