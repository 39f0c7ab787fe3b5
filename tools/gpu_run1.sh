#!/bin/bash
# GPU session: full -m gpu test suite, default bench line, kernel-trace stats of the bench.
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_r02.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_r02.log
tail -25 gpurun_out/pytest_r02.log
timeout -k 10 400 python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err; echo "bench rc $?"
tail -c 3000 gpurun_out/bench_r02.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stats --no-host-leg > gpurun_out/prof_r02.log 2>&1; echo "prof rc $?"
find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
