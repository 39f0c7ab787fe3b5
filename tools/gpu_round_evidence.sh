#!/bin/bash
# Evidence of a round, one GPU session: full -m gpu suite, the default bench line, kernel-trace stats of the same
# command, the PMC passes, K8 against batch size, the host-to-host sweep, the CLI end to end, -mam on the genome pair.
# usage: tools/gpu_round_evidence.sh <tag>      (results under gpurun_out/; copy what is kept into profiles/)
set -o pipefail
TAG=${1:-r03}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_$TAG.log
tail -12 gpurun_out/pytest_$TAG.log
timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stats --no-host-leg > gpurun_out/prof_$TAG.log 2>&1; echo "prof rc $?"
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
head -6 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-140
bash tools/profile_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; echo "pmc rc $?"
cp gpurun_out/prof/$TAG/summary.json gpurun_out/${TAG}_pmc_summary.json
bash tools/k8_size_sweep.sh > gpurun_out/${TAG}_k8_size_sweep.txt 2>&1
python tools/host_leg_r03.py default 2>/dev/null > gpurun_out/${TAG}_host_leg.jsonl
SLAMEM_STREAM_CARRY=1 python tools/host_leg_r03.py carry 2>/dev/null >> gpurun_out/${TAG}_host_leg.jsonl
SLAMEM_STREAM_UPLOAD_SPLIT=1 python tools/host_leg_r03.py one_copy_stream 2>/dev/null >> gpurun_out/${TAG}_host_leg.jsonl
bash tools/cli_config3.sh > gpurun_out/${TAG}_cli_config3.txt 2>&1; tail -4 gpurun_out/${TAG}_cli_config3.txt
python tests/tools/mam_genome_pair.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_mam_genome_pair.json; cat gpurun_out/${TAG}_mam_genome_pair.json
