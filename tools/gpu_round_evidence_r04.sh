#!/bin/bash
# Evidence of round 4, in two GPU sessions (a session is at most 20 minutes):
#   part 1: the -m gpu suite, the default bench line, kernel-trace stats of the bench command
#   part 2: the PMC passes of the bench command, the repeat loads, the host-to-host legs, the command line end to end,
#           the bench on other batches (one strand, other read lengths, mixed lengths)
# usage: tools/gpu_round_evidence_r04.sh 1|2      (results under gpurun_out/; what is kept is copied into profiles/)
set -o pipefail
TAG=r04
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
if [ "$1" = 1 ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_$TAG.log
  tail -14 gpurun_out/pytest_$TAG.log
  timeout -k 10 500 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc $?"
  rm -rf gpurun_out/prof_$TAG
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stats --no-host-leg > gpurun_out/prof_$TAG.log 2>&1; echo "prof rc $?"
  find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
  head -8 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
  rm -rf gpurun_out/prof_$TAG
else
  bash tools/profile_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; echo "pmc rc $?"
  cp gpurun_out/prof/$TAG/summary.json gpurun_out/${TAG}_pmc_summary.json
  rm -rf gpurun_out/prof/$TAG
  python tools/repeat_load.py > gpurun_out/${TAG}_repeat_load.jsonl 2>/dev/null; echo "repeat load rc $?"
  python tools/host_leg_quick.py 2>/dev/null > gpurun_out/${TAG}_host_leg.jsonl; echo "host leg rc $?"
  bash tools/cli_config3.sh > gpurun_out/${TAG}_cli_config3.txt 2>&1; tail -4 gpurun_out/${TAG}_cli_config3.txt
  # the same bench on other batches of the headline text (one strand only, other read lengths), and a batch of mixed lengths
  : > gpurun_out/${TAG}_other_batches.jsonl
  for ARGS in "--forward-only" "--forward-only --reads 1000000" "--read-len 100" "--read-len 125" "--read-len 250 --reads 5000000" "--read-len 300 --reads 5000000"; do
    timeout -k 10 200 python bench.py $ARGS --steps 8 --warmup 2 --no-cpu-baseline --no-host-leg --long-stream-reads 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({"args": sys.argv[1], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "k8s_ms": d["k8s_ms"], "mems_per_step": d["mems_per_step"], "config": d["config"]}))' "$ARGS" >> gpurun_out/${TAG}_other_batches.jsonl || break
  done
  timeout -k 10 300 python tools/mixed_lengths_bench.py > gpurun_out/${TAG}_mixed_lengths.jsonl 2>/dev/null; echo "mixed lengths rc $?"
fi
