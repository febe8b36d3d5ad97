#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RBQ_SCAN_WAVE=1 timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests_wave1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5_tests_wave1.log
for v in default ${VARIANTS}; do
  for cfg in ${CFGS:-cfg3}; do
    BENCH_ARGS="--config $cfg" bash tools/ab_variants.sh $v 2>&1 | sed "s/^/$cfg /"
  done
done
STAGE_MASKS=15,8 timeout -k 10 300 python tools/stage_rates.py 2>/dev/null
