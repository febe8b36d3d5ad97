#!/bin/bash
# A/B of library variants on one box: tools/r5_ab2.sh "cfgA cfgB" v1 v2 ... (default = the in-tree library); twice each, interleaved
cd "$(dirname "$0")/.."
cfgs=$1; shift
for rep in 1 2; do
for v in "$@"; do
  for cfg in $cfgs; do
    BENCH_ARGS="--config $cfg ${EXTRA_ARGS}" bash tools/ab_variants.sh $v 2>&1 | sed "s/^/$cfg /"
  done
done
done
