#!/bin/bash
cd "$(dirname "$0")/.."
for v in head default; do
  for sw in 0 1; do
    BENCH_ARGS="--config top100 --option scan_wave=$sw" bash tools/ab_variants.sh $v 2>&1 | sed "s/^/sw=$sw /"
  done
done
