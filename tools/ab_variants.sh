#!/bin/bash
# tools/ab_variants.sh v1 v2 ... : run bench.py once per kernel variant (GPU box), print scan ms + QPS
cd "$(dirname "$0")/.."
for v in "$@"; do
  RBQ_LIB_PATH=$PWD/rabitq-rs_amd/csrc/variants/librbq_$v.so python bench.py --steps 30 --no-cpu ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$v', 'qps', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'stage', d['stage_ms'], 'frac', round(d['roofline']['frac'],3), 'recall', round(d['recall_at_10'],4))
"
done
