#!/bin/bash
# tools/ab_variants.sh v1 v2 ... : run bench.py once per kernel variant (GPU box; "default" = the in-tree library),
# print rate, per-stage times alone, roofline and pruned figures
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = default ]; then unset RBQ_LIB_PATH; else export RBQ_LIB_PATH=$PWD/rabitq-rs_amd/csrc/variants/librbq_$v.so; fi
  python bench.py --steps 40 --no-cpu --ab ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); p = d['pruned']
        print('$v', 'qps', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'stage', d['stage_ms'], 'roofline', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4),
              'pruned ms', round(p['avg_launch_ms'],4), 'req MB', round(p['bytes_requested_per_launch']/1e6,1), 'ex MB', round(p['bytes_requested_by_array']['ex_codes']/1e6,1), 'recall', round(d.get('recall_at_10', 0),4))
"
done
