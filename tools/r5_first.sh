#!/bin/bash
# round 5, first GPU run of k_scanw: the whole GPU suite with the wave kernel forced, then an A/B bench (scan_wave 0 / 1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RBQ_SCAN_WAVE=1 timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests_wave1.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r5_tests_wave1.log
tail -5 gpurun_out/r5_tests_wave1.log
for sw in 0 1; do
  timeout -k 10 200 python bench.py --steps 40 --no-cpu --ab --option scan_wave=$sw > gpurun_out/r5_ab_sw$sw.json 2> gpurun_out/r5_ab_sw$sw.err
  echo "bench sw=$sw rc=$?"
  python - <<PY
import json
for line in open('gpurun_out/r5_ab_sw$sw.json'):
    if line.startswith('{'):
        d = json.loads(line); p = d['pruned']
        print('sw=$sw qps', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'stage', d['stage_ms'], 'roofline', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4), 'pruned ms', round(p['avg_launch_ms'],4), 'recall', round(d.get('recall_at_10', 0),4), 'restarts', d.get('heap_restarts'))
PY
done
