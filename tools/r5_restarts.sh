#!/bin/bash
cd "$(dirname "$0")/.."
for sw in 0 1; do
  python bench.py --config ${CFG:-top100} --steps 40 --no-cpu --ab --option scan_wave=$sw 2>/dev/null > gpurun_out/r5_restarts_$sw.json
  python - <<PY
import json
for line in open('gpurun_out/r5_restarts_$sw.json'):
    if line.startswith('{'):
        d = json.loads(line); print('sw', $sw, 'qps', round(d['value']), 'restarts', d['heap_restarts'], 'stage', d['stage_ms'], 'pruned ms', round(d['pruned']['avg_launch_ms'], 4))
PY
done
