#!/bin/bash
# tools/records_cfg5.sh : cfg5 at full size (100 M x 768, nlist 65 536, nprobe 512, batch 16 384; streamed GPU encoder) with the
# self-check leg (GPU box, ~10 minutes; writes gpurun_out/final/bench_cfg5.json)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/final
python bench.py --config cfg5 --no-cpu --no-latency --kmeans-iters 3 2> gpurun_out/final/bench_cfg5.err | grep '^{' | tail -n 1 > gpurun_out/final/bench_cfg5.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/final/bench_cfg5.json')); p = d['pruned']
print('cfg5', round(d['value']), 'q/s', round(d['ms_per_step'], 3), 'ms/step recall', d.get('recall_at_10'), 'roofline', round(d['roofline']['frac'], 3),
      'stream entries/launch', p.get('stream_entries_per_launch'), 'req GB', p['bytes_requested_per_launch'] / 1e9, 'self_check', d.get('self_check'), 'encoder', d.get('encoder'))
PY
