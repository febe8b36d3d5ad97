#!/bin/bash
# round 5: k_scanw stamps (alone and under load), A/B on the other configurations
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/rabitq-rs_amd/csrc/variants/librbq_wst.so
echo "== stamps, alone"; RBQ_LIB_PATH=$V timeout -k 10 200 python tools/wstamps.py 2>gpurun_out/r5_wst.err | tee gpurun_out/r5_wst_alone.log
echo "== stamps, 11 streams of load"; WSTAMPS_LOAD=11 RBQ_LIB_PATH=$V timeout -k 10 200 python tools/wstamps.py 2>>gpurun_out/r5_wst.err | tee gpurun_out/r5_wst_load.log
for cfg in cfg2 top100 cfg4; do
  for sw in 0 1; do
    timeout -k 10 250 python bench.py --config $cfg --steps 40 --no-cpu --ab --option scan_wave=$sw > gpurun_out/r5_ab_${cfg}_sw$sw.json 2> gpurun_out/r5_ab_${cfg}_sw$sw.err
    python - <<PY
import json
for line in open('gpurun_out/r5_ab_${cfg}_sw$sw.json'):
    if line.startswith('{'):
        d = json.loads(line); p = d['pruned']
        print('$cfg sw=$sw qps', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'stage', d['stage_ms'], 'roofline', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4), 'recall', round(d.get('recall_at_10', d.get('recall_at_k', 0)),4), 'restarts', d.get('heap_restarts'))
PY
  done
done
