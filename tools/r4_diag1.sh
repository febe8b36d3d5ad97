#!/bin/bash
# round-4 diagnostics: the audit test, live lists of cfg2, lazy taps of cfg2 / headline / a mid-size cfg5 analogue
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_round4.py -m gpu -q --timeout 900 -k "audit" 2>&1 | tail -3
python tools/live_lists.py --config cfg2 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_live_cfg2.log; cat gpurun_out/r4_live_cfg2.log
python tools/lazy_taps.py --config cfg2 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_taps_cfg2.log; cat gpurun_out/r4_taps_cfg2.log
python tools/lazy_taps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_taps_head.log; cat gpurun_out/r4_taps_head.log
python tools/lazy_taps.py --n 8000000 --dim 768 --nlist 8192 --nprobe 512 --batch 2048 --kmeans-iters 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_taps_cfg5m.log; cat gpurun_out/r4_taps_cfg5m.log
