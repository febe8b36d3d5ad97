#!/bin/bash
# round-4: exact head evaluation — parity of every lazy-selection test, then its effect on the lists scanned and the rates
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -x --timeout 900 -k "lazy or drops or audit or duplicate or degenerate or non_finite or scales" 2>&1 | tail -5
python tools/lazy_taps.py --config cfg2 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_taps2_cfg2.log; cat gpurun_out/r4_taps2_cfg2.log
python tools/lazy_taps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_taps2_head.log; cat gpurun_out/r4_taps2_head.log
python tools/lazy_taps.py --n 8000000 --dim 768 --nlist 8192 --nprobe 512 --batch 2048 --kmeans-iters 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_taps2_cfg5m.log; cat gpurun_out/r4_taps2_cfg5m.log
for opt in 1 0; do
  python bench.py --config cfg2 --steps 100 --no-cpu --no-extras --option head_exact=$opt > gpurun_out/r4_cfg2_hx$opt.json 2>/dev/null
  python bench.py --steps 100 --no-cpu --no-extras --option head_exact=$opt > gpurun_out/r4_head_hx$opt.json 2>/dev/null
  python bench.py --config top100 --steps 100 --no-cpu --no-extras --option head_exact=$opt > gpurun_out/r4_top100_hx$opt.json 2>/dev/null
  python bench.py --config cfg4 --steps 100 --no-cpu --no-extras --option head_exact=$opt > gpurun_out/r4_cfg4_hx$opt.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_*_hx?.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "q/s %.0f ms/step %.4f recall %.4f roofline %.3f skip %.4f stream %.0f" % (d["value"], d["ms_per_step"], d.get("recall_at_10", d.get("recall_at_100", 0)), d["roofline"]["frac"], d["pruned"]["block_skip_frac"], d["pruned"]["stream_entries_per_launch"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
