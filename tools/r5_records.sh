#!/bin/bash
# tools/r5_records.sh : the round-5 bench records committed under profiles/r5/ (GPU box; writes gpurun_out/final_r5/*.json).
# Per-configuration records carry cpu_baseline and the oracle id check (round-3 VERDICT item 6).
cd "$(dirname "$0")/.."
out=gpurun_out/final_r5; rm -rf $out; mkdir -p $out
run() { name=$1; shift; python bench.py "$@" 2> $out/$name.err | grep '^{' | tail -n 1 > $out/$name.json; python - $out/$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d.get('pruned') or {}
rec = [v for k, v in d.items() if k.startswith('recall_at')]
hc = d.get('host_call') or {}
cb = d.get('cpu_baseline') or {}
print(sys.argv[1], round(d['value']), 'q/s', round(d['ms_per_step'], 4), 'ms/step recall', rec, 'roofline', round(d['roofline']['frac'], 3) if d.get('roofline') else None,
      'skip', round(p.get('block_skip_frac', 0), 4), 'stage', d.get('stage_ms'), 'host_call', round(hc.get('queries_per_s', 0)), round(hc.get('pageable_queries_per_s', 0)),
      'cpu', round(cb.get('value', 0)), cb.get('ids_identical_to_gpu'), 'lat', (d.get('latency') or {}).get('1', {}).get('p50_us'))
PY
}
run bench_headline_steps20 --steps 20 --warmup 5
run bench_headline_steps200 --steps 200
run bench_cfg2 --config cfg2 --steps 100
run bench_cfg4 --config cfg4 --steps 100
run bench_top100 --config top100 --steps 100
run bench_cfg3_b4096 --config cfg3_b4096 --steps 50 --no-cpu
run bench_isotropic --dataset isotropic --steps 100 --no-cpu
