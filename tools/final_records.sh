#!/bin/bash
# tools/final_records.sh : the bench records committed under profiles/rN/ (GPU box; writes gpurun_out/final/*.json)
cd "$(dirname "$0")/.."
out=gpurun_out/final; rm -rf $out; mkdir -p $out
run() { name=$1; shift; python bench.py "$@" 2> $out/$name.err | grep '^{' | tail -n 1 > $out/$name.json; python - $out/$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d.get('pruned') or {}
rec = [v for k, v in d.items() if k.startswith('recall_at')]
print(sys.argv[1], round(d['value']), 'q/s', round(d['ms_per_step'], 4), 'ms/step recall', rec, 'roofline', round(d['roofline']['frac'], 3) if d.get('roofline') else None,
      'skip', round(p.get('block_skip_frac', 0), 4), 'stage', d.get('stage_ms'))
PY
}
run bench_headline_steps20 --steps 20 --warmup 5
run bench_headline_steps200 --steps 200 --no-cpu --ab
run bench_cfg2 --config cfg2 --steps 100 --no-cpu --ab
run bench_cfg3_b4096 --config cfg3_b4096 --steps 50 --no-cpu --ab
run bench_cfg4 --config cfg4 --steps 100 --no-cpu --ab
run bench_top100 --config top100 --steps 100 --no-cpu --ab
run bench_top100_steps200 --config top100 --steps 200 --no-cpu --ab
run bench_d512 --dim 512 --steps 100 --no-cpu --ab
run bench_d1024 --dim 1024 --steps 100 --no-cpu --ab
