#!/bin/bash
# tools/final_records.sh : the bench records committed under profiles/rN/ (GPU box; writes gpurun_out/final/*.json)
cd "$(dirname "$0")/.."
out=gpurun_out/final; rm -rf $out; mkdir -p $out
run() { name=$1; shift; python bench.py "$@" 2> $out/$name.err | grep '^{' | tail -n 1 > $out/$name.json; python - $out/$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d.get('pruned') or {}
print(sys.argv[1], round(d['value']), 'q/s', round(d['ms_per_step'], 4), 'ms/step recall', d.get('recall_at_10', d.get('recall')), 'roofline', round(d['roofline']['frac'], 3), 'skip', round(p.get('block_skip_frac', 0), 4))
PY
}
run bench_end_gist1m_b1024_steps20 --steps 20
run bench_end_gist1m_b1024_steps200 --steps 200 --no-cpu --no-extras
run bench_end_top100 --top-k 100 --no-cpu --no-extras
run bench_end_top100_steps200 --top-k 100 --steps 200 --no-cpu --no-extras
run bench_end_b4096 --batch 4096 --nbatches 8 --no-cpu --no-extras
run bench_end_sift --dim 128 --nlist 1024 --nprobe 64 --no-cpu --no-extras
run bench_end_ip3 --bits 3 --metric 1 --nprobe 256 --no-cpu --no-extras
