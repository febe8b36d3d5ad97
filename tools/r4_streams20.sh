#!/bin/bash
# streams sweep at the driver's protocol (--steps 20 --warmup 5): how many streams suit a 20-step region?
cd "$(dirname "$0")/.."
for ns in 4 5 7 10 12 14 20; do
  python bench.py --steps 20 --warmup 5 --no-cpu --no-extras --streams $ns 2>/dev/null | python -c "
import sys, json, statistics
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['region_ms']
        print('streams $ns: %.0f q/s  ms/step %.4f  regions %d  region ms p10 %.3f median %.3f p90 %.3f' % (d['value'], d['ms_per_step'], len(r), sorted(r)[len(r)//10], statistics.median(r), sorted(r)[9*len(r)//10]))
"
done
