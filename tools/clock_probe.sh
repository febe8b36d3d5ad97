#!/bin/bash
# tools/clock_probe.sh : which shader clock the GPU runs at while (a) single-query calls run back to back, (b) the pipelined bench runs
# (rocm-smi sampled from the side; read-only)
cd "$(dirname "$0")/.."
sample() { for i in 1 2 3 4 5 6; do sleep 1.5; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | tr '\n' ' '; echo; done; }
echo "== idle"; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | tr '\n' ' '; echo
echo "== single-query calls back to back (tools/latency_probe.py)"
LAT_PROBE_NQ=1 LAT_PROBE_CALLS=120000 python tools/latency_probe.py > /tmp/lp.txt 2>&1 &
pid=$!
sleep 8; sample
wait $pid; grep nq /tmp/lp.txt
echo "== pipelined bench (bench.py --steps 4000)"
python bench.py --steps 4000 --no-cpu --no-extras --min-seconds 0 > /tmp/b.txt 2>&1 &
pid=$!
sleep 12; sample
wait $pid
