#!/bin/bash
# round 5 diagnosis: what is the overlapped step made of?  stage rates alone (scan_wave 0 / 1), kernel trace with gaps
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for sw in 1 0; do
  echo "== stage rates, scan_wave=$sw"
  STAGE_STREAMS=${STAGE_STREAMS:-12} timeout -k 10 300 python tools/stage_rates.py --option scan_wave=$sw 2>gpurun_out/r5_stage_rates_sw$sw.err | tee gpurun_out/r5_stage_rates_sw$sw.log
done
rm -rf gpurun_out/r5_trace_sw1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_trace_sw1 -- python3 bench.py --steps 100 --no-cpu --no-extras --min-seconds 0 --option scan_wave=1 > gpurun_out/r5_trace_sw1.json 2> gpurun_out/r5_trace_sw1.err
python tools/trace_overlap.py $(ls gpurun_out/r5_trace_sw1/*/*_kernel_trace.csv | head -1) | tee gpurun_out/r5_trace_sw1.log
find gpurun_out/r5_trace_sw1 -name "*.csv" -size +20M -delete
