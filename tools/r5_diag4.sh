#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RBQ_SCAN_WAVE=1 timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests_wave1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5_tests_wave1.log
WSTAMPS_LOAD=11 WSTAMPS_MODE=2 RBQ_LIB_PATH=$PWD/rabitq-rs_amd/csrc/variants/librbq_wst2.so timeout -k 10 200 python tools/wstamps.py 2>/dev/null
CFGS="$CFGS" RATES=1 bash tools/r5_diag3.sh
