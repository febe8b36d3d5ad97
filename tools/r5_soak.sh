#!/bin/bash
# round-5 soak: the diagnostic soak (tests/diag/soak.py) with k_scanw forced wherever it serves the call (RBQ_SCAN_WAVE=1)
cd "$(dirname "$0")/.."
b=${1:-210000}
mkdir -p gpurun_out/soak5
run() { mode=$1; first=$2; last=$3; RBQ_SCAN_WAVE=1 timeout -k 10 700 python tests/diag/soak.py $first $last $mode > gpurun_out/soak5/$mode.log 2>&1; echo "== $mode $first..$last: $(tail -n 2 gpurun_out/soak5/$mode.log | tr '\n' ' ')"; }
run wide $b $((b+400))
run ties $((b+1000)) $((b+1300))
run lists $((b+3000)) $((b+3080))
run streams $((b+4000)) $((b+4040))
run threads $((b+5000)) $((b+5040))
run lazy $((b+6000)) $((b+6250))
