#!/bin/bash
# tools/soak_all.sh BASE : the diagnostic soak (tests/diag/soak.py) in all its modes on fresh seed ranges starting at BASE
cd "$(dirname "$0")/.."
b=${1:-70000}
mkdir -p gpurun_out/soak
run() { mode=$1; first=$2; last=$3; timeout -k 10 900 python tests/diag/soak.py $first $last $mode > gpurun_out/soak/$mode.log 2>&1; echo "== $mode $first..$last: $(tail -n 2 gpurun_out/soak/$mode.log | tr '\n' ' ')"; }
run wide $b $((b+500))
run ties $((b+1000)) $((b+1400))
run mstg $((b+2000)) $((b+2200))
run lists $((b+3000)) $((b+3100))
run streams $((b+4000)) $((b+4060))
run threads $((b+5000)) $((b+5060))
run lazy $((b+6000)) $((b+6300))
