#!/bin/bash
# round 5, second half: the diagnostic soak (tests/diag/soak.py) on the final build with its default kernel choice — small calls take the
# latency-first front, k_scan keeps the tie log (tied queries replay it through ParHeap) — and once more with k_scan forced everywhere
cd "$(dirname "$0")/.."
b=${1:-230000}
mkdir -p gpurun_out/soak5b
run() { tag=$1; mode=$2; first=$3; last=$4; timeout -k 10 420 python tests/diag/soak.py $first $last $mode > gpurun_out/soak5b/$tag.log 2>&1; echo "== $tag $first..$last: $(tail -n 2 gpurun_out/soak5b/$tag.log | tr '\n' ' ')"; }
run wide wide $b $((b+300))
run ties ties $((b+1000)) $((b+1300))
run lazy lazy $((b+6000)) $((b+6150))
run streams streams $((b+4000)) $((b+4030))
run threads threads $((b+5000)) $((b+5030))
export RBQ_SCAN_WAVE=0
run ties_kscan ties $((b+1300)) $((b+1500))
run wide_kscan wide $((b+300)) $((b+450))
