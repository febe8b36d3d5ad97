#!/bin/bash
# round 5, second half: a second soak of the final build on fresh seed ranges (600 tie-heavy, 400 wide, 150 MSTG, 80 lists)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/soak5c
run() { tag=$1; mode=$2; first=$3; last=$4; timeout -k 10 500 python tests/diag/soak.py $first $last $mode > gpurun_out/soak5c/$tag.log 2>&1; echo "== $tag $first..$last: $(tail -n 2 gpurun_out/soak5c/$tag.log | tr '\n' ' ')"; }
run ties ties ${1:-241000} $((${1:-241000}+600))
run wide wide $((${1:-241000}-1000)) $((${1:-241000}-600))
run mstg mstg $((${1:-241000}+1000)) $((${1:-241000}+1150))
run lists lists $((${1:-241000}+2000)) $((${1:-241000}+2080))
