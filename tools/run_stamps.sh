#!/bin/bash
# tools/run_stamps.sh K... : per-phase cycle stamps of k_scan (tools/stamps.py) for the given top_k values; needs the
# variants st1/st2 (tools/build_variant.sh stN -DRBQ_STAMPS=N)
cd "$(dirname "$0")/.."
for k in "$@"; do
for m in 1 2; do
echo "== top_k $k mode $m"
RBQ_STAMPS_MODE=$m RBQ_LIB_PATH=$PWD/rabitq-rs_amd/csrc/variants/librbq_st$m.so python tools/stamps.py --top-k $k 2>&1 | grep -v "Warning\|amdgpu.ids"
done; done
