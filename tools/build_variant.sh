#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : builds rabitq-rs_amd/csrc/variants/librbq_NAME.so for kernel A/B runs
# (select at run time with RBQ_LIB_PATH=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p rabitq-rs_amd/csrc/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -Wno-unused-function \
  -I include "$@" -o rabitq-rs_amd/csrc/variants/librbq_$name.so rabitq-rs_amd/csrc/device/rbq_api.hip
