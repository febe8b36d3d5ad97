#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : builds rabitq-rs_amd/csrc/variants/librbq_NAME.so (all translation units with
# the extra flags) for kernel A/B runs; select at run time with RBQ_LIB_PATH=...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=rabitq-rs_amd/csrc/variants; mkdir -p $out/obj_$name
for u in k_scan k_scan2 k_scanw k_scanw2 k_query k_build rbq_api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Wno-unused-function \
    -I include "$@" -c rabitq-rs_amd/csrc/device/$u.hip -o $out/obj_$name/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $out/librbq_$name.so $out/obj_$name/*.o
echo built $out/librbq_$name.so
