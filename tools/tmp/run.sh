RBQ_STAMPS_MODE=4 RBQ_LIB_PATH=$PWD/rabitq-rs_amd/csrc/variants/librbq_st4.so timeout -k 10 300 python tools/stamps.py --top-k 100 2>&1 | grep -v "Warning\|amdgpu.ids"
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu -k "top_k or ties or run or large or duplicate or random" > gpurun_out/t1.log 2>&1; tail -n 2 gpurun_out/t1.log
timeout -k 10 300 python tests/diag/soak.py 51000 51300 ties > gpurun_out/soak_ties.log 2>&1; tail -n 2 gpurun_out/soak_ties.log
BENCH_ARGS="--top-k 100" timeout -k 10 300 bash tools/ab_variants.sh default base default base 2>&1 | tee gpurun_out/ab_rr.log
