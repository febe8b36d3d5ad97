for k in 10 100; do
for m in 1 2 3; do
echo "== top_k $k mode $m"
RBQ_STAMPS_MODE=$m RBQ_LIB_PATH=$PWD/rabitq-rs_amd/csrc/variants/librbq_st$m.so python tools/stamps.py --top-k $k 2>&1 | grep -v Warning
done; done
