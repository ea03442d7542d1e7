cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/k1.log; : > $L
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -p no:cacheprovider -x -k "linear or fuzz_linear" 2>&1 | tail -3 >> $L
for B in 16 8; do
timeout 120 python tools/bench_conv.py $B 72 72 256 1024 --k1 --ln --gelu 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py $B 72 72 256 768 --k1 --ln 2>&1 | grep conv >> $L
done
cat $L
