cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/h4.log; : > $L
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4 >> $L
timeout 120 python tools/bench_conv.py 16 144 144 256 256 --gn --res --fp32 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 144 144 256 256 --fp32 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 144 144 256 256 --up2 --fp32 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --fp32 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 576 576 64 64 --gn --res --fp32 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 576 576 64 3 --fp32 2>&1 | grep conv >> $L
cat $L
