cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/abl.log; : > $L
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k bf16 2>&1 | tail -3 >> $L
timeout 120 python tools/bench_conv.py 16 576 576 64 64 --gn --res --gn-part 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 576 576 64 64 --gn --gn-part 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 576 576 64 3 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 288 288 128 64 --up2 2>&1 | grep conv >> $L
cat $L
