cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_network.py -m gpu -q -x -p no:cacheprovider -k "attention or test_test_path or swin" 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --steps 3 --warmup 1 2>&1 | tail -1 | python tools/bench_summary.py | grep -i "MPix\|attention"
