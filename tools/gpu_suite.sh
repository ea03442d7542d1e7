cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?" > gpurun_out/rc.txt
timeout 1200 python -m pytest tests/test_gpu_network.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_network.log 2>&1; echo "pytest net rc=$?" >> gpurun_out/rc.txt
for m in bf16x3; do timeout 600 python bench.py --steps 3 --warmup 1 --decoder-math $m --no-cpu-baseline > gpurun_out/bench_$m.log 2>&1; echo "bench $m rc=$?" >> gpurun_out/rc.txt; done
tail -3 gpurun_out/pytest_kernels.log | cut -c1-220; tail -3 gpurun_out/pytest_network.log | cut -c1-220; tail -1 gpurun_out/bench_bf16x3.log; cat gpurun_out/rc.txt
