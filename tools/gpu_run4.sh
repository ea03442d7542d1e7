cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" > gpurun_out/rc.txt
for s in 2; do timeout 600 python bench.py --steps 3 --warmup 1 --streams $s > gpurun_out/bench_s$s.log 2>&1; echo "bench s=$s rc=$?" >> gpurun_out/rc.txt; done
tail -4 gpurun_out/pytest_gpu.log; for s in 2; do tail -1 gpurun_out/bench_s$s.log; done; cat gpurun_out/rc.txt
