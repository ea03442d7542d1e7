cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16" > gpurun_out/pytest_b16.log 2>&1; echo "pytest b16 rc=$?" > gpurun_out/rc.txt
timeout 1200 python -m pytest tests/test_gpu_network.py -m gpu -q --tb=short -p no:cacheprovider -s -k "bf16x3" > gpurun_out/pytest_b16n.log 2>&1; echo "pytest b16 net rc=$?" >> gpurun_out/rc.txt
for m in fp32 bf16x3; do timeout 600 python bench.py --steps 3 --warmup 1 --decoder-math $m --no-cpu-baseline > gpurun_out/bench_$m.log 2>&1; echo "bench $m rc=$?" >> gpurun_out/rc.txt; done
tail -25 gpurun_out/pytest_b16.log | cut -c1-220; tail -12 gpurun_out/pytest_b16n.log | cut -c1-220; for m in fp32 bf16x3; do tail -1 gpurun_out/bench_$m.log; done; cat gpurun_out/rc.txt
