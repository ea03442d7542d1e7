cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(lscpu | egrep "Model name|^CPU\(s\)|Thread|Socket"; rocminfo | egrep "Marketing|gfx" | head -4; free -g | head -2) > gpurun_out/box.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_kernels.log 2>&1; echo "kernels rc=$?" >> gpurun_out/box.txt
timeout 1500 python -m pytest tests/test_gpu_network.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_network.log 2>&1; echo "network rc=$?" >> gpurun_out/box.txt
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/box.txt
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/box.txt
cd $GRAFT_REPO_ROOT; tail -5 gpurun_out/pytest_kernels.log; tail -5 gpurun_out/pytest_network.log; tail -3 gpurun_out/bench.log; cat gpurun_out/box.txt
