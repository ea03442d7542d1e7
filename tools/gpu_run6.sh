cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" > gpurun_out/rc.txt
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/rc.txt
tail -4 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench.log; tail -2 gpurun_out/smoke.log; cat gpurun_out/rc.txt
