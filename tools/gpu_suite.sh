# what a gpurun call executes during development: all GPU tests (stop at first failure) + default bench.py
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
tail -25 $O/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py ${BENCH_ARGS:-} > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-5000
