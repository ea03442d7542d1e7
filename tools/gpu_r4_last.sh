# round 4, last call: the whole GPU suite, smoke() and the default bench line on the final commit (no profiler passes)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/last_pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -2 $O/last_pytest_gpu.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/last_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/last_smoke.log | cut -c1-300
timeout 600 python bench.py > $O/last_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/last_bench.log | cut -c1-300
