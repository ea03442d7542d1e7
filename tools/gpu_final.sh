# what the driver runs at round end: the whole GPU test suite, smoke(), default bench.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" > gpurun_out/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/rc.txt
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/rc.txt
tail -3 gpurun_out/pytest_gpu.log | cut -c1-200; tail -2 gpurun_out/smoke.log | cut -c1-200; tail -1 gpurun_out/bench.log | python tools/bench_summary.py; cat gpurun_out/rc.txt
