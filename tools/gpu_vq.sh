cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vq_twopass.py -q -x -s -p no:cacheprovider > $O/pytest_vq.log 2>&1; echo "pytest vq rc=$?"
grep -E "mean candidates|passed|failed|Error|error" $O/pytest_vq.log | cut -c1-250 | tail -30
timeout 300 python tools/bench_vq.py > $O/bench_vq.log 2>&1; echo "bench_vq rc=$?"; tail -12 $O/bench_vq.log | cut -c1-300
FEMASR_VQ_SLOTS=4 timeout 300 python tools/bench_vq.py --regime gauss > $O/bench_vq4.log 2>&1; tail -3 $O/bench_vq4.log | cut -c1-300
