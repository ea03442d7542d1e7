# round 4, end-of-round measurements: full GPU suite, smoke(), bench of record, rocprofv3 passes (TAG), streams sweep, latency table,
# other BASELINE configs, tile2048, parity table, the testset/ test's summary line.  Logs under gpurun_out/.
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-200
FEMASR_VQ_FUSED=0 timeout 600 python -m pytest tests/test_gpu_vq_twopass.py -q -x -p no:cacheprovider > $O/pytest_vq_two_launches.log 2>&1; echo "vq tests, two-launch form rc=$?"; tail -1 $O/pytest_vq_two_launches.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-300
timeout 600 python bench.py > $O/bench_final.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_final.log | cut -c1-400
TAG=${TAG:-r04_run3} bash tools/gpu_prof.sh > $O/prof_summary.log 2>&1; cat $O/prof_rc.txt
cd $GRAFT_REPO_ROOT
for st in 1 2 4; do
  echo -n "streams=$st: "; timeout 300 python bench.py --streams $st --no-cpu-baseline --no-bf16x3-leg --no-profile 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'])"
done
timeout 600 python -m pytest tests/test_gpu_r4.py -q -x -s -p no:cacheprovider -k "testset_dir or size_limit" 2>&1 | grep -v amdgpu | grep "testset/\|limit 2\|passed\|failed" | cut -c1-400 > $O/r4_tests_summary.log; cat $O/r4_tests_summary.log
timeout 600 python tools/latency.py > $O/latency.log 2>&1; grep "graph=0" $O/latency.log | head -20
timeout 900 python tools/other_configs.py > $O/other_configs.log 2>&1; tail -6 $O/other_configs.log
timeout 600 python bench.py --workload tile2048 --steps 3 --warmup 1 --no-cpu-baseline --no-bf16x3-leg --no-profile > $O/bench_tile2048.log 2>&1; tail -1 $O/bench_tile2048.log | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print('tile2048', j['ms_per_step'], j['value'], j['timed_region'].get('last_step_split_ms_per_rank'))"
timeout 600 python tools/parity_report.py 2>&1 | grep -v amdgpu.ids > $O/parity_report.log; cat $O/parity_report.log | cut -c1-330
for f in 1 0; do
  echo "# FEMASR_VQ_FUSED=$f (1 = one launch, the default; 0 = candidate kernel + exact kernel)"
  FEMASR_VQ_FUSED=$f timeout 300 python tools/bench_vq.py 2>&1 | grep -v amdgpu.ids | grep "two-pass" | cut -c1-300
  FEMASR_VQ_FUSED=$f timeout 300 python tools/bench_vq.py --m 31104 2>&1 | grep -v amdgpu.ids | grep "two-pass" | cut -c1-300
done > $O/vq_fused.log 2>&1; cat $O/vq_fused.log
