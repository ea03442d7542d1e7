# round 4, call P: window attention with two (window, head) units per wave (the second unit's Q / K rows in flight under the first one's softmax)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -x -p no:cacheprovider -k "attention" 2>&1 | tail -1 | cut -c1-200
for u in 2 1 2 1; do
  echo -n "FEMASR_ATT_UNITS=$u: "; FEMASR_ATT_UNITS=$u timeout 120 python tools/bench_attn.py 16 72 72 4 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "FEMASR_ATT_UNITS=$u: "; FEMASR_ATT_UNITS=$u timeout 120 python tools/bench_attn.py 16 72 72 0 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "FEMASR_ATT_UNITS=$u: "; FEMASR_ATT_UNITS=$u timeout 120 python tools/bench_attn.py 5 72 72 4 2>&1 | grep -v amdgpu.ids | tail -1
done > $O/p_attn.log 2>&1; cat $O/p_attn.log
timeout 600 python -m pytest tests/test_gpu_network.py -q -x -p no:cacheprovider -k "golden or reference or default" 2>&1 | tail -1 | cut -c1-200
for rep in 1 2; do for u in 2 1; do
  echo -n "bench FEMASR_ATT_UNITS=$u: "; FEMASR_ATT_UNITS=$u timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --no-profile 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'], j['timed_region']['step_ms_first_median_last'])"
done; done
