cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for tp in 0 90; do
  echo "== FEMASR_GEMM_TAIL_PCT=$tp"
  FEMASR_GEMM_TAIL_PCT=$tp timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --no-profile 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'], j['timed_region']['step_ms_first_median_last'], j['timed_region']['mfma_clock_ghz_before_after'])"
done; done
