# round 3: small-launch block shapes (64x64 GEMM tiles, 64-column conv blocks): parity of every configuration, latency table, bench A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/latency.py --quick 2>&1 | grep -v amdgpu.ids | tee gpurun_out/latency_quick.log
for st in 700 100000000; do
  echo "== FEMASR_GEMM_SMALL_TILES=$st"
  FEMASR_GEMM_SMALL_TILES=$st timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --no-profile 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'], j['timed_region']['step_ms_first_median_last'], j['timed_region']['mfma_clock_ghz_before_after'])"
done
