# round-2 experiment: the four Swin linear shapes at B=16 (M=82944) under GEMM pipeline configurations $CFGS, then the bench
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; L=$O/r2_lin.log; : > $L
for cfg in ${CFGS:-0 2 3}; do
  export FEMASR_GEMM_CFG=$cfg
  echo "#### FEMASR_GEMM_CFG=$cfg" >> $L
  timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider -k "linear or vq or general" 2>&1 | tail -2 >> $L
  for spec in "256 768" "256 1024 --gelu" "256 256 --res" "1024 256 --res"; do
    timeout 120 python tools/bench_conv.py 1 82944 1 $spec --k1 --iters 20 2>&1 | grep "^conv" >> $L
  done
  timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --steps 3 --warmup 1 2>&1 | tail -1 | python tools/bench_summary.py >> $L 2>&1
done
cat $L
