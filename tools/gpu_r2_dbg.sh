cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; L=$O/r2_dbg.log; : > $L
for cfg in 0; do for dbg in 0 4 8 12 14 15; do
  export FEMASR_GEMM_CFG=$cfg FEMASR_GEMM_DBG=$dbg
  echo "#### CFG=$cfg DBG=$dbg (1 = no barriers, 2 = no stores, 4 = no A DMA, 8 = no W DMA)" >> $L
  for spec in "256 768" "1024 256 --res"; do
    timeout 120 python tools/bench_conv.py 1 82944 1 $spec --k1 --iters 20 2>&1 | grep "^conv" >> $L
  done
done; done
cat $L
