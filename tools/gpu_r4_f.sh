# round 4, call F: channel-pair input transform (variant bit 5) with / without staging first; the loop with the T phase removed (abl100) and
# with T phase and MFMAs removed (abl108): what the M phase alone costs
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for so in new v48 v50 abl100 abl108; do
  if [ $so = new ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$so.so; fi
  case $so in v*) echo -n "$so parity: "; FEMASR_TEST_SO=$FEMASR_SO timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "winograd or wino" 2>&1 | tail -1;; esac
  for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
    echo -n "$so: "; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
done > $O/f_ab.log 2>&1
unset FEMASR_SO
cat $O/f_ab.log | cut -c1-200
for st in 1 2 3; do echo -n "streams=$st: "; timeout 300 python bench.py --streams $st --no-cpu-baseline --no-bf16x3-leg --no-profile 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'])"; done
