# round 4, call D: schedule variants of the two Winograd kernels (tools/build_debug.sh v<N>): parity of each, timing, block-life stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for v in v0 v1 v4 v5 v8 v9; do
  export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$v.so
  echo -n "$v parity: "; FEMASR_TEST_SO=$FEMASR_SO timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "winograd or wino" 2>&1 | tail -1
  for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
    echo -n "$v: "; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
  for shp in "16 144 144 256 128" "16 288 288 128 64"; do
    echo -n "$v: "; timeout 120 python tools/bench_conv.py $shp --up2 --gn-part --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
done > $O/d_ab.log 2>&1
cat $O/d_ab.log | cut -c1-200
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for shp in "16 288 288 128 128" "16 576 576 64 64"; do
  timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 3 --wino 2>&1 | grep -v amdgpu.ids
done > $O/d_tt.log 2>&1
unset FEMASR_SO
cat $O/d_tt.log | cut -c1-230
