# round 4, call B: where a block of the Winograd kernels spends its life (cycle stamps to a global buffer) + ablation builds
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
  timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 5 --wino 2>&1 | grep -v amdgpu.ids
done > $O/b_tt.log 2>&1
for shp in "16 144 144 256 128" "16 288 288 128 64"; do
  timeout 120 python tools/bench_conv.py $shp --up2 --gn-part --iters 5 --wino 2>&1 | grep -v amdgpu.ids
done >> $O/b_tt.log 2>&1
cat $O/b_tt.log | cut -c1-220
for tag in new abl1 abl2 abl4 abl8 abl16 abl32 abl96; do
  if [ $tag = new ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$tag.so; fi
  for shp in "16 288 288 128 128" "16 576 576 64 64"; do
    echo -n "$tag: "; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
done > $O/b_abl.log 2>&1
cat $O/b_abl.log | cut -c1-200
