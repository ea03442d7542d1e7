# round 4, call C: earlier patch loads + reordered T phase: parity, A/B (round-3 kernel, first round-4 kernel, this one), block-life stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "winograd or wino" > $O/c_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/c_kernels.log | cut -c1-300
for so in old prev new; do
  if [ $so = new ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$so.so; fi
  for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
    echo -n "$so: "; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
done > $O/c_ab.log 2>&1
cat $O/c_ab.log | cut -c1-200
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
  timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 3 --wino 2>&1 | grep -v amdgpu.ids
done > $O/c_tt.log 2>&1
for shp in "16 144 144 256 128" "16 288 288 128 64"; do
  timeout 120 python tools/bench_conv.py $shp --up2 --gn-part --iters 3 --wino 2>&1 | grep -v amdgpu.ids
done >> $O/c_tt.log 2>&1
unset FEMASR_SO
cat $O/c_tt.log | cut -c1-230
