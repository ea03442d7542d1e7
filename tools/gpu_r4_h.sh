# round 4, call H: two-set patch prefetch (FEMASR_WINO_DEEP) in both Winograd kernels: parity, A/B against the one-set form, stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -x -p no:cacheprovider -k "winograd or wino or conv" > $O/h_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/h_kernels.log | cut -c1-300
for so in deep0 new; do
  if [ $so = new ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$so.so; fi
  for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
    echo -n "$so: "; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
  for shp in "16 144 144 256 128" "16 288 288 128 64"; do
    echo -n "$so: "; timeout 120 python tools/bench_conv.py $shp --up2 --gn-part --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
done > $O/h_ab.log 2>&1
cat $O/h_ab.log | cut -c1-200
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part --fast-act --iters 3 --wino 2>&1 | grep -v amdgpu.ids > $O/h_tt.log
timeout 120 python tools/bench_conv.py 16 288 288 128 64 --up2 --gn-part --iters 3 --wino 2>&1 | grep -v amdgpu.ids >> $O/h_tt.log
unset FEMASR_SO
cat $O/h_tt.log | cut -c1-230
timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg > $O/h_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/h_bench.log | cut -c1-250
