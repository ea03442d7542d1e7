# round 4, call E: epilogue trims (border masks only at the border, scalar offsets formed in place), prologue requests first: A/B vs the
# first round-4 kernel (prev), the patch-1 preload variant (v16), stamps, then the whole GPU suite and the bench
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_r4.py -q -x -p no:cacheprovider -k "winograd or wino" > $O/e_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/e_kernels.log | cut -c1-300
for so in prev new v16; do
  if [ $so = new ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$so.so; fi
  if [ $so = v16 ]; then echo -n "v16 parity: "; FEMASR_TEST_SO=$FEMASR_SO timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "winograd or wino" 2>&1 | tail -1; fi
  for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
    echo -n "$so: "; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
  for shp in "16 144 144 256 128" "16 288 288 128 64"; do
    echo -n "$so: "; timeout 120 python tools/bench_conv.py $shp --up2 --gn-part --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
done > $O/e_ab.log 2>&1
cat $O/e_ab.log | cut -c1-200
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for shp in "16 288 288 128 128" "16 576 576 64 64"; do
  timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 3 --wino 2>&1 | grep -v amdgpu.ids
done > $O/e_tt.log 2>&1
timeout 120 python tools/bench_conv.py 16 288 288 128 64 --up2 --gn-part --iters 3 --wino 2>&1 | grep -v amdgpu.ids >> $O/e_tt.log
unset FEMASR_SO
cat $O/e_tt.log | cut -c1-230
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/e_pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -5 $O/e_pytest_gpu.log | cut -c1-300
timeout 600 python bench.py > $O/e_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/e_bench.log | cut -c1-300
