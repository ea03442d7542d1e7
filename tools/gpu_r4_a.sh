# round 4, call A: new Winograd epilogues / staging - parity of the two kernels, A/B against the round-3 kernels on the same box, phase stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "winograd or wino or gn" > $O/a_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -5 $O/a_kernels.log | cut -c1-300
for so in old new tt; do
  if [ $so = new ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$so.so; fi
  echo "== $so"
  for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
    timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -3
    timeout 120 python tools/bench_conv.py $shp --gn --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -3
  done
  for shp in "16 144 144 256 128" "16 288 288 128 64"; do
    timeout 120 python tools/bench_conv.py $shp --up2 --gn-part --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -3
  done
done > $O/a_ab.log 2>&1
unset FEMASR_SO
cat $O/a_ab.log | cut -c1-250
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/a_pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -5 $O/a_pytest_gpu.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg > $O/a_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/a_bench.log | cut -c1-600
