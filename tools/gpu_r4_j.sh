# round 4, call J: the 16x16-pixel x 128-channel block shape of the F(4x4,3x3) conv (kernels_wino_c128.hip): MFMA layout probe, parity
# of both block shapes, A/B per layer shape (FEMASR_WINO_C128=0 = the 2 x 16x16 x 64 form everywhere), block-life stamps, a bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/ubench/mfma16_layout.hip -o /tmp/mfma16_layout && timeout 60 /tmp/mfma16_layout) > $O/j_layout.log 2>&1; head -3 $O/j_layout.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_r4.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "both_block_shapes or winograd" > $O/j_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -4 $O/j_kernels.log | cut -c1-400
for form in 0 1; do
  export FEMASR_WINO_C128=$form
  for shp in "16 144 144 256 256" "16 288 288 128 128"; do
    echo -n "c128=$form: "; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
  echo -n "c128=$form: "; timeout 120 python tools/bench_conv.py 16 72 72 512 256 --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
done > $O/j_ab.log 2>&1
unset FEMASR_WINO_C128
cat $O/j_ab.log | cut -c1-200
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part --fast-act --iters 3 --wino 2>&1 | grep -v amdgpu.ids > $O/j_tt.log
timeout 120 python tools/bench_conv.py 16 144 144 256 256 --gn --res --gn-part --fast-act --iters 3 --wino 2>&1 | grep -v amdgpu.ids >> $O/j_tt.log
unset FEMASR_SO
cat $O/j_tt.log | cut -c1-230
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_network_r3.py -q -x -p no:cacheprovider > $O/j_network.log 2>&1; echo "network tests rc=$?"; tail -3 $O/j_network.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg > $O/j_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/j_bench.log | cut -c1-250
FEMASR_WINO_C128=0 timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --no-profile > $O/j_bench_c0.log 2>&1; echo "bench (x64 form) rc=$?"; tail -1 $O/j_bench_c0.log | cut -c1-250
