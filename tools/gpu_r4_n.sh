# round 4, call N: where the joules go - package power and clock of the F(4x4,3x3) kernel with parts removed (tools/build_debug.sh abl<N>:
# 1 no patch loads, 2 no U loads, 4 no transform, 8 no MFMAs, 16 no output items, 32 no activation, 64 no staging stores; results garbage)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for tag in full abl1 abl2 abl4 abl8 abl16 abl32 abl96 abl108; do
  if [ $tag = full ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$tag.so; fi
  echo "== $tag"; timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part --fast-act --iters 2500 --wino --power 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-220
done > $O/n_power_abl.log 2>&1
unset FEMASR_SO
for shp in "16 144 144 256 256" "16 576 576 64 64"; do
  echo "== full $shp"; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 2500 --wino --power 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-220
done >> $O/n_power_abl.log 2>&1
echo "== x2 form 128->64 to 576^2"; timeout 120 python tools/bench_conv.py 16 288 288 128 64 --up2 --gn-part --iters 2500 --wino --power 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-220 >> $O/n_power_abl.log
echo "== direct form 128->128 at 144^2"; timeout 120 python tools/bench_conv.py 16 144 144 128 128 --gn --res --gn-part --fp32 --iters 4000 --power 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-220 >> $O/n_power_abl.log
for shp in "16 72 72 256 768" "16 72 72 256 256" "16 72 72 256 1024 --gelu" "16 72 72 1024 256"; do
  echo "== gemm $shp"; timeout 120 python tools/bench_conv.py $shp --k1 --fp32 --iters 8000 --power 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-220
done >> $O/n_power_abl.log 2>&1
cat $O/n_power_abl.log
