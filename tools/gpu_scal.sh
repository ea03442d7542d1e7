cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/scal.log; : > $L
for cls in 3; do for cin in 64 128 256 512; do
FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 128 160 $cin 256 --gn --res --gn-part 2>&1 | grep conv >> $L
done; done
for cin in 64 128 256 512; do
timeout 120 python tools/bench_conv.py 16 128 160 $cin 256 --gn --res --gn-part 2>&1 | grep conv >> $L
done
cat $L
