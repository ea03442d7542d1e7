cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/abl.log; : > $L
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for cls in 0 1; do
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part 2>&1 | grep conv >> $L
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --gn-part 2>&1 | grep conv >> $L
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 144 144 256 128 --up2 2>&1 | grep conv >> $L
done
cat $L
