cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/tt.log; : > $L
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for cls in 0 3; do
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 144 144 256 256 --gn --res --gn-part >> $L 2>&1
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 144 144 256 256 --gn --gn-part >> $L 2>&1
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 144 144 256 256 >> $L 2>&1
done
timeout 120 python tools/bench_conv.py 16 72 72 256 768 --k1 --ln >> $L 2>&1
timeout 120 python tools/bench_conv.py 16 72 72 256 1024 --k1 --ln --gelu >> $L 2>&1
timeout 120 python tools/bench_conv.py 16 72 72 256 256 --k1 --res >> $L 2>&1
timeout 120 python tools/bench_conv.py 16 72 72 1024 256 --k1 --res >> $L 2>&1
cat $L
