cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/abl.log; : > $L
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k bf16 2>&1 | tail -8 >> $L
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for cls in 0 3; do
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 144 144 256 256 --gn --res --gn-part >> $L 2>&1
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 144 144 256 256 >> $L 2>&1
FEMASR_BF16_CLS=$cls timeout 120 python tools/bench_conv.py 16 144 144 256 256 --up2 >> $L 2>&1
done
FEMASR_BF16_CLS=1 timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part >> $L 2>&1
FEMASR_BF16_CLS=0 timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part >> $L 2>&1
FEMASR_BF16_CLS=1 timeout 120 python tools/bench_conv.py 16 576 576 64 64 --gn --res --gn-part >> $L 2>&1
FEMASR_BF16_CLS=1 timeout 120 python tools/bench_conv.py 16 288 288 128 64 --up2 >> $L 2>&1
FEMASR_BF16_CLS=2 timeout 120 python tools/bench_conv.py 16 576 576 64 3 >> $L 2>&1
cat $L
