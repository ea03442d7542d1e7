cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/h4.log; : > $L
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for v in 0 1; do
if [ $v = 1 ]; then export FEMASR_HALO_4W=1; fi
timeout 120 python tools/bench_conv.py 16 576 576 64 64 --gn --res --fp32 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 288 288 128 64 --up2 --fp32 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py 16 576 576 64 64 --fp32 2>&1 | grep conv >> $L
done
cat $L
