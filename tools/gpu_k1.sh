cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/k1.log; : > $L
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for v in 0 1; do
if [ $v = 1 ]; then export FEMASR_IGEMM_WIDE=1; fi
for B in 16 8; do
timeout 120 python tools/bench_conv.py $B 72 72 256 768 --k1 --ln 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py $B 72 72 256 1024 --k1 --ln --gelu 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py $B 72 72 256 256 --k1 --res 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py $B 72 72 1024 256 --k1 --res 2>&1 | grep conv >> $L
done; done
cat $L
