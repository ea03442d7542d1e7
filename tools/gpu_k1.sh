cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/k1.log; : > $L
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for f in 0 1 2 3 4 7 8 15; do
FEMASR_DBG=$f timeout 120 python tools/bench_conv.py 16 72 72 1024 256 --k1 --res 2>&1 | grep conv >> $L
FEMASR_DBG=$f timeout 120 python tools/bench_conv.py 16 72 72 256 768 --k1 --ln 2>&1 | grep conv >> $L
done
cat $L
