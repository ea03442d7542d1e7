cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/k1.log; : > $L
for B in 16 8; do
timeout 120 python tools/bench_conv.py $B 72 72 256 768 --k1 --ln 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py $B 72 72 256 1024 --k1 --ln --gelu 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py $B 72 72 256 256 --k1 --res 2>&1 | grep conv >> $L
timeout 120 python tools/bench_conv.py $B 72 72 1024 256 --k1 --res 2>&1 | grep conv >> $L
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-leg 2>&1 | tail -1 | python tools/bench_summary.py | head -4 >> $L
cat $L
