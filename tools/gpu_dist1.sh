cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-leg --force-gather > gpurun_out/bench_fg.log 2>&1; echo "bench force-gather rc=$?"; tail -1 gpurun_out/bench_fg.log | cut -c1-200; grep -i "error\|Traceback" gpurun_out/bench_fg.log | head -5
echo skip
