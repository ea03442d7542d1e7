cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for s in 1 2 3 4; do timeout 600 python bench.py --steps 3 --warmup 1 --streams $s --no-cpu-baseline --no-exact-leg > gpurun_out/bench_s$s.log 2>&1; echo "s=$s rc=$?"; tail -1 gpurun_out/bench_s$s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
