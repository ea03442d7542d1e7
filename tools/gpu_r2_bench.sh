# round-2: default bench (fp32 of record + bf16x3 leg + cpu baselines), tile2048 on one GPU, one-rank RCCL gather
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python bench.py > $O/r2_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r2_bench.log | cut -c1-6000
timeout 900 python bench.py --workload tile2048 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/r2_tile2048.log 2>&1; echo "tile2048 rc=$?"; tail -1 $O/r2_tile2048.log | cut -c1-1500
timeout 600 python bench.py --force-gather --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-bf16x3-leg > $O/r2_fg.log 2>&1; echo "force-gather rc=$?"; tail -1 $O/r2_fg.log | cut -c1-800
