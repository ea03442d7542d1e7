cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for a in 16 7 23; do FEMASR_ABLATE=$a timeout 300 python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/ab_$a.log 2>&1; echo "ablate=$a"; tail -1 gpurun_out/ab_$a.log | python tools/bench_summary.py | grep -E "MPix|halo"; done
