cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for a in 0 1; do FEMASR_BN256=$a timeout 300 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/ex_$a.log 2>&1; echo "bn256=$a"; tail -1 gpurun_out/ex_$a.log | python tools/bench_summary.py | grep -E "MPix|halo"; done
