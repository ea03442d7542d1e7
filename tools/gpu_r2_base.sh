# round-2 baseline: fp32-mode (bit-exact) serialized kernel trace + bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/r2base
CMD="python $R/bench.py --decoder-math fp32 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-exact-leg"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r2base -o trace -- $CMD > $O/r2base.log 2>&1; echo "trace rc=$?"
cd $R
python tools/rocpd_summary.py $(find $O/r2base -name "*.db" | head -1) $O/r2base_kernel_stats.txt | cut -c1-220 | head -40
timeout 600 python bench.py --decoder-math fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-exact-leg > $O/r2base_bench.log 2>&1; tail -1 $O/r2base_bench.log | cut -c1-3000
