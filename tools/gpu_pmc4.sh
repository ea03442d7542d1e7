cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pmc_i $O/pmc_j
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_CACHE)[A-Z_]*)\b" | sort -u | head -30 > $O/icache_counters.txt; cat $O/icache_counters.txt
CMD="python $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-exact-leg"
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_i -o i -- $CMD > $O/pmc_i.log 2>&1; echo "i rc=$?"; tail -3 $O/pmc_i.log
cd $R; python tools/rocpd_sq_summary.py $(find $O/pmc_i -name "*.db") 2>&1 | grep -E "calls=|ICACHE|IFETCH" | head -80
