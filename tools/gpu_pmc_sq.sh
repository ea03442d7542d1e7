cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pmc_a $O/pmc_b
CMD="python $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-exact-leg"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_a -o a -- $CMD > $O/pmc_a.log 2>&1; echo "a rc=$?"
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_b -o b -- $CMD > $O/pmc_b.log 2>&1; echo "b rc=$?"
cd $R; python tools/rocpd_sq_summary.py $(find $O/pmc_a $O/pmc_b -name "*.db") --filter igemm > $O/sq_igemm.txt; python tools/rocpd_sq_summary.py $(find $O/pmc_a $O/pmc_b -name "*.db") --filter halo > $O/sq_halo.txt; head -50 $O/sq_igemm.txt
