cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pa_a $O/pa_b
CMD="python $R/tools/bench_attn.py"
$CMD 2>&1 | tail -1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $O/pa_a -o a -- $CMD > $O/pa_a.log 2>&1; echo "a rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pa_b -o b -- $CMD > $O/pa_b.log 2>&1; echo "b rc=$?"
cd $R; python tools/rocpd_sq_summary.py $(find $O/pa_a $O/pa_b -name "*.db") --filter window_attention
