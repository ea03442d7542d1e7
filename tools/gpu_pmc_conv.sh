cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pc_a $O/pc_b $O/pc_c
CMD="python $R/tools/bench_conv.py 16 144 144 256 256 --gn --res --gn-part --iters 5"
export FEMASR_BF16_CLS=${CLS:-0}
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $O/pc_a -o a -- $CMD > $O/pc_a.log 2>&1; echo "a rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pc_b -o b -- $CMD > $O/pc_b.log 2>&1; echo "b rc=$?"
timeout 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pc_c -o c -- $CMD > $O/pc_c.log 2>&1; echo "c rc=$?"
tail -3 $O/pc_c.log
cd $R; python tools/rocpd_sq_summary.py $(find $O/pc_a $O/pc_b $O/pc_c -name "*.db") --filter bf16x3_kernel
