# SQ counters for the qkv-shaped GEMM (K=256, N=768, M=82944) under GEMM pipeline config $CFG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/pl_a $O/pl_b $O/pl_c
export FEMASR_GEMM_CFG=${CFG:-0}
CMD="python $R/tools/bench_conv.py 1 82944 1 256 ${N:-768} --k1 ${EXTRA:-} --iters 5"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $O/pl_a -o a -- $CMD > $O/pl_a.log 2>&1; echo "a rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pl_b -o b -- $CMD > $O/pl_b.log 2>&1; echo "b rc=$?"
timeout 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pl_c -o c -- $CMD > $O/pl_c.log 2>&1; echo "c rc=$?"
cd $R; python tools/rocpd_sq_summary.py $(find $O/pl_a $O/pl_b $O/pl_c -name "*.db") --filter gemm_dma | tee $O/sq_gemm_cfg${CFG:-0}.txt
