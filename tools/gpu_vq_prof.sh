# per-kernel durations + SQ counters of the codebook-lookup kernels (tools/bench_vq.py, one regime)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; TAG=${TAG:-vq}
rm -rf $O/vq_trace $O/vq_sq $O/vq_sq2
CMD="python $R/tools/bench_vq.py --regime ${REGIME:-trained}"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/vq_trace -o trace -- $CMD > $O/vq_trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/vq_sq -o sq -- $CMD > $O/vq_sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace -d $O/vq_sq2 -o sq -- $CMD > $O/vq_sq2.log 2>&1; echo "sq2 rc=$?"
cd $R
python tools/rocpd_summary.py $(find $O/vq_trace -name "*.db" | head -1) $O/${TAG}_kernel_stats.txt | cut -c1-200 | head -16
python tools/rocpd_pmc_summary.py $O/${TAG}_pmc_summary.txt $(find $O/vq_sq $O/vq_sq2 -name "*.db") | cut -c1-250 | head -30
