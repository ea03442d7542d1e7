# rocprofv3 passes for profiles/ over the bench of record (fp32 mode, serialized: --streams 1 so that kernels do not
# overlap and per-kernel durations are separable): kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs:
# TCC slot limit; SQ/GRBM pass for MFMA busy and the effective clock).  TAG names the output files.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; TAG=${TAG:-r02}
rm -rf $O/prof_trace $O/prof_fetch $O/prof_write $O/prof_sq
CMD="python $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-bf16x3-leg ${BENCH_ARGS:-}"
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_trace -o trace -- $CMD > $O/prof_trace.log 2>&1; echo "trace rc=$?" > $O/prof_rc.txt
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_fetch -o fetch -- $CMD > $O/prof_fetch.log 2>&1; echo "fetch rc=$?" >> $O/prof_rc.txt
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_write -o write -- $CMD > $O/prof_write.log 2>&1; echo "write rc=$?" >> $O/prof_rc.txt
timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_sq -o sq -- $CMD > $O/prof_sq.log 2>&1; echo "sq rc=$?" >> $O/prof_rc.txt
cat $O/prof_rc.txt
cd $R
python tools/rocpd_summary.py $(find $O/prof_trace -name "*.db" | head -1) $O/${TAG}_kernel_stats_streams1.txt | cut -c1-200 | head -30
python tools/rocpd_pmc_summary.py $O/${TAG}_pmc_summary_streams1.txt $(find $O/prof_fetch $O/prof_write $O/prof_sq -name "*.db") | cut -c1-160 | head -30
