# rocprofv3 passes for profiles/: kernel trace + two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs: TCC slot limit)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
CMD="python $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-exact-leg"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_trace -o trace -- $CMD > $O/prof_trace.log 2>&1; echo "trace rc=$?" > $O/prof_rc.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_fetch -o fetch -- $CMD > $O/prof_fetch.log 2>&1; echo "fetch rc=$?" >> $O/prof_rc.txt
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_write -o write -- $CMD > $O/prof_write.log 2>&1; echo "write rc=$?" >> $O/prof_rc.txt
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_sq -o sq -- $CMD > $O/prof_sq.log 2>&1; echo "sq rc=$?" >> $O/prof_rc.txt
ls -la $O/prof_*/ ; cat $O/prof_rc.txt; tail -3 $O/prof_sq.log | cut -c1-300
