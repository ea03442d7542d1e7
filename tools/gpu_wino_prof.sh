# PMC counters of the Winograd conv kernel on one decoder layer (separate passes; --kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/wi_sq $O/wi_sq2 $O/wi_sq3
CMD="python $R/tools/bench_conv.py ${WSHAPE:-16 288 288 128 128} ${WFLAGS:---gn --res --gn-part} --iters 5 --wino"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/wi_sq -o sq -- $CMD > $O/wi_sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --kernel-trace -d $O/wi_sq2 -o sq -- $CMD > $O/wi_sq2.log 2>&1; echo "sq2 rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace -d $O/wi_sq3 -o sq -- $CMD > $O/wi_sq3.log 2>&1; echo "sq3 rc=$?"
cd $R
python - <<'PY'
import sqlite3,glob
for db in sorted(glob.glob('gpurun_out/wi_sq*/*.db')):
    cur=sqlite3.connect(db).cursor()
    try:
        rows=list(cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%wino4_kernel%' group by kernel_name, counter_name"))
    except Exception as e:
        print(db, 'ERR', e); continue
    for name,ctr,n,val,dur in rows:
        print(ctr, n, f'{val:.5g}', f'{dur/1e3:.1f}us')
PY
