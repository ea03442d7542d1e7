# PMC passes over the stand-alone split-GEMM driver (tools/ubench/gemm_bf16s): MFMA busy, effective clock, waits, L2 requests per kernel variant.
# gpurun -- 'bash tools/gemm_pmc.sh'  ->  gpurun_out/${TAG}_gemm_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; TAG=${TAG:-r06}
B=$R/tools/ubench/gemm_bf16s
export GS_NOVERIFY=1 GS_WARM=${GS_WARM:-30}
rm -rf $O/gp_*
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/gp_sq -o sq -- $B 5 > $O/gp_sq.log 2>&1; echo "sq rc=$?"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/gp_sq2 -o sq2 -- $B 5 > $O/gp_sq2.log 2>&1; echo "sq2 rc=$?"
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $O/gp_tcc -o tcc -- $B 5 > $O/gp_tcc.log 2>&1; echo "tcc rc=$?"
timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --kernel-trace -d $O/gp_ta -o ta -- $B 5 > $O/gp_ta.log 2>&1; echo "ta rc=$?"
cd $R
python tools/rocpd_counters.py $(find $O/gp_sq $O/gp_sq2 $O/gp_tcc $O/gp_ta -name "*.db") 2>&1 | tee $O/${TAG}_gemm_pmc.txt | cut -c1-400
