# round 4, call M: one-launch lookup with the deeper code-row prefetch; which hwmon files PowerWatch has to read
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vq_twopass.py -q -x -p no:cacheprovider > $O/m_vq.log 2>&1; echo "vq tests rc=$?"; tail -2 $O/m_vq.log | cut -c1-300
for f in 1 0; do
  echo "FEMASR_VQ_FUSED=$f"; FEMASR_VQ_FUSED=$f timeout 300 python tools/bench_vq.py 2>&1 | grep -v amdgpu.ids | grep "two-pass" | cut -c1-300
  FEMASR_VQ_FUSED=$f timeout 300 python tools/bench_vq.py --m 31104 2>&1 | grep -v amdgpu.ids | grep "two-pass" | cut -c1-300
done > $O/m_bench_vq.log 2>&1; cat $O/m_bench_vq.log
bash tools/hwmon_diag.sh > $O/m_hwmon.log 2>&1; cat $O/m_hwmon.log | cut -c1-200
