# round 4, call O: the candidate search with its LDS operand one k step ahead (two register sets): parity of both forms, timings
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vq_twopass.py -q -x -p no:cacheprovider 2>&1 | tail -1 | cut -c1-200
FEMASR_VQ_FUSED=0 timeout 600 python -m pytest tests/test_gpu_vq_twopass.py -q -x -p no:cacheprovider 2>&1 | tail -1 | cut -c1-200
for f in 1 0; do
  echo "# FEMASR_VQ_FUSED=$f"
  FEMASR_VQ_FUSED=$f timeout 300 python tools/bench_vq.py 2>&1 | grep -v amdgpu.ids | grep "two-pass\|MFMA rate" | cut -c1-300
  FEMASR_VQ_FUSED=$f timeout 300 python tools/bench_vq.py --m 31104 2>&1 | grep -v amdgpu.ids | grep "two-pass" | cut -c1-300
done > $O/o_vq.log 2>&1; cat $O/o_vq.log
timeout 600 python -m pytest tests/test_gpu_network.py -q -x -p no:cacheprovider -k "golden or reference or default" 2>&1 | tail -1 | cut -c1-200
