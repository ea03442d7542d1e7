# round 4, call L: the codebook lookup as ONE launch (exact phase inside the candidate kernel's block): parity of both forms, timings
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vq_twopass.py -q -x -p no:cacheprovider > $O/l_vq_fused.log 2>&1; echo "vq tests (one launch) rc=$?"; tail -2 $O/l_vq_fused.log | cut -c1-300
FEMASR_VQ_FUSED=0 timeout 900 python -m pytest tests/test_gpu_vq_twopass.py -q -x -p no:cacheprovider > $O/l_vq_two.log 2>&1; echo "vq tests (two launches) rc=$?"; tail -2 $O/l_vq_two.log | cut -c1-300
for f in 1 0; do
  echo "FEMASR_VQ_FUSED=$f"; FEMASR_VQ_FUSED=$f timeout 300 python tools/bench_vq.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
  FEMASR_VQ_FUSED=$f timeout 300 python tools/bench_vq.py --m 31104 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
done > $O/l_bench_vq.log 2>&1; cat $O/l_bench_vq.log
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_network_r3.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "vq or network or golden or tile or batch" > $O/l_network.log 2>&1; echo "network / vq kernel tests rc=$?"; tail -2 $O/l_network.log | cut -c1-300
for rep in 1 2; do
  for f in 1 0; do
    echo -n "bench FEMASR_VQ_FUSED=$f: "; FEMASR_VQ_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --no-profile 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'], j.get('power'))"
  done
done
