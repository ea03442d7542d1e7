# round 4, call G: the fused Swin MLP kernel - parity, stand-alone timing against the two GEMM launches, then the network with / without it
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_r4.py -q -x -p no:cacheprovider -k "mlp_fused or out_parameter" > $O/g_mlp_test.log 2>&1; echo "mlp test rc=$?"; tail -12 $O/g_mlp_test.log | cut -c1-250
timeout 300 python tools/bench_mlp.py > $O/g_mlp_bench.log 2>&1; cat $O/g_mlp_bench.log | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider > $O/g_net.log 2>&1; echo "network+kernel tests rc=$?"; tail -3 $O/g_net.log | cut -c1-250
for m in fused split; do
  echo -n "bench FEMASR_MLP=$m: "; FEMASR_MLP=$m timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'], {k[:40]: v['ms_per_step'] for k, v in j['roofline']['per_kernel'].items() if 'mlp' in k or 'gemm' in k})"
done
