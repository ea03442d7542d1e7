cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( cd tools/ubench; for o in 0 1 2; do echo "== GS_ORDER=$o"; GS_WARM=400 GS_NOVERIFY=1 timeout 120 ./gemm_bf16s_o$o 100 2>&1 | grep -v "^resident" | head -6; done; echo "== verify GS_ORDER=1"; timeout 200 ./gemm_bf16s_o1 20 2>&1 | head -9; echo "== verify GS_ORDER=2"; timeout 200 ./gemm_bf16s_o2 20 2>&1 | head -9 ) > $O/r05f_ubench_order.log 2>&1
cut -c1-200 $O/r05f_ubench_order.log
TAG=r05f SUITE_ARGS="tests/test_gpu_kernels.py tests/test_gpu_vq_twopass.py tests/test_gpu_r4.py tests/test_gpu_r5.py tests/test_gpu_fuzz.py" SUITE_K="wino or vq or up2 or linear or twopass or candidates or network" SUITE_T=600 bash tools/gpu_run.sh suite
TAG=r05f bash tools/gpu_run.sh bench_quick pmc_calib
timeout 120 python tools/bench_vq.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/r05f_bench_vq.txt
