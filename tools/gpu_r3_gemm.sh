# round 3: GEMM block configurations (64x64 tiles for small launches): parity tests of every configuration, then the timing table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "linear" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python tools/bench_gemm.py --batches 1 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_gemm.log
