cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ARGS="tests/test_gpu_network.py" K="cli_counterpart" bash tools/gpu_t.sh
timeout 300 python tools/latency.py --quick 2>&1 | grep -v amdgpu.ids | tee gpurun_out/latency_quick.log
