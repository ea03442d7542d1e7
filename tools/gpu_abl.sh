cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/abl.log; : > $L
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k bf16 2>&1 | tail -5 >> $L
cat $L
