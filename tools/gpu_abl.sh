cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FEMASR_FUZZ_MULT=3 timeout 1700 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --tb=short -p no:cacheprovider -k "test_tile" > gpurun_out/fuzz.log 2>&1
tail -30 gpurun_out/fuzz.log | cut -c1-220
