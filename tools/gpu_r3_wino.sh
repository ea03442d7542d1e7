# round 3: Winograd F(4x4,3x3) kernel - parity tests, then per-layer timings of the decoder shapes (exact and hardware SiLU)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" 2>&1 | tail -5
for fa in "" "--fast-act"; do
for sh in "16 288 288 128 128" "16 576 576 64 64" "16 144 144 256 256"; do
  timeout 120 python tools/bench_conv.py $sh --gn --res --gn-part --iters 5 --wino $fa 2>&1 | tail -1
done; done
timeout 120 python tools/bench_conv.py 16 72 72 512 256 --iters 5 --wino 2>&1 | tail -1
