cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" 2>&1 | tail -2
for tag in default nostag yield1 yield3 yield4; do
  if [ $tag = default ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$tag.so; fi
  for sh in "16 288 288 128 128" "16 576 576 64 64" "16 144 144 256 256"; do
    echo -n "$tag $sh fast: "; timeout 120 python tools/bench_conv.py $sh --gn --res --gn-part --iters 5 --wino --fast-act 2>&1 | tail -1 | sed 's/.*cls=-: //'
  done
  echo -n "$tag exact 128: "; timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part --iters 5 --wino 2>&1 | tail -1 | sed 's/.*cls=-: //'
done
