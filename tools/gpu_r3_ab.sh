# A/B of the library against experiment builds under tools/dbg/ (libfemasr_hip_<tag>.so): TAGS="default base" bash tools/gpu_r3_ab.sh
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -k "winograd or conv" -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2 3; do
for tag in ${TAGS:-default}; do
  if [ $tag = default ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$tag.so; fi
  for sh in "16 288 288 128 128" "16 576 576 64 64" "16 144 144 256 256"; do
    echo -n "$tag $sh fast: "; timeout 120 python tools/bench_conv.py $sh --gn --res --gn-part --iters 10 --wino --fast-act 2>&1 | tail -1 | sed 's/.*cls=-: //'
  done
done; done
