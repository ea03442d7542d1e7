cd $GRAFT_REPO_ROOT
for tag in tt ttfast; do
  export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$tag.so; echo "== $tag"
  timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part --iters 5 --wino 2>&1 | tail -3
done
