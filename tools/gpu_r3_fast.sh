cd $GRAFT_REPO_ROOT
for so in "" tools/dbg/libfemasr_hip_fastact.so; do
  export FEMASR_SO=${so:+$GRAFT_REPO_ROOT/$so}; [ -z "$so" ] && unset FEMASR_SO; echo "== ${so:-default}"
  for sh in "16 288 288 128 128" "16 576 576 64 64" "16 144 144 256 256"; do
    timeout 120 python tools/bench_conv.py $sh --gn --res --gn-part --iters 5 --wino 2>&1 | tail -1
  done
done
