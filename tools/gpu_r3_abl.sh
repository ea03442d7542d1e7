# ablation timings of the Winograd kernel (tools/build_debug.sh ablN builds; garbage results, timings only)
cd $GRAFT_REPO_ROOT
echo -n "base: "; timeout 120 python tools/bench_conv.py ${WSHAPE:-16 288 288 128 128} --gn --res --gn-part --iters 5 --wino ${WFLAGS:---fast-act} 2>&1 | tail -1 | sed 's/.*cls=-: //'
for tag in ${ABLS:-abl1 abl2 abl4 abl8 abl16 abl64 abl24 abl103 abl71}; do
  export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$tag.so; echo -n "$tag: "
  timeout 120 python tools/bench_conv.py ${WSHAPE:-16 288 288 128 128} --gn --res --gn-part --iters 5 --wino ${WFLAGS:---fast-act} 2>&1 | tail -1 | sed 's/.*cls=-: //'
done
