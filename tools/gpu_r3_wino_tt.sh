# round 3: per-phase cycle shares of the Winograd kernel (debug build, tools/build_debug.sh)
cd $GRAFT_REPO_ROOT
export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_tt.so
for sh in "16 288 288 128 128" "16 576 576 64 64"; do
  timeout 120 python tools/bench_conv.py $sh --gn --res --gn-part --iters 5 --wino 2>&1 | tail -3
done
timeout 120 python tools/bench_conv.py 16 288 288 128 128 --iters 5 --wino 2>&1 | tail -3
