cd $GRAFT_REPO_ROOT
for A in "16 288 288 128 128 --gn --res --gn-part" "16 288 288 128 128 --gn" "16 144 144 256 256 --gn --res --gn-part" "16 576 576 64 64 --gn --res --gn-part" "16 72 72 512 256"; do
  python tools/bench_conv.py $A --iters 5 --fp32 2>&1 | tail -1
  python tools/bench_conv.py $A --iters 5 --wino 2>&1 | tail -1
done
