cd $GRAFT_REPO_ROOT
A="16 288 288 128 128 --gn --res --gn-part --iters 5"
for d in 0 1 2 3 4 8 11; do FEMASR_WINO_DBG=$d python tools/bench_conv.py $A --wino 2>&1 | tail -1 | sed "s/^/dbg=$d  /" | cut -c1-120; done
