cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python tools/latency.py > $O/latency.log 2>&1; echo "rc=$?"; grep -v Warn $O/latency.log | tail -40
