# round 3: kernel trace of batch-1 forwards -> launch-by-launch timeline (small-batch latency work); SUITE=1 runs the GPU suite first
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
if [ -n "$SUITE" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-200; fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/prof_b1
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_b1 -o trace -- python $R/bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-bf16x3-leg > $R/$O/prof_b1.log 2>&1; echo "trace rc=$?"
cd $R
python tools/rocpd_timeline.py $(find $O/prof_b1 -name "*.db" | head -1) pad_nchw_to_nhwc $O/b1_timeline.txt | head -3
