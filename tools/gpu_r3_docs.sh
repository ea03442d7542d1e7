# round 3: numbers for DESIGN.md - rocprofv3 passes over the bench of record, small-batch latency, the other BASELINE configs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${TAG:-r03_run1} bash tools/gpu_prof.sh > gpurun_out/prof_summary.log 2>&1; tail -5 gpurun_out/prof_rc.txt
cd $GRAFT_REPO_ROOT
timeout 600 python tools/latency.py > gpurun_out/latency.log 2>&1; grep "graph=0" gpurun_out/latency.log | head -20
timeout 900 python tools/other_configs.py > gpurun_out/other_configs.log 2>&1; tail -12 gpurun_out/other_configs.log
