cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/dbg_r3_tiled.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dbg_tiled.log | tail -30
