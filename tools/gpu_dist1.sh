cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo skip bench
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 tools/dist_sanity.py > gpurun_out/dist_sanity.log 2>&1; echo "sanity rc=$?"; tail -3 gpurun_out/dist_sanity.log | cut -c1-300
