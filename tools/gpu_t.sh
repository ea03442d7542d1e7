# run a pytest selection on the GPU box and keep the log: ARGS="tests/x.py" K="name or other" bash tools/gpu_t.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ -n "$K" ]; then
  timeout ${T:-900} python -m pytest ${ARGS:-tests -m gpu} -k "$K" -x -q -s -p no:cacheprovider > gpurun_out/pytest_sel.log 2>&1; echo "rc=$?"
else
  timeout ${T:-900} python -m pytest ${ARGS:-tests -m gpu} -x -q -s -p no:cacheprovider > gpurun_out/pytest_sel.log 2>&1; echo "rc=$?"
fi
grep -n "Error\|^E \|passed\|failed\|tiled \[" gpurun_out/pytest_sel.log | head -30 | cut -c1-400
