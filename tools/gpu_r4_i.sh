# round 4, call I: non-temporal cache policy on the Winograd kernels' streaming accesses (nt1 = input patches, nt2 = residuals / outputs, nt3 = both);
# the GEMMs' 64x64-tile tail rule with three sub-batch streams
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for so in new nt1 nt2 nt3; do
  if [ $so = new ]; then unset FEMASR_SO; else export FEMASR_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$so.so; fi
  for shp in "16 144 144 256 256" "16 288 288 128 128" "16 576 576 64 64"; do
    echo -n "$so: "; timeout 120 python tools/bench_conv.py $shp --gn --res --gn-part --fast-act --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
  for shp in "16 144 144 256 128" "16 288 288 128 64"; do
    echo -n "$so: "; timeout 120 python tools/bench_conv.py $shp --up2 --gn-part --iters 10 --wino 2>&1 | grep -v amdgpu.ids | tail -1
  done
done > $O/i_ab.log 2>&1
unset FEMASR_SO
cat $O/i_ab.log | cut -c1-200
for so in new nt1 nt3; do
  if [ $so = new ]; then unset FEMASR_TEST_SO; else export FEMASR_TEST_SO=$GRAFT_REPO_ROOT/tools/dbg/libfemasr_hip_$so.so; fi
  echo -n "bench lib=$so: "; timeout 300 python - <<PY 2>/dev/null | tail -1
import os, sys, runpy
so = os.environ.get('FEMASR_TEST_SO')
if so:
    from femasr_amd import _lib
    _lib.SO_PATH = so
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-bf16x3-leg', '--no-profile']
import io, contextlib, json
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path('bench.py', run_name='__main__')
j = json.loads(buf.getvalue().strip().splitlines()[-1])
print(j['ms_per_step'], j['value'])
PY
done
unset FEMASR_TEST_SO
echo -n "bench FEMASR_GEMM_TAIL_PCT=90: "; FEMASR_GEMM_TAIL_PCT=90 timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --no-profile 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'])"
