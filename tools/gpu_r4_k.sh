# round 4, call K: what the chip does under the two F(4x4,3x3) block shapes: engine clock and package power sampled (rocm-smi, 2 Hz) while
# one layer shape runs in a loop; then the bench line of both forms, alternating, twice each
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_r4.py -q -x -p no:cacheprovider -k "both_block_shapes" > $O/k_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/k_tests.log | cut -c1-300
watch_run() {      # $1 = tag, rest = command
  tag=$1; shift
  "$@" > $O/k_run_$tag.log 2>&1 &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    (date +%s.%N; timeout 5 rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power" | tr -s ' ' | cut -c1-120) | tr '\n' ' '; echo
    sleep 0.4
  done > $O/k_smi_$tag.log
  wait $pid
  grep -v amdgpu.ids $O/k_run_$tag.log | tail -1 | cut -c1-200
}
for form in 0 1; do
  export FEMASR_WINO_C128=$form
  watch_run wino288_c$form timeout 200 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part --fast-act --iters 4000 --wino
  watch_run wino144_c$form timeout 200 python tools/bench_conv.py 16 144 144 256 256 --gn --res --gn-part --fast-act --iters 4000 --wino
done
unset FEMASR_WINO_C128
watch_run halo timeout 200 python tools/bench_conv.py 16 144 144 128 128 --gn --res --gn-part --fp32 --iters 3000
watch_run gemm timeout 200 python tools/bench_conv.py 16 72 72 256 1024 --k1 --fp32 --gelu --iters 12000
for tag in wino288_c0 wino288_c1 wino144_c0 wino144_c1 halo gemm; do echo "== $tag"; awk 'NR % 3 == 0' $O/k_smi_$tag.log | tail -6 | cut -c1-260; done
for rep in 1 2; do
  for form in 0 1; do
    echo -n "bench c128=$form: "; FEMASR_WINO_C128=$form timeout 300 python bench.py --no-cpu-baseline --no-bf16x3-leg --no-profile 2>/dev/null | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'])"
  done
done
