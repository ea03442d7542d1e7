#!/bin/bash
# One parameterised driver for `gpurun` calls (replaces the per-call gpu_r*.sh scripts of rounds 2-4):
#   gpurun --timeout N -- 'bash tools/gpu_run.sh <task> [<task> ...]'         logs -> gpurun_out/<TAG>_<task>.log  (TAG from the env, default r05)
# tasks:  suite            every -m gpu test (SUITE_K='expr' selects, SUITE_ARGS='files')
#         smoke            __graft_entry__.smoke()
#         bench            default bench.py (BENCH_ARGS='...')
#         bench_quick      bench.py without CPU baseline / secondary legs / profile
#         prof             rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE / SQ passes of two serialized steps -> profiles-ready tables
#         other            bench.py for the other BASELINE configs (x2b32, hq8, tile2048)
#         parity           tools/parity_report.py
#         ubench_gemm      tools/ubench/gemm_bf16s (stand-alone split GEMM driver; UB_ARGS)
#         pmc_calib        tools/ubench/pmc_calib under the FETCH_SIZE / WRITE_SIZE (+ request) counters: what the counters read for known bytes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O; TAG=${TAG:-r05}
export TMPDIR=/tmp
for task in "$@"; do
  case $task in
    suite)
      if [ -n "$SUITE_K" ]; then timeout ${SUITE_T:-1500} python -m pytest ${SUITE_ARGS:-tests} -m gpu -k "$SUITE_K" -q -s -p no:cacheprovider > $O/${TAG}_suite.log 2>&1
      else timeout ${SUITE_T:-1500} python -m pytest ${SUITE_ARGS:-tests} -m gpu -q -s -p no:cacheprovider > $O/${TAG}_suite.log 2>&1; fi
      echo "suite rc=$?"; grep -n "^E  \|Error\|passed\|failed\|testset/:\|OST_120 tiled" $O/${TAG}_suite.log | head -40 | cut -c1-600 ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_smoke.log | tail -3 ;;
    bench)
      timeout ${BENCH_T:-900} python bench.py $BENCH_ARGS 2>$O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench.json; echo "bench rc=$?"
      python tools/bench_summary.py $O/${TAG}_bench.json 2>/dev/null | head -60 ;;
    bench_quick)
      timeout 600 python bench.py --no-cpu-baseline --no-bf16x3-leg --no-profile $BENCH_ARGS 2>/dev/null | tail -1 > $O/${TAG}_bench_quick.json
      python -c "import json; j=json.load(open('$O/${TAG}_bench_quick.json')); print('quick bench:', j['ms_per_step'], 'ms/step', j['value'], j['unit'], j['timed_region']['step_ms_first_median_last'], j.get('power', {}).get('package_w_mean'))" ;;
    prof)
      bash tools/gpu_prof.sh > $O/${TAG}_prof.log 2>&1; echo "prof rc=$?"; tail -5 $O/${TAG}_prof.log ;;
    other)
      for wl in x2b32 hq8; do timeout 600 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_$wl.json; python -c "import json; j=json.load(open('$O/${TAG}_bench_$wl.json')); print('$wl', j['ms_per_step'], 'ms/step', j['value'], j['unit'])"; done ;;
    parity)
      timeout 900 python tools/parity_report.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_parity_report.txt | tail -20 ;;
    ubench_gemm)
      (cd tools/ubench && GS_NOVERIFY=${GS_NOVERIFY:-1} GS_WARM=${GS_WARM:-400} timeout 300 ./gemm_bf16s ${UB_ARGS:-100}) 2>&1 | tee $O/${TAG}_ubench_gemm.log | tail -12 ;;
    pmc_calib)
      # known-bytes kernels under the FETCH_SIZE / WRITE_SIZE counters (tools/ubench/pmc_calib.hip; built here: hipcc cross-compiles)
      R=$PWD; ( cd /tmp; rm -rf $R/$O/calib_*; B=$R/tools/ubench/pmc_calib
        timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/calib_w -o w -- $B > $R/$O/calib_w.log 2>&1; echo "write rc=$?"
        timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/calib_f -o f -- $B > $R/$O/calib_f.log 2>&1; echo "fetch rc=$?"
        timeout 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $R/$O/calib_q -o q -- $B > $R/$O/calib_q.log 2>&1; echo "wrreq rc=$?"
        timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $R/$O/calib_r -o r -- $B > $R/$O/calib_r.log 2>&1; echo "rdreq rc=$?" )
      python tools/rocpd_counters.py $(find $O/calib_w $O/calib_f $O/calib_q $O/calib_r -name "*.db") 2>&1 | tee $O/${TAG}_pmc_calib.txt | cut -c1-200 ;;
    *) echo "unknown task $task" ;;
  esac
done
