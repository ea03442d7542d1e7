cd $GRAFT_REPO_ROOT
for s in 2 3 4; do echo "streams=$s"; timeout 300 python bench.py --streams $s --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-bf16x3-leg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
