cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for s in 1 2 3 4; do timeout 300 python bench.py --steps 4 --warmup 2 --streams $s --no-cpu-baseline --no-profile --no-exact-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('streams', d['config']['streams'], 'MPix/s', d['value'], 'ms', d['ms_per_step'])
"; done
