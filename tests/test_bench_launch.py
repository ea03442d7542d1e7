"""bench.py launches its own ranks: `python bench.py --gpus N` with WORLD_SIZE unset must re-exec under
torch.distributed.run, barrier, all-gather and print ONE JSON line with n_gpus == N.  Exercised here on CPU with
`--backend gloo --dry-net` (stand-in network; the launch, partition, gather and paste code is the product's)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '2'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--dry-net',
                        '--steps', '2', '--warmup', '1'] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    last = [ln for ln in p.stdout.splitlines() if ln.strip()][-1]
    return json.loads(last)


@pytest.mark.parametrize('extra,scaling,units', [
    (['--batch', '2'], 'weak', 4),
    (['--workload', 'tile2048', '--image', '384', '--batch', '4'], 'strong', 9),
])
def test_self_launch_two_ranks(extra, scaling, units):
    r = _run(extra)
    assert r['n_gpus'] == 2 and r['steps'] == 2 and r['warmup'] == 1
    assert r['scaling'] == scaling and r['config']['global_batch'] == units
    assert r['config']['gather'] is True and r['config']['backend'] == 'gloo'
    assert r['value'] > 0 and r['higher_is_better'] is True and r['vs_baseline'] is None
    assert 'dry-net' in r['data']
    if scaling == 'weak':
        # one driver call, both scalings: the default workload on N > 1 ranks appends the strong-scaling leg
        ss = r['strong_scaling']
        assert ss['scaling'] == 'strong' and ss['value'] > 0 and ss['ms_per_step'] > 0 and len(ss['per_rank_ms_per_step']) == 2
        assert sum(ss['tiles_per_rank']) == 16 and ss['partition_bound'] == 2.0        # dry net: a 512x512 stand-in image = 16 tiles of 128
    else:
        assert 'strong_scaling' not in r
