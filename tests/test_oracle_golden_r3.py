"""Round-3 pins of the CPU oracle against full-size golden vectors recorded from the REFERENCE (tests/golden/make_golden_r3.py):
the units of BASELINE configs 4 and 5 at their real sizes (x2 SR of a 256x256 tile, HQ autoencode of a 512x512 image).
Tolerance 1e-3 max-abs fp32 (north-star); VQ indices exact up to the documented near-tie rule (currently zero mismatches)."""
import numpy as np
import pytest

from femasr_amd import synth
from helpers import cfg_name_of, check_indices_near_tie, load_golden, oracle_net, probe_err, synth_weights

TOL = 1e-3


@pytest.mark.parametrize('name', ['x2_tile256_trained', 'hq_full512_trained'])
def test_full_size_units_match_reference(name):
    """~20 s of CPU each (1075 / 544 GFLOP through the C oracle)."""
    g = load_golden(name)
    cn = cfg_name_of(g)
    net = oracle_net(cn, synth_weights(cn, int(g['seed']), str(g['codebook'])))
    x = synth.synth_input(int(g['input_seed']), tuple(g['in_shape']))
    net.probes = {}
    y, idx = net.test(x, return_indices=True) if str(g['mode']) == 'test' else net.forward(x)
    st = int(g['out_stride'])
    assert y.shape == tuple(g['out_shape'])
    assert np.abs(y[:, :, ::st, ::st] - g['output']).max() < TOL
    assert abs(float(y.astype(np.float64).mean()) - float(g['out_mean'])) < 1e-5
    nbad, _ = check_indices_near_tie(idx, g)
    assert nbad == 0
    for k, v in net.probes.items():
        if 'probe_pos_' + k in g:
            err, scale = probe_err(g, k, v)
            assert err <= 2e-5 * max(scale, 1.0), (k, err, scale)
