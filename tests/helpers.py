"""Shared test helpers: golden loading, synthetic weights, the VQ near-tie rule."""
import json
import os

import numpy as np

from femasr_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CONFIGS = {
    'x4': dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4),
    'x2': dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=2),
    'hq': dict(codebook_params=[[32, 1024, 512]], LQ_stage=False),
}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def key_table(cfg_name):
    with open(os.path.join(GOLDEN, 'state_dict_keys.json')) as f:
        return json.load(f)[cfg_name]


def cfg_name_of(g):
    if not int(g['cfg_LQ_stage']):
        return 'hq'
    return 'x4' if int(g['cfg_scale_factor']) == 4 else 'x2'


def synth_weights(cfg_name, seed, codebook):
    """{key: ndarray} for every float tensor of the config, from the committed key table."""
    out = {}
    for key, shape, dtype in key_table(cfg_name):
        leaf = key.rsplit('.', 1)[-1]
        if leaf in ('relative_position_index', 'attn_mask'):
            continue
        out[key] = synth.synth_tensor(seed, key, tuple(shape), codebook)
    return out


_ORACLE_MEMO = {}


def _fingerprint(v):
    import hashlib
    if isinstance(v, dict):
        h = hashlib.sha1()
        for k in sorted(v):
            h.update(k.encode())
            h.update(np.ascontiguousarray(v[k]).tobytes())
        return h.hexdigest()
    if isinstance(v, np.ndarray):
        return (v.shape, str(v.dtype), hashlib.sha1(np.ascontiguousarray(v).tobytes()).hexdigest())
    if isinstance(v, (list, tuple)):
        return tuple(_fingerprint(e) for e in v)
    return repr(v)


class _MemoOracle:
    """OracleNet whose forward-type calls are memoised for the test session: the restated matrix-instruction arithmetic makes one full-size
    oracle forward cost ~20 s of CPU, and several tests check the GPU path against the SAME (weights, input) in different modes
    (the GPU suite has to stay well inside the driver's step limit).  Key: weights, constructor, method, arguments,
    the oracle's Winograd limits.  Results are returned as copies."""

    def __init__(self, net, key):
        self._net, self._key = net, key

    def __getattr__(self, name):
        attr = getattr(self._net, name)
        if not callable(attr) or name not in ('test', 'test_tile', 'forward', 'decode_indices', 'encode_and_decode'):
            return attr

        def call(*a, **kw):
            from oracle import oracle as orc
            k = (self._key, name, _fingerprint(a), _fingerprint(sorted(kw.items())), tuple(orc.WINO_LOG2_LIMITS))
            if k not in _ORACLE_MEMO:
                _ORACLE_MEMO[k] = attr(*a, **kw)
            r = _ORACLE_MEMO[k]
            return tuple(np.copy(e) if isinstance(e, np.ndarray) else e for e in r) if isinstance(r, tuple) else (np.copy(r) if isinstance(r, np.ndarray) else r)
        return call


def oracle_net(cfg_name, weights, linear_math='bf16_split'):
    """linear_math: arithmetic of the 1x1 / Linear layers - 'bf16_split' = the product default (FeMaSRNet.linear_math),
    'fp32' = the fp32 fmaf chain (FeMaSRNet(linear_math='fp32')); the CPU-only golden tests run their large cases in 'fp32'
    (the restated matrix-instruction arithmetic costs ~30x the fmaf chain on a CPU)."""
    from oracle import oracle as orc
    cfg = CONFIGS[cfg_name] if isinstance(cfg_name, str) else cfg_name
    net = orc.OracleNet(weights, codebook_params=cfg['codebook_params'], LQ_stage=cfg['LQ_stage'], scale_factor=cfg.get('scale_factor', 4),
                        linear_math=linear_math)
    return _MemoOracle(net, (_fingerprint(weights), repr(sorted((k, repr(v)) for k, v in cfg.items())), linear_math))


def golden_cfg(g):
    """Constructor kwargs of a round-2 golden (tests/golden/make_golden_r2.py)."""
    return dict(codebook_params=np.asarray(g['codebook_params']).tolist(), LQ_stage=bool(int(g['cfg_LQ_stage'])),
                scale_factor=int(g['cfg_scale_factor']))


def weights_from_arch(cfg, seed, codebook='trained', variant='default'):
    """{key: ndarray} for any constructor config, key/shape table taken from the build's own arch (CPU construction)."""
    from femasr_amd.archs import build_network
    net = build_network(dict(type='FeMaSRNet', **cfg))
    return synth.fill_state_dict(net.state_dict(), seed, codebook, variant)


def ulp_of(x):
    return np.spacing(np.abs(np.asarray(x, np.float32)))


def check_indices_near_tie(idx, g, max_ulp=None):
    """VQ index parity against the reference with the documented near-tie allowance (hard part 1):
    a mismatch is accepted only if the REFERENCE's own distances of its winner and its runner-up are
    within `max_ulp` ulp AND the candidate picked that runner-up.  Returns (n_mismatch, n_accepted)."""
    max_ulp = 2.0 if max_ulp is None else max_ulp      # (single-call goldens: callers assert n_mismatch == 0, the allowance only labels; the tiled /
                                                       # testset fixtures pass oracle.near_tie.NEAR_TIE_ULP = 4 themselves)
    ref = g['vq_indices'].reshape(-1)
    idx = np.asarray(idx).reshape(-1)
    assert idx.shape == ref.shape
    bad = np.nonzero(idx != ref)[0]
    accepted = 0
    for r in bad:
        gap = float(g['vq_d_second'][r]) - float(g['vq_d_best'][r])
        tol = max_ulp * float(ulp_of(g['vq_d_best'][r]))
        if gap <= tol and idx[r] == g['vq_idx_second'][r]:
            accepted += 1
    return len(bad), accepted


def probe_err(g, name, arr):
    pos = g['probe_pos_' + name]
    val = g['probe_val_' + name]
    flat = np.ascontiguousarray(arr).reshape(-1)
    assert tuple(g['probe_shape_' + name]) == tuple(arr.shape), (name, g['probe_shape_' + name], arr.shape)
    got = flat[pos]
    return float(np.max(np.abs(got - val))), float(np.max(np.abs(val)))
