"""-m gpu, round 4:
  * both sides of the Winograd-form size limit (the kernels address tensors with 32-bit byte offsets; a layer beyond the limit runs
    in the direct / phase-filter form - femasr_conv_wino_shape_ok, oracle.wino_fits): the limit is moved down to sizes a test can
    allocate with the femasr_debug_wino_limits hook, on the GPU and in the oracle together;
  * the whole reference testset/ directory through the CLI arithmetic (tests/golden/make_golden_r4.py), see test_testset_dir."""
import numpy as np
import pytest
import torch

from femasr_amd import _lib, synth
from helpers import oracle_net, synth_weights

pytestmark = pytest.mark.gpu


@pytest.fixture
def wino_limits():
    """(log2_total, log2_image) -> set on the library and on the oracle; restored afterwards."""
    from oracle import oracle as orc
    lib = _lib.load()
    prev = list(orc.WINO_LOG2_LIMITS)

    def set_(total, image):
        _lib.check(lib.femasr_debug_wino_limits(total, image))
        orc.WINO_LOG2_LIMITS[:] = [total or 31, image or 27]
    yield set_
    _lib.check(lib.femasr_debug_wino_limits(0, 0))
    orc.WINO_LOG2_LIMITS[:] = prev


@pytest.mark.parametrize('log2_image,crossing', [(19, 'the 64-channel 96x96 layers and the x2 conv that makes them fall back'),
                                                 (18, 'the 128-channel 48x48 layers fall back too'),
                                                 (12, 'every layer falls back: the fp32_direct network')])
def test_winograd_size_limit_both_sides(cuda_device, wino_limits, log2_image, crossing):
    """x4 on a 32x32 input (padded to 48: decoder layers 256 @ 24^2 = 2^17.2, 128 @ 48^2 = 2^18.2, 64 @ 96^2 = 2^19.2 elements per
    image).  With the per-image limit between two of them, part of the decoder runs in the Winograd forms and part in the direct
    forms INSIDE one forward: the plan's GroupNorm-partial sizes, the weights each launch takes and the oracle's rule must agree
    layer by layer.  'fp32_strict': bit-identical to the oracle under the same limit; and within 1e-4 of the unlimited result."""
    import gpu_utils as G
    w = synth_weights('x4', 4, 'trained')
    x = synth.synth_input(77, (2, 3, 32, 32))
    xg = torch.from_numpy(x).to(cuda_device)
    net0 = G.build_net('x4', w, cuda_device, decoder_math='fp32_strict')
    y0, i0 = net0.test_with_indices(xg)
    y0, i0 = y0.cpu().numpy(), i0.cpu().numpy()
    del net0
    wino_limits(0, log2_image)
    net = G.build_net('x4', w, cuda_device, decoder_math='fp32_strict')
    y, idx = net.test_with_indices(xg)
    y, idx = y.cpu().numpy(), idx.cpu().numpy()
    yo, io = oracle_net('x4', w).test(x, return_indices=True)
    assert np.array_equal(idx, io) and np.array_equal(idx, i0)
    assert np.array_equal(y, yo), f'limit 2^{log2_image} ({crossing}): max-abs vs oracle {np.abs(y - yo).max():.3e}'
    d = float(np.abs(y - y0).max())
    assert 0.0 < d < 1e-4, d          # a different algebraic form for some layers: different rounding, same function
    prof_names = [k for k in net.profile()] if hasattr(net, 'profile') else []
    print(f'limit 2^{log2_image}: {crossing}; max-abs vs the unlimited forward {d:.2e}; bit-identical to the oracle under the same limit', prof_names[:0])
