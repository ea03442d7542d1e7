"""-m gpu, round 6: the uint8 tile path (FeMaSRNet.test_tile_u8: crops, forwards, the all-gather and the paste on bytes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_test_tile_u8_equals_the_fp32_tile_path(cuda_device):
    """FeMaSRNet.test_tile_u8 (uint8 crops -> test_u8 -> uint8 paste; inference_femasr.py:58-67 with femasr_arch.py:387-447) ==
    output_to_u8(test_tile(u8_to_input(img))) bit for bit: ragged border tiles, several shape classes, batch 2, BGR order, and
    `test_u8(out=...)` writing into a slice."""
    import gpu_utils as G
    from femasr_amd import imgproc, synth
    from helpers import synth_weights
    w = synth_weights('x4', 11, 'trained')
    net = G.build_net('x4', w, cuda_device)
    rng = np.random.default_rng(5)
    img = torch.from_numpy(rng.integers(0, 256, (2, 75, 100, 3), dtype=np.uint8)).to(cuda_device)
    want = torch.stack([imgproc.output_to_u8(net.test_tile(imgproc.u8_to_input(img[i]), 32, 8)) for i in range(2)])
    got = net.test_tile_u8(img, 32, 8)
    assert got.dtype == torch.uint8 and got.shape == (2, 300, 400, 3) and torch.equal(got, want), int((got != want).sum())
    one = net.test_tile_u8(img[1], 32, 8)                               # (H,W,3) in -> (sH,sW,3) out
    assert torch.equal(one, want[1])
    bgr = net.test_tile_u8(img.flip(-1), 32, 8, bgr=True)               # cv2-style channel order in and out
    assert torch.equal(bgr.flip(-1), want)
    buf = torch.zeros((3, 160, 192, 3), dtype=torch.uint8, device=cuda_device)
    y = net.test_u8(img[:1, :40, :48], out=buf[1:2])
    assert y.data_ptr() == buf[1:2].data_ptr() and torch.equal(buf[1:2], net.test_u8(img[:1, :40, :48])) and int(buf[0].max()) == 0 and int(buf[2].max()) == 0
    with pytest.raises(ValueError):
        net.test_u8(img[:1, :40, :48], out=buf[1:2, :, :100])
